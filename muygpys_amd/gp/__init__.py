from .muygps import MuyGPS

__all__ = ["MuyGPS"]
