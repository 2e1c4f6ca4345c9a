from .kernel_fn import RBF, KernelFn, Matern

__all__ = ["KernelFn", "Matern", "RBF"]
