from .chassis import Bayes_optimize, L_BFGS_B_optimize, OptimizeFn

__all__ = ["Bayes_optimize", "L_BFGS_B_optimize", "OptimizeFn"]
