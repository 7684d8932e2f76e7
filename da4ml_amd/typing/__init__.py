from ..cmvm import solver_options_t

__all__ = ['solver_options_t']
