"""Drop-in for global_variables.py: the only constant the hot path reads."""
g_zero_tol = 1.0e-6
