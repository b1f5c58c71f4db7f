"""Inert placeholder for mpi4py (not on the hot path)."""
MPI = None
