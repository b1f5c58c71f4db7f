"""Inert placeholder for h5py (not on the hot path)."""
