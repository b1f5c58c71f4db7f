"""Inert placeholder for pypolychord (not on the hot path)."""
