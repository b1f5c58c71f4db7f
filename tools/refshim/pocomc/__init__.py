"""Inert placeholder for pocomc (not on the hot path)."""
