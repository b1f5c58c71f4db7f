"""Inert placeholder for git (not on the hot path)."""
