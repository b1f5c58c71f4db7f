"""Inert placeholder for iminuit (the minimiser is outside the hot path)."""


class Minuit:
    def __init__(self, *a, **k):
        raise RuntimeError('iminuit is a placeholder in refshim')
