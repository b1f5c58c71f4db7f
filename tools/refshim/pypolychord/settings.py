class PolyChordSettings:
    def __init__(self, *a, **k):
        raise RuntimeError('placeholder')
