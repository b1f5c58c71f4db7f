class MCSamples:
    def __init__(self, *a, **k):
        raise RuntimeError('getdist is a placeholder in refshim')
