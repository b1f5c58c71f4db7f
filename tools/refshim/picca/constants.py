ABSORBER_IGM = {}


class Cosmo:
    def __init__(self, *a, **k):
        raise RuntimeError('picca.constants.Cosmo is a placeholder in refshim')
