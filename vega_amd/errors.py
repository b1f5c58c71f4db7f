"""Error types with the reference's names (reference vega/utils.py:444-453)."""


class VegaModelError(Exception):
    pass


class VegaBoundsError(VegaModelError):
    pass


class VegaArinyoError(VegaModelError):
    pass
