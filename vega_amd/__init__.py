"""vega_amd - MI355X-native model + chi2 engine behind Vega's VegaInterface surface."""
from .interface import VegaInterface  # noqa: F401
from .errors import VegaModelError, VegaBoundsError, VegaArinyoError  # noqa: F401
