"""astropy.io.fits stand-in backed by the engine's own FITS reader."""
from vega_amd.fitslite import open, HDUList, HDU  # noqa: F401,A004
