"""Stand-in for picca.constants (absent from the image): the absorber rest wavelengths and the comoving-distance
cosmology restated in vega_amd/metal_matrices.py, so that the reference's `new_metals` matrix construction can run
on the same two inputs as the engine's set-up code."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from vega_amd.metal_matrices import ABSORBER_IGM, PiccaCosmo as Cosmo      # noqa: E402,F401
