"""mcfit stand-in: the restated FFTLog of oracle/fftlog.py."""
from oracle.fftlog import P2xi  # noqa: F401
