from . import fits  # noqa: F401
