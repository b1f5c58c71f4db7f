from . import constants  # noqa: F401
