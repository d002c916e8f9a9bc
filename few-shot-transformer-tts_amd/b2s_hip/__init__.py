from .lib import load, B2SError, LIB_PATH, EXPORTS  # noqa: F401
