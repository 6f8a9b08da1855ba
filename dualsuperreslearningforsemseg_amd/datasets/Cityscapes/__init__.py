from . import settings  # noqa: F401
