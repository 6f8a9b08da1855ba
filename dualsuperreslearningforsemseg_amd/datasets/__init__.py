from . import Cityscapes  # noqa: F401
