from .ASPP import ASPP  # noqa: F401
