from .ResNet101 import ResNet101  # noqa: F401
