from .FALoss import FALoss  # noqa: F401
