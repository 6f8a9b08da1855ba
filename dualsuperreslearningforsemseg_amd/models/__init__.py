from .DSRL import DSRL  # noqa: F401
