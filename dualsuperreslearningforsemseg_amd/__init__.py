"""MI355X-native DSRL stage-3 training hot path.  Importing the package does not need a GPU; every
operator does, and fails loudly without libdsrl_hip.so or a HIP device."""
from . import _lib, functional, nn_modules, ddp  # noqa: F401
from .models import DSRL  # noqa: F401
from .models.losses import FALoss  # noqa: F401
from .models.modules import ASPP  # noqa: F401

__version__ = '0.1.0'
