"""MI355X-native (gfx950) forward/backward of DOSE-PYFER and OAR-TRANSEG behind the reference's nn.Module surface."""
from .config import (set_compute_dtype, compute_dtype, set_loss_scale, loss_scale, set_activation_checkpointing,  # noqa: F401
                     activation_checkpointing, compute_mode, compute_mode_as)
from . import _lib  # noqa: F401

__all__ = ["set_compute_dtype", "compute_dtype", "set_loss_scale", "loss_scale", "set_activation_checkpointing",
           "activation_checkpointing", "compute_mode", "compute_mode_as"]
