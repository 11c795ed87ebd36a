"""MI355X-native (gfx950) forward/backward of DOSE-PYFER and OAR-TRANSEG behind the reference's nn.Module surface."""
from .config import set_compute_dtype, compute_dtype  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["set_compute_dtype", "compute_dtype"]
