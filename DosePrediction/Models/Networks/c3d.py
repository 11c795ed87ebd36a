"""Drop-in for the reference's DosePrediction/Models/Networks/c3d.py: same names, HIP-backed implementation
(namespace-package portion: no __init__.py, see dose_pyfer.py next to this file)."""
import time  # noqa: F401

import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from torch import optim  # noqa: F401

try:        # c3d.py:3 star-imports the trainer module
    from NetworkTrainer.network_trainer import *  # noqa: F401,F403
except ImportError:
    pass
from dose_prediction_amd.models.c3d import BaseUNet, Model, Encoder, Decoder  # noqa: F401
from dose_prediction_amd.blocks import SingleConv, UpConv  # noqa: F401
