"""Drop-in for the reference's DosePrediction/Models/Networks/dose_pyfer.py: same names, HIP-backed implementation.

The reference's trainers star-import this module (train_light_pyfer.py:14 `from ...dose_pyfer import *`) and rely on the names
it re-exports from `NetworkTrainer.network_trainer import *` (dose_pyfer.py:15: torch, nn, optim, time, NetworkTrainer, ...),
so the same public surface is provided here.  This directory is a *namespace package portion* (no __init__.py, exactly like the
reference's own tree), so `DosePrediction.Train.*`, `DosePrediction.DataLoader.*`, ... keep resolving to the reference's files
when this repository is placed in front of it on PYTHONPATH."""
import time  # noqa: F401
from typing import Sequence, Union, Tuple  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
from torch import optim  # noqa: F401

try:        # the reference's own file when its tree is on the path (dose_pyfer.py:15)
    from NetworkTrainer.network_trainer import *  # noqa: F401,F403
except ImportError:
    pass
from dose_prediction_amd.blocks import ViT, UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock  # noqa: F401
from dose_prediction_amd.models.base_blocks import ModifiedUnetrUpBlock  # noqa: F401
from dose_prediction_amd.models.c3d import BaseUNet  # noqa: F401
from dose_prediction_amd.models.dose_pyfer import (  # noqa: F401
    Model, MainSubsetModel, ViTEncoder, PyMSCDecoder, create_pretrained_unet, ensure_tuple_rep)
