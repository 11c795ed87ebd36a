"""Drop-in for the reference's DosePrediction/Models/Networks/dose_pyfer.py: same names, HIP-backed implementation."""
from dose_prediction_amd.models.dose_pyfer import (  # noqa: F401
    Model, MainSubsetModel, ViTEncoder, PyMSCDecoder, create_pretrained_unet)
