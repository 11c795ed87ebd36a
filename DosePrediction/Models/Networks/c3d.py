"""Drop-in for the reference's DosePrediction/Models/Networks/c3d.py: same names, HIP-backed implementation."""
from dose_prediction_amd.models.c3d import BaseUNet, Model, Encoder, Decoder  # noqa: F401
from dose_prediction_amd.blocks import SingleConv, UpConv  # noqa: F401
