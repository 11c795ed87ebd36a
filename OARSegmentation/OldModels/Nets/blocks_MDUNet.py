"""Drop-in for the reference's OARSegmentation/OldModels/Nets/blocks_MDUNet.py (BatchNorm multi-scale block)."""
from dose_prediction_amd.blocks import conv_3_1_old as conv_3_1  # noqa: F401
