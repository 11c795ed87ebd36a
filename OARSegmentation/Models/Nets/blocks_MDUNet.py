"""Drop-in for the reference's OARSegmentation/Models/Nets/blocks_MDUNet.py (the blocks on the hot path)."""
from dose_prediction_amd.blocks import (  # noqa: F401
    conv_3_1, conv_block_3, conv_block_7, dilated_conv_block_5, dilated_conv_block_7, DualDilatedBlock)
