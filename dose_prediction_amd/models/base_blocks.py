"""OARSegmentation/Models/Nets/base_blocks.py counterparts on the HIP path (MultiUnetBasicBlock 12-28,
ModifiedUnetrUpBlock 91-141, ModifiedUnetOutBlock 144-165)."""
import torch.nn as nn

from .. import ops
from ..blocks import conv_3_1, conv_3_1_old, DualDilatedBlock, get_conv_layer, _run_conv


class MultiUnetBasicBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, multiS_conv=True, act="relu", old=False):
        super().__init__()
        if old:
            self.cov_ = conv_3_1_old(ch_in=in_channels, ch_out=out_channels)
        else:
            self.cov_ = conv_3_1(ch_in=in_channels, ch_out=out_channels, act=act) if multiS_conv else \
                DualDilatedBlock(ch_in=in_channels, ch_out=out_channels, act=act)

    def forward(self, inp):
        return self.cov_(inp)


class ModifiedUnetrUpBlock(nn.Module):
    """ConvTranspose3d(k2,s2,no bias) -> cat(out, skip) -> multi-scale conv block."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, upsample_kernel_size, act="relu",
                 norm="instance", multiS_conv=True, old=False) -> None:
        super().__init__()
        self.act = act
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=upsample_kernel_size,
                                          stride=upsample_kernel_size, conv_only=True, is_transposed=True, norm=norm)
        self.conv_block = MultiUnetBasicBlock(out_channels + out_channels, out_channels, act=act, multiS_conv=multiS_conv, old=old)

    def forward(self, inp, skip):
        out = _run_conv(self.transp_conv, inp)
        return self.conv_block((out, skip))          # virtual torch.cat((out, skip), dim=1): base_blocks.py:139


class ModifiedUnetOutBlock(nn.Module):
    """1x1x1 conv with bias (key conv.conv.{weight,bias})."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, dropout=None):
        super().__init__()
        self.conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=1, bias=True)

    def forward(self, inp):
        return _run_conv(self.conv, inp)
