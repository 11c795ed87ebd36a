"""C3D U-Net (net_A of the DOSE-PYFER cascade) on the HIP path.  Mirrors DosePrediction/Models/Networks/c3d.py:
same class names, constructor signatures, state_dict keys and forward I/O (NCDHW fp32 in / out)."""
import torch
import torch.nn as nn

from .. import config, ops
from ..blocks import SingleConv, UpConv


def to_ndhwc(x):
    """Boundary conversion: the trainer's NCDHW fp32 tensor (network_trainer.py:230) -> NDHWC compute dtype,
    channels zero-padded to a multiple of 16 (whole 16-channel chunks for the tiled convolution's fast staging path; a
    1-channel CT padded to 8 sent OAR-TRANSEG's first 3x3x3 layer through the guarded loads)."""
    c = x.shape[1]
    return ops.ToNDHWC.apply(x, (c + 15) // 16 * 16, config.compute_dtype())


def from_ndhwc(x):
    return ops.FromNDHWC.apply(x)


class Encoder(nn.Module):
    """c3d.Encoder (c3d.py:41-72)."""

    def __init__(self, in_ch, list_ch):
        super().__init__()
        for lvl in range(1, 6):
            cin = in_ch if lvl == 1 else list_ch[lvl - 1]
            setattr(self, f"encoder_{lvl}", nn.Sequential(
                SingleConv(cin, list_ch[lvl], kernel_size=3, stride=1 if lvl == 1 else 2, padding=1),
                SingleConv(list_ch[lvl], list_ch[lvl], kernel_size=3, stride=1, padding=1)))

    def forward(self, x):
        outs = []
        for lvl in range(1, 6):
            blk = getattr(self, f"encoder_{lvl}")
            # the first convolution's output feeds only the second one (fp32x3: its normalisation writes that convolution's split operand)
            x = blk[1](blk[0](x, next_conv=blk[1].single_conv[0]))
            outs.append(x)
        return outs


class Decoder(nn.Module):
    """c3d.Decoder (c3d.py:75-115)."""

    def __init__(self, list_ch):
        super().__init__()
        for lvl in (4, 3, 2, 1):
            setattr(self, f"upconv_{lvl}", UpConv(list_ch[lvl + 1], list_ch[lvl]))
            convs = [SingleConv(2 * list_ch[lvl], list_ch[lvl], kernel_size=3, stride=1, padding=1)]
            if lvl > 1:
                convs.append(SingleConv(list_ch[lvl], list_ch[lvl], kernel_size=3, stride=1, padding=1))
            setattr(self, f"decoder_conv_{lvl}", nn.Sequential(*convs))

    def forward(self, out_encoder):
        d = out_encoder[4]
        for lvl in (4, 3, 2, 1):
            up = getattr(self, f"upconv_{lvl}")(d)
            convs = getattr(self, f"decoder_conv_{lvl}")
            d = convs[0]((up, out_encoder[lvl - 1]),                                   # virtual torch.cat (c3d.py:103-113)
                         next_conv=convs[1].single_conv[0] if len(convs) > 1 else None)
            if len(convs) > 1:
                d = convs[1](d)
        return d


class BaseUNet(nn.Module):
    """c3d.BaseUNet (c3d.py:118-149), including its kaiming-uniform / IN(1,0) initialisation (127-142)."""

    def __init__(self, in_ch, list_ch):
        super().__init__()
        self.encoder = Encoder(in_ch, list_ch)
        self.decoder = Decoder(list_ch)
        self.initialize()

    @staticmethod
    def init_conv_IN(modules):
        for m in modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.)
            elif isinstance(m, nn.InstanceNorm3d):
                nn.init.constant_(m.weight, 1.)
                nn.init.constant_(m.bias, 0.)

    def initialize(self):
        self.init_conv_IN(self.decoder.modules)
        self.init_conv_IN(self.encoder.modules)

    def forward_ndhwc(self, x):
        return self.decoder(self.encoder(x))

    def forward(self, x):
        return from_ndhwc(self.forward_ndhwc(to_ndhwc(x)))


class Model(nn.Module):
    """c3d.Model: the two-U-Net cascade baseline (c3d.py:152-169)."""

    def __init__(self, in_ch, out_ch, list_ch_A, list_ch_B):
        super().__init__()
        self.net_A = BaseUNet(in_ch, list_ch_A)
        self.net_B = BaseUNet(in_ch + list_ch_A[1], list_ch_B)
        self.conv_out_A = nn.Conv3d(list_ch_A[1], out_ch, kernel_size=1, padding=0, bias=True)
        self.conv_out_B = nn.Conv3d(list_ch_B[1], out_ch, kernel_size=1, padding=0, bias=True)

    def forward(self, x):
        xh = to_ndhwc(x)
        a = self.net_A.forward_ndhwc(xh)
        b = self.net_B.forward_ndhwc(ops.cat((a, xh)))
        return [from_ndhwc(ops.conv3d(a, self.conv_out_A.weight, self.conv_out_A.bias)),
                from_ndhwc(ops.conv3d(b, self.conv_out_B.weight, self.conv_out_B.bias))]
