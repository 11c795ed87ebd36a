"""OAR-TRANSEG on the HIP path.  Mirrors OARSegmentation/Models/Networks/oar_transeg.py:14-185 (and, with ``old=True`` /
class TRANSEG, the OldModels variant the shipped trainer imports)."""
from typing import Sequence, Tuple, Union

import torch.nn as nn

from ..blocks import ViT, UnetrBasicBlock, UnetrPrUpBlock
from .base_blocks import ModifiedUnetrUpBlock, ModifiedUnetOutBlock
from .c3d import to_ndhwc, from_ndhwc
from .dose_pyfer import ensure_tuple_rep, run_vit_beside, small_blocks_beside


class Model(nn.Module):
    _OLD = False

    def __init__(self, in_channels: int, out_channels: int, img_size: Union[Sequence[int], int], feature_size: int = 16,
                 hidden_size: int = 768, mlp_dim: int = 3072, num_heads: int = 12, pos_embed: str = "conv",
                 norm_name: Union[Tuple, str] = "instance", conv_block: bool = True, res_block: bool = True,
                 dropout_rate: float = 0.0, spatial_dims: int = 3) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.num_layers = 12
        img_size = ensure_tuple_rep(img_size, spatial_dims)
        self.patch_size = ensure_tuple_rep(16, spatial_dims)
        self.feat_size = tuple(img_d // p_d for img_d, p_d in zip(img_size, self.patch_size))
        self.hidden_size = hidden_size
        self.classification = False
        self.vit = ViT(in_channels=in_channels, img_size=img_size, patch_size=self.patch_size, hidden_size=hidden_size,
                       mlp_dim=mlp_dim, num_layers=self.num_layers, num_heads=num_heads, pos_embed=pos_embed,
                       classification=self.classification, dropout_rate=dropout_rate, spatial_dims=spatial_dims)
        self.encoder1 = UnetrBasicBlock(spatial_dims, in_channels, feature_size, kernel_size=3, stride=1, norm_name=norm_name,
                                        res_block=res_block)
        self.encoder2 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 2, num_layer=2, kernel_size=3, stride=1,
                                       upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        self.encoder3 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 4, num_layer=1, kernel_size=3, stride=1,
                                       upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        self.encoder4 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 8, num_layer=0, kernel_size=3, stride=1,
                                       upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        old = self._OLD
        self.decoder5 = ModifiedUnetrUpBlock(spatial_dims, hidden_size, feature_size * 8, upsample_kernel_size=2, old=old)
        self.decoder4 = ModifiedUnetrUpBlock(spatial_dims, feature_size * 8, feature_size * 4, upsample_kernel_size=2, old=old)
        self.decoder3 = ModifiedUnetrUpBlock(spatial_dims, feature_size * 4, feature_size * 2, upsample_kernel_size=2, old=old)
        self.decoder2 = ModifiedUnetrUpBlock(spatial_dims, feature_size * 2, feature_size, upsample_kernel_size=2, old=old)
        self.out = ModifiedUnetOutBlock(spatial_dims=spatial_dims, in_channels=feature_size, out_channels=out_channels)
        self.proj_axes = (0, spatial_dims + 1) + tuple(d + 1 for d in range(spatial_dims))
        self.proj_view_shape = list(self.feat_size) + [self.hidden_size]

    def proj_feat(self, x):
        return x.view([x.size(0)] + self.proj_view_shape)       # free view in NDHWC (reference: 165-169)

    def forward_ndhwc(self, x_in):
        run, enc1 = run_vit_beside(self.vit, x_in, lambda: self.encoder1(x_in))   # the transformer on a second HIP stream
        with small_blocks_beside(x_in, run) as join:
            enc2 = self.encoder2(self.proj_feat(run.hidden(3)))
            enc3 = self.encoder3(self.proj_feat(run.hidden(6)))
            enc4 = self.encoder4(self.proj_feat(run.hidden(9)))
            join(enc2, enc3, enc4)
        dec3 = self.decoder5(self.proj_feat(run.final(enc1, enc2, enc3, enc4)), enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        return self.out(self.decoder2(dec1, enc1))

    def forward(self, x_in):
        return from_ndhwc(self.forward_ndhwc(to_ndhwc(x_in)))


class TRANSEG(Model):
    """OARSegmentation/OldModels/Networks/oar_transeg.py TRANSEG (BatchNorm multi-scale blocks, bare 1x1x1 conv)."""
    _OLD = True
