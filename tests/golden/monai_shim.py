"""Test-only stand-in for the parts of ``monai`` (pinned ``monai==0.7.0`` in the reference's
requirements.txt:5, NOT installed on this machine) that the reference's network files import.

Used ONLY by tests/golden/make_golden.py inside the authoring container, so that the reference's
own ``dose_pyfer.py`` / ``oar_transeg.py`` / ``base_blocks.py`` *wiring* can be executed and its
outputs committed as golden vectors (fixtures G7).  The blocks below are written from the published
semantics of MONAI 0.7.0 (plain torch eager); they pin nothing about MONAI itself -- the MONAI
leaves stay "parity unpinned" (oracle/__init__.py).  Never imported by the product.
"""
import sys
import types

import torch
import torch.nn as nn


def ensure_tuple_rep(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def get_padding(kernel_size, stride):
    return (kernel_size - stride + 1) // 2


def get_output_padding(kernel_size, stride, padding):
    return 2 * padding + stride - kernel_size


class _Conv(nn.Sequential):
    """monai.networks.blocks.Convolution(conv_only=True): nn.Sequential with one child 'conv'."""

    def __init__(self, conv):
        super().__init__()
        self.add_module("conv", conv)


def get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, act=None, norm=None,
                   dropout=None, bias=False, conv_only=True, is_transposed=False):
    assert spatial_dims == 3
    pad = get_padding(kernel_size, stride)
    if is_transposed:
        conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride, pad,
                                  get_output_padding(kernel_size, stride, pad), bias=bias)
    else:
        conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, pad, bias=bias)
    return _Conv(conv)


class UnetResBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name,
                 act_name=None, dropout=None):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, 1, stride)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1 = nn.InstanceNorm3d(out_channels)
        self.norm2 = nn.InstanceNorm3d(out_channels)
        self.norm3 = nn.InstanceNorm3d(out_channels)
        self.downsample = in_channels != out_channels or stride != 1

    def forward(self, inp):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        out = out + residual
        return self.lrelu(out)


class UnetBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name,
                 act_name=None, dropout=None):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1 = nn.InstanceNorm3d(out_channels)
        self.norm2 = nn.InstanceNorm3d(out_channels)

    def forward(self, inp):
        out = self.lrelu(self.norm1(self.conv1(inp)))
        return self.lrelu(self.norm2(self.conv2(out)))


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        cls = UnetResBlock if res_block else UnetBasicBlock
        self.layer = cls(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp):
        return self.layer(inp)


class UnetrPrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, num_layer, kernel_size, stride,
                 upsample_kernel_size, norm_name, conv_block=False, res_block=False):
        super().__init__()
        assert conv_block and res_block
        up = upsample_kernel_size
        self.transp_conv_init = get_conv_layer(spatial_dims, in_channels, out_channels, up, up, is_transposed=True)
        self.blocks = nn.ModuleList([
            nn.Sequential(
                get_conv_layer(spatial_dims, out_channels, out_channels, up, up, is_transposed=True),
                UnetResBlock(spatial_dims, out_channels, out_channels, kernel_size, stride, norm_name))
            for _ in range(num_layer)])

    def forward(self, x):
        x = self.transp_conv_init(x)
        for blk in self.blocks:
            x = blk(x)
        return x


class UnetrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name,
                 res_block=False):
        super().__init__()
        up = upsample_kernel_size
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, up, up, is_transposed=True)
        cls = UnetResBlock if res_block else UnetBasicBlock
        self.conv_block = cls(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        return self.conv_block(torch.cat((out, skip), dim=1))


class _Rearrange(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):  # "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)"
        import einops
        return einops.rearrange(x, "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)",
                                p1=self.p[0], p2=self.p[1], p3=self.p[2])


class PatchEmbeddingBlock(nn.Module):
    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed, dropout_rate,
                 spatial_dims):
        super().__init__()
        assert pos_embed == "perceptron" and spatial_dims == 3
        n_patches = 1
        for i, p in zip(img_size, patch_size):
            n_patches *= i // p
        patch_dim = in_channels * patch_size[0] * patch_size[1] * patch_size[2]
        self.patch_embeddings = nn.Sequential(_Rearrange(patch_size), nn.Linear(patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        nn.init.trunc_normal_(self.position_embeddings, std=0.02)

    def forward(self, x):
        return self.patch_embeddings(x) + self.position_embeddings


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)
        self.scale = (hidden_size // num_heads) ** -0.5

    def forward(self, x):
        import einops
        q, k, v = einops.rearrange(self.qkv(x), "b h (qkv l d) -> qkv b l h d", qkv=3, l=self.num_heads)
        att = (torch.einsum("blxd,blyd->blxy", q, k) * self.scale).softmax(dim=-1)
        x = torch.einsum("bhxy,bhyd->bhxd", att, v)
        return self.out_proj(einops.rearrange(x, "b h l d -> b l (h d)"))


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()

    def forward(self, x):
        return self.linear2(self.fn(self.linear1(x)))


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0):
        super().__init__()
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate)
        self.norm2 = nn.LayerNorm(hidden_size)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class ViT(nn.Module):
    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12,
                 num_heads=12, pos_embed="conv", classification=False, num_classes=2, dropout_rate=0.0,
                 spatial_dims=3):
        super().__init__()
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads,
                                                   pos_embed, dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)

    def forward(self, x):
        x = self.patch_embedding(x)
        hidden = []
        for blk in self.blocks:
            x = blk(x)
            hidden.append(x)
        return self.norm(x), hidden


def install():
    """Register the stand-in under the ``monai.*`` names the reference imports."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    dummy = lambda *a, **k: None  # noqa: E731  (non-arithmetic imports: never called by the networks)

    class _Names:  # Act.PRELU / Norm.INSTANCE are only used as default-argument values
        def __getattr__(self, k):
            return k.lower()
    mod("monai")
    mod("monai.utils", ensure_tuple_rep=ensure_tuple_rep)
    mod("monai.networks")
    mod("monai.networks.nets")
    mod("monai.networks.nets.vit", ViT=ViT)
    mod("monai.networks.blocks", ADN=dummy)
    mod("monai.networks.blocks.unetr_block", UnetrBasicBlock=UnetrBasicBlock, UnetrPrUpBlock=UnetrPrUpBlock,
        UnetrUpBlock=UnetrUpBlock)
    mod("monai.networks.blocks.dynunet_block", UnetBasicBlock=UnetBasicBlock, UnetResBlock=UnetResBlock,
        get_conv_layer=get_conv_layer, get_padding=get_padding, get_output_padding=get_output_padding)
    mod("monai.networks.layers")
    mod("monai.networks.layers.convutils", same_padding=dummy)
    mod("monai.networks.layers.factories", Act=_Names(), Norm=_Names())
    mod("monai.transforms", Activations=dummy, AsDiscrete=dummy, Compose=dummy)
