"""Leaf operators of the CPU oracle (fp32 PyTorch eager, NCDHW).  TEST INFRASTRUCTURE ONLY.

Each function states the arithmetic of one torch op the reference hot path issues
("Kernel worklist"); the HIP kernels are checked against these.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "conv3d", "conv_transpose3d_k2s2", "instance_norm", "batch_norm", "layer_norm", "activation",
    "trilinear_up2", "linear", "gelu", "attention", "patchify", "bf16_round",
]


def bf16_round(t):
    """Round an fp32 tensor to the nearest bf16 value (kept in fp32)."""
    return t.to(torch.bfloat16).to(torch.float32)


def conv3d(x, w, b=None, stride=1, padding=0, dilation=1):
    """nn.Conv3d (reference: c3d.py:16, blocks_MDUNet.py:68,102,146)."""
    return F.conv3d(x, w, b, stride=stride, padding=padding, dilation=dilation)


def conv_transpose3d_k2s2(x, w):
    """nn.ConvTranspose3d(kernel 2, stride 2, no bias) (reference: base_blocks.py:118-127 via
    MONAI get_conv_layer(is_transposed=True)).  w: [Cin, Cout, 2, 2, 2]."""
    return F.conv_transpose3d(x, w, None, stride=2)


def instance_norm(x, weight=None, bias=None, eps=1e-5):
    """nn.InstanceNorm3d, biased variance, eps 1e-5 (reference: c3d.py:17 affine,
    blocks_MDUNet.py:69 non-affine).  Stated explicitly rather than through F.instance_norm."""
    dims = tuple(range(2, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = x.var(dim=dims, unbiased=False, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    if weight is not None:
        shp = (1, -1) + (1,) * (x.dim() - 2)
        y = y * weight.view(shp) + bias.view(shp)
    return y


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm3d (reference: blocks_MDUNet.py:103,106).  Returns (y, new_running_mean,
    new_running_var); training mode normalises with biased batch variance and updates the running
    buffers with the unbiased one."""
    shp = (1, -1) + (1,) * (x.dim() - 2)
    if training:
        dims = (0,) + tuple(range(2, x.dim()))
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        n = x.numel() / x.shape[1]
        new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
        new_rv = (1 - momentum) * running_var + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    y = (x - mean.view(shp)) * torch.rsqrt(var.view(shp) + eps) * weight.view(shp) + bias.view(shp)
    return y, new_rm, new_rv


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dim (MONAI TransformerBlock.norm1/norm2, ViT.norm)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * weight + bias


def activation(x, kind):
    """ReLU / LeakyReLU(0.01) / Mish / GELU(erf) / identity."""
    if kind in (None, "none"):
        return x
    if kind == "relu":
        return torch.clamp_min(x, 0)
    if kind == "lrelu":
        return torch.where(x >= 0, x, 0.01 * x)
    if kind == "mish":
        return x * torch.tanh(F.softplus(x))
    if kind == "gelu":
        return gelu(x)
    raise ValueError(kind)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def trilinear_up2(x):
    """F.interpolate(scale_factor=2, mode='trilinear', align_corners=True) (reference: c3d.py:36),
    stated as three separable 1-D linear interpolations with src = dst*(in-1)/(out-1)."""
    def interp_axis(t, axis):
        n_in = t.shape[axis]
        n_out = 2 * n_in
        if n_in == 1:
            return t.repeat_interleave(2, dim=axis)
        pos = torch.arange(n_out, dtype=t.dtype) * torch.tensor((n_in - 1) / (n_out - 1), dtype=t.dtype)
        i0 = pos.floor().long().clamp_(0, n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        f = (pos - i0.to(t.dtype))
        shp = [1] * t.dim()
        shp[axis] = n_out
        f = f.view(shp)
        return t.index_select(axis, i0) * (1 - f) + t.index_select(axis, i1) * f
    for ax in (2, 3, 4):
        x = interp_axis(x, ax)
    return x


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def attention(x, qkv_w, out_w, out_b, num_heads):
    """MONAI 0.7.0 SABlock: qkv = Linear(h, 3h, bias=False); split "(qkv l d)" with l = heads;
    softmax(q k^T * d^-0.5) v; merge heads "(l d)"; out_proj.  [parity unpinned: MONAI absent]"""
    B, N, H = x.shape
    d = H // num_heads
    qkv = linear(x, qkv_w).view(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)  # qkv b l n d
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(B, N, H)
    return linear(o, out_w, out_b)


def patchify(x, p=16):
    """einops "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)" (MONAI PatchEmbeddingBlock,
    pos_embed='perceptron'; call site dose_pyfer.py:273)."""
    B, C, S0, S1, S2 = x.shape
    h, w, d = S0 // p, S1 // p, S2 // p
    x = x.view(B, C, h, p, w, p, d, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
    return x.reshape(B, h * w * d, p * p * p * C)
