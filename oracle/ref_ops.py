"""Leaf operators of the CPU oracle (fp32 PyTorch eager, NCDHW).  TEST INFRASTRUCTURE ONLY.

Each function states the arithmetic of one torch op the reference hot path issues
("Kernel worklist"); the HIP kernels are checked against these.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "conv3d", "conv_transpose3d_k2s2", "instance_norm", "batch_norm", "layer_norm", "activation",
    "trilinear_up2", "linear", "gelu", "attention", "patchify", "bf16_round", "storage", "store", "store_weight", "dice_ce_loss", "grad_noise", "round_bits",
]


def bf16_round(t):
    """Round an fp32 tensor to the nearest bf16 value (kept in fp32)."""
    return t.to(torch.bfloat16).to(torch.float32)


# ---------------------------------------------------------------------------------------------- 16-bit storage emulation
# The HIP path in bf16 / fp16 mode keeps fp32 accumulators inside every kernel and rounds ONLY where a tensor is stored:
# each op output (convolution, transposed convolution, Linear, fused norm(+residual)+activation, LayerNorm, GELU, residual
# add, trilinear up-sampling, attention output and the attention probabilities fed to the P.V MFMA), the packed weight copies,
# and the same points of the backward pass (gradients are stored in the same type).  `with storage(torch.bfloat16):` makes
# the oracle round at exactly those points (in whatever precision it otherwise computes, fp32 or fp64), so that
# err(oracle-with-storage vs fp64 oracle) is the error budget of the storage format itself -- what a correct 16-bit
# implementation must show, no more (tests/test_precision_budget_gpu.py).
_STORE = None


class storage:
    def __init__(self, dtype):
        self.dtype = None if dtype in (None, torch.float32, torch.float64) else dtype

    def __enter__(self):
        global _STORE
        self.prev, _STORE = _STORE, self.dtype
        return self

    def __exit__(self, *a):
        global _STORE
        _STORE = self.prev


class _Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dt):
        ctx.dt = dt
        return t.to(dt).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


# Round-off sensitivity of the BACKWARD pass (tools/probes/determinism_probe.py): `with grad_noise(eps, seed):` multiplies the gradient arriving
# at every stored tensor by (1 + eps u), u uniform in [-1, 1] per element -- one fp32 rounding (eps = 2^-24) at each point where the
# HIP path writes a gradient.  Run in float64, the spread of the results over a few seeds is the band inside which ANY correct fp32
# evaluation order of this network's backward pass must be expected to land; the forward values are untouched.
_GRAD_NOISE = None


class grad_noise:
    """bits: additionally round every stored gradient to that many significand bits AFTER the perturbation (8 = the operand grid of the
    fp32x3 mode's one-product backward pass and of the bf16 mode, 16 = its three-product backward pass).  A rounding to u = 2^-bits turns
    a perturbation of relative size d into one of size ~sqrt(d u) (the elements within d of a rounding boundary flip by a whole u), so a
    CHAIN of such roundings drives any perturbation, however small, to the grid's own noise level within a few layers."""

    def __init__(self, eps, seed=0, bits=None, fwd_eps=0.0):
        """fwd_eps: the same relative perturbation on the FORWARD value of every stored tensor (round-off that reaches the activation
        gates: a ReLU / LeakyReLU pre-activation within fwd_eps of zero flips, and the gradient field around it changes by O(1))."""
        self.cfg = (float(eps), torch.Generator().manual_seed(int(seed)), bits, float(fwd_eps))

    def __enter__(self):
        global _GRAD_NOISE
        self.prev, _GRAD_NOISE = _GRAD_NOISE, self.cfg
        return self

    def __exit__(self, *a):
        global _GRAD_NOISE
        _GRAD_NOISE = self.prev


def round_bits(t, bits):
    """t rounded to `bits` significand bits (round to nearest; bits = 8: the bf16 grid, 16: a [hi | lo] bf16 pair, 24: fp32)."""
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * (2.0 ** bits)) / (2.0 ** bits), e)


class _Noise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, cfg):
        ctx.cfg = cfg
        if cfg[3]:
            return t * (1 + cfg[3] * (torch.rand(t.shape, generator=cfg[1], dtype=t.dtype) * 2 - 1))
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        eps, gen, bits, _ = ctx.cfg
        u = torch.rand(g.shape, generator=gen, dtype=g.dtype) * 2 - 1
        g = g * (1 + eps * u)
        return (g if bits is None else round_bits(g, bits)), None


def store(t):
    """A tensor written to HBM by the HIP path: rounded to the storage type (forward value and incoming gradient)."""
    if _GRAD_NOISE is not None and t.requires_grad:
        t = _Noise.apply(t, _GRAD_NOISE)
    return t if _STORE is None else _Store.apply(t, _STORE)


def store_weight(w):
    """The packed 16-bit copy of an fp32 master weight (its gradient stays fp32: weight gradients are produced in fp32)."""
    if _STORE is None:
        return w
    return w + (w.detach().to(_STORE).to(w.dtype) - w.detach())


def conv3d(x, w, b=None, stride=1, padding=0, dilation=1):
    """nn.Conv3d (reference: c3d.py:16, blocks_MDUNet.py:68,102,146)."""
    return store(F.conv3d(x, store_weight(w), b, stride=stride, padding=padding, dilation=dilation))


def conv_transpose3d_k2s2(x, w):
    """nn.ConvTranspose3d(kernel 2, stride 2, no bias) (reference: base_blocks.py:118-127 via
    MONAI get_conv_layer(is_transposed=True)).  w: [Cin, Cout, 2, 2, 2]."""
    return store(F.conv_transpose3d(x, store_weight(w), None, stride=2))


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
    return store((x - mean) * torch.rsqrt(var + eps) * weight + bias)


def activation(x, kind):
    """ReLU / LeakyReLU(0.01) / Mish / identity, applied to a normalised tensor (+ residual).  In the HIP path this is the
    tail of the fused norm(+residual)+activation kernel, i.e. a storage point (instance_norm / batch_norm themselves are not)."""
    if kind in (None, "none"):
        return store(x)
    if kind == "relu":
        return store(torch.clamp_min(x, 0))
    if kind == "lrelu":
        return store(torch.where(x >= 0, x, 0.01 * x))
    if kind == "mish":
        return store(x * torch.tanh(F.softplus(x)))
    if kind == "gelu":
        return gelu(x)
    raise ValueError(kind)


def gelu(x):
    return store(0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0))))


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
    return store(x)


def linear(x, w, b=None):
    y = x @ store_weight(w).t()
    return store(y if b is None else y + b)


def attention(x, qkv_w, out_w, out_b, num_heads):
    """MONAI 0.7.0 SABlock: qkv = Linear(h, 3h, bias=False); split "(qkv l d)" with l = heads;
    softmax(q k^T * d^-0.5) v; merge heads "(l d)"; out_proj.  [parity unpinned: MONAI absent]"""
    B, N, H = x.shape
    d = H // num_heads
    qkv = linear(x, qkv_w).view(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)  # qkv b l n d
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = store(torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1))      # (16-bit P operand of the P.V MFMA)
    o = store((att @ v).permute(0, 2, 1, 3).reshape(B, N, H))
    return linear(o, out_w, out_b)


def patchify(x, p=16):
    """einops "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)" (MONAI PatchEmbeddingBlock,
    pos_embed='perceptron'; call site dose_pyfer.py:273)."""
    B, C, S0, S1, S2 = x.shape
    h, w, d = S0 // p, S1 // p, S2 // p
    x = x.view(B, C, h, p, w, p, d, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
    return x.reshape(B, h * w * d, p * p * p * C)


def dice_ce_loss(logits, labels, smooth_nr=1e-5, smooth_dr=1e-5, lambda_dice=1.0, lambda_ce=1.0):
    """MONAI 0.7.0 ``DiceCELoss(to_onehot_y=True, softmax=True)`` as OAR-TRANSEG builds and calls it
    (OARSegmentation/train_light_transeg.py:148, 196, 212): logits [B, C, ...], labels [B, 1, ...] (class indices, any dtype).
    DiceLoss.forward with the defaults (include_background, no squared_pred / jaccard / batch, reduction="mean"):
    input = softmax(logits, 1); target = one_hot(labels); reduce over the SPATIAL axes only; f = 1 - (2 I + smooth_nr) /
    (ground + pred + smooth_dr); mean over batch and channel.  DiceCELoss.ce: nn.CrossEntropyLoss(reduction="mean") on
    squeeze(labels, 1).long().  total = lambda_dice * dice + lambda_ce * ce.  [parity unpinned: MONAI absent; the CE half is torch's
    own F.cross_entropy, the Dice half is checked against a per-class loop in tests/test_oracle_leaves_cpu.py]"""
    C = logits.shape[1]
    lab = labels.reshape(labels.shape[0], *labels.shape[2:]).long()
    p = torch.softmax(logits, 1)
    onehot = torch.nn.functional.one_hot(lab, C).movedim(-1, 1).to(p.dtype)
    axes = list(range(2, logits.dim()))
    inter = (onehot * p).sum(axes)
    den = onehot.sum(axes) + p.sum(axes)
    dice = (1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)).mean()
    ce = torch.nn.functional.cross_entropy(logits, lab)
    return lambda_dice * dice + lambda_ce * ce
