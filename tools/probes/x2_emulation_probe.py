"""CPU emulation of candidate split schemes for the fp32x3 mode (3 -> 2 bf16/fp16 products), end to end on the
oracle: DOSE-PYFER at production width on a 64^3 sample, forward, float64 everywhere EXCEPT the operand roundings of the scheme
under test, applied where the HIP path would apply them (the k in {3, 7} stride-1 convolutions and the Linear layers).

    python tools/probes/x2_emulation_probe.py [size]        (test infrastructure: uses oracle/, nothing of the product)

schemes:  x16     x rounded to fp16 (x_hi only), weights exact   -> y = x_hi (w_hi + w_lo): TWO fp16 products
          w16     weights rounded to fp16, x exact               -> y = (x_hi + x_lo) w_hi:  TWO fp16 products
          xb/wb   the same with bf16 (the 2-product bf16 variants DESIGN section 3 rejects)
          x3      hi/lo bf16 of both, three products (what ships): only the lo.lo term missing
"""
import sys
import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                                   # noqa: E402
from oracle import ref_ops as R                  # noqa: E402
import torch.nn.functional as F                  # noqa: E402


def rnd(t, dt):
    return t.to(dt).to(t.dtype)


def f8_block(t):
    """e4m3 with one power-of-two scale per block of 32 elements along the contraction (channel) axis, dim 1 of activations [N, C, ...] and of
    weights [Cout, Cin, ...] (matrices [rows, K]: the last axis): the block's largest magnitude is scaled into [128, 256) (e4m3 tops out at
    448), smaller elements keep 4 significant bits down to 2^-6 of the scaled range and flush through the subnormals below."""
    ax = 1 if t.dim() > 2 else t.dim() - 1
    C = t.shape[ax]
    out = torch.empty_like(t)
    for c0 in range(0, C, 32):
        sl = [slice(None)] * t.dim()
        sl[ax] = slice(c0, min(C, c0 + 32))
        blk = t[tuple(sl)]
        amax = blk.abs().amax(dim=ax, keepdim=True).clamp_min(1e-300)
        sc = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)
        out[tuple(sl)] = (blk / sc).float().to(torch.float8_e4m3fn).to(t.dtype) * sc
    return out


def make(scheme, which="all"):
    def split2(t, dt):
        hi = rnd(t, dt)
        return hi, rnd(t - hi, dt)

    def contract(x, w, f):
        if scheme == "exact":
            return f(x, w)
        if scheme == "x16":
            return f(rnd(x, torch.float16), w)
        if scheme == "w16":
            return f(x, rnd(w, torch.float16))
        if scheme == "xb":
            return f(rnd(x, torch.bfloat16), w)
        if scheme == "wb":
            return f(x, rnd(w, torch.bfloat16))
        if scheme == "x3":
            xh, xl = split2(x, torch.bfloat16)
            wh, wl = split2(w, torch.bfloat16)
            return f(xh, wh) + f(xl, wh) + f(xh, wl)
        if scheme in ("h16f8", "hb16f8", "h16f8i"):
            # hi x hi in fp16 (or bf16) + the two correction terms on the block-scaled fp8 matrix instruction
            # (v_mfma_scale_f32_16x16x128_f8f6f4: e4m3 operands, one power-of-two scale per 32 contraction elements): 2 x the bf16 MFMA
            # time instead of 3 x.  "i": the idealised variant (4 significant bits, unlimited exponent range) -- what a per-ELEMENT scale
            # would give; the gap to it is the cost of sharing one scale among 32 channels.
            hdt = torch.bfloat16 if scheme == "hb16f8" else torch.float16
            xh, xl = split2(x, hdt)
            wh, wl = split2(w, hdt)
            q = (lambda t: oracle.round_bits(t, 4)) if scheme == "h16f8i" else f8_block
            return f(xh, wh) + f(q(xl), q(wh)) + f(q(xh), q(wl))
        if scheme == "x16w22":      # x_hi(fp16) x (w_hi + w_lo) with fp16 halves: what the kernel would really compute
            wh, wl = split2(w, torch.float16)
            return f(rnd(x, torch.float16), wh + wl)
        raise ValueError(scheme)

    def conv3d(x, w, b=None, stride=1, padding=0, dilation=1):
        k = w.shape[2]
        if k in (3, 7) and stride == 1 and dilation == 1 and w.shape[0] >= 8:
            y = contract(x, w, lambda a, c: F.conv3d(a, c, None, stride=stride, padding=padding, dilation=dilation))
            return y if b is None else y + b.view(1, -1, 1, 1, 1)
        if which == "all":      # 1x1x1 mixers / heads, stride-2 convolutions: same scheme
            y = contract(x, w, lambda a, c: F.conv3d(a, c, None, stride=stride, padding=padding, dilation=dilation))
            return y if b is None else y + b.view(1, -1, 1, 1, 1)
        return F.conv3d(x, w, b, stride=stride, padding=padding, dilation=dilation)

    def linear(x, w, b=None):
        y = contract(x, w, lambda a, c: a @ c.t())
        return y if b is None else y + b

    def tconv(x, w):
        if which == "all":
            return contract(x, w, lambda a, c: F.conv_transpose3d(a, c, None, stride=2))
        return F.conv_transpose3d(x, w, None, stride=2)
    return conv3d, linear, tconv


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from dose_prediction_amd import synth
    from dose_prediction_amd.models import dose_pyfer
    torch.manual_seed(4321)
    shape = (S, S, S)
    net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=shape, num_layers=8,
                           num_heads=6, act="mish")
    sd = {k: (v.detach().double() if v.dtype.is_floating_point else v.detach()) for k, v in net.state_dict().items()}
    x = synth.dose_input(1, shape).double()
    gt = synth.dose_target(1, shape)
    mask = gt[:, 1:2] > 0
    orig = (R.conv3d, R.linear, R.conv_transpose3d_k2s2)
    res = {}
    only = os.environ.get("PROBE_SCHEMES")
    todo = (("exact", "all"), ("x16", "all"), ("x16", "conv"), ("w16", "all"), ("x16w22", "all"), ("x3", "all"), ("xb", "all"), ("h16f8", "conv"), ("h16f8i", "conv"),
            ("hb16f8", "conv"))
    if only:
        todo = (("exact", "all"),) + tuple(t for t in todo if t[0] in only.split(","))
    for scheme, which in todo:
        R.conv3d, R.linear, R.conv_transpose3d_k2s2 = make(scheme, which)
        t0 = time.time()
        with torch.no_grad():
            out = oracle.dose_pyfer(sd, x, num_layers=8, num_heads=6, act="mish", training=True)
        R.conv3d, R.linear, R.conv_transpose3d_k2s2 = orig
        maps = [out[0]] + list(out[1])
        if scheme == "exact":
            ref = maps
            print(f"exact: {time.time() - t0:.1f} s", flush=True)
            continue
        errs = [float((m - r).abs().max() / r.abs().max()) for m, r in zip(maps, ref)]
        l2 = [float((m - r).norm() / r.norm()) for m, r in zip(maps, ref)]
        mae = float(70.0 * (maps[1] - ref[1]).abs()[mask].mean())
        res[(scheme, which)] = errs
        print(f"{scheme:7s} {which:4s}: rel-err-max [net_A, 128, 64, 32, 16] = {['%.2e' % e for e in errs]}  rel-L2 {['%.1e' % e for e in l2]}  "
              f"dose-MAE {mae:.2e} Gy  ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
