#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the DOSE-PYFER layer shapes (HIP events on the launch stream).
usage: python tools/bench_conv.py [fwd|wgrad|all] [--dtype bf16|fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dose_prediction_amd import ops, _lib  # noqa: E402

# (name, N, Cin, Cout, S, k)   S^3 volume
SHAPES = [
    ("dec1.c7a 32->16 @128", 2, 32, 16, 128, 7), ("dec1.c7b 16->16 @128", 2, 16, 16, 128, 7),
    ("dec1.c7a.dgrad 16->32 @128", 2, 16, 32, 128, 7),
    ("dec2.c7a 64->32 @64", 2, 64, 32, 64, 7), ("dec2.c7b 32->32 @64", 2, 32, 32, 64, 7), ("dec2.dgrad 32->64 @64", 2, 32, 64, 64, 7),
    ("dec3.c7a 128->64 @32", 2, 128, 64, 32, 7), ("dec3.c7b 64->64 @32", 2, 64, 64, 32, 7), ("dec3.dgrad 64->128 @32", 2, 64, 128, 32, 7),
    ("c3 128->64 @32", 2, 128, 64, 32, 3), ("c3 64->128 @32", 2, 64, 128, 32, 3),
    ("dec4.c7a 256->128 @16", 2, 256, 128, 16, 7), ("dec4.c7b 128->128 @16", 2, 128, 128, 16, 7),
    ("c3 128->128 @16", 2, 128, 128, 16, 3), ("c3 256->128 @16", 2, 256, 128, 16, 3), ("dec4.dgrad 128->256 @16", 2, 128, 256, 16, 7),
    ("c3 32->16 @128", 2, 32, 16, 128, 3), ("c3 16->16 @128", 2, 16, 16, 128, 3), ("c3 25->16 @128", 2, 32, 16, 128, 3),
    ("c3 64->32 @64", 2, 64, 32, 64, 3), ("c3 32->32 @64", 2, 32, 32, 64, 3), ("c3 64->64 @32", 2, 64, 64, 32, 3),
]


# row lengths other than 128: the 96^3 crop of the segmentation network (OARSegmentation/config.py:24, the cascade's
# sliding-window roi, train_light_linked_model.py:152-154) and the 192-wide volume of BASELINE configs[4]; (D, H, W) tuples
SHAPES_W = [
    ("c7 16->16 @2x128^3", 2, 16, 16, (128, 128, 128), 7), ("c7 16->16 @4x96^3", 4, 16, 16, (96, 96, 96), 7), ("c7 16->16 @128x192x192", 1, 16, 16, (128, 192, 192), 7),
    ("c7 32->16 @2x128^3", 2, 32, 16, (128, 128, 128), 7), ("c7 32->16 @4x96^3", 4, 32, 16, (96, 96, 96), 7), ("c7 32->16 @128x192x192", 1, 32, 16, (128, 192, 192), 7),
    ("c3 16->16 @2x128^3", 2, 16, 16, (128, 128, 128), 3), ("c3 16->16 @4x96^3", 4, 16, 16, (96, 96, 96), 3), ("c3 16->16 @128x192x192", 1, 16, 16, (128, 192, 192), 3),
    ("c7 16->32 @2x128^3", 2, 16, 32, (128, 128, 128), 7), ("c7 16->32 @4x96^3", 4, 16, 32, (96, 96, 96), 7), ("c7 16->32 @128x192x192", 1, 16, 32, (128, 192, 192), 7),
    ("c7 64->32 @2x64^3", 2, 64, 32, (64, 64, 64), 7), ("c7 64->32 @4x48^3", 4, 64, 32, (48, 48, 48), 7), ("c7 64->32 @64x96x96", 1, 64, 32, (64, 96, 96), 7),
    ("c7 32->32 @2x64^3", 2, 32, 32, (64, 64, 64), 7), ("c7 32->32 @4x48^3", 4, 32, 32, (48, 48, 48), 7), ("c7 32->32 @64x96x96", 1, 32, 32, (64, 96, 96), 7),
    ("c7 64->64 @2x32^3", 2, 64, 64, (32, 32, 32), 7), ("c7 64->64 @4x24^3", 4, 64, 64, (24, 24, 24), 7), ("c7 64->64 @32x48x48", 1, 64, 64, (32, 48, 48), 7),
    ("c3 32->16 @2x128^3", 2, 32, 16, (128, 128, 128), 3), ("c3 32->16 @4x96^3", 4, 32, 16, (96, 96, 96), 3), ("c3 32->16 @128x192x192", 1, 32, 16, (128, 192, 192), 3),
    ("c3 16->32 @2x128^3", 2, 16, 32, (128, 128, 128), 3), ("c3 16->32 @4x96^3", 4, 16, 32, (96, 96, 96), 3), ("c3 16->32 @128x192x192", 1, 16, 32, (128, 192, 192), 3),
    ("c3 64->32 @2x64^3", 2, 64, 32, (64, 64, 64), 3), ("c3 64->32 @64x96x96", 1, 64, 32, (64, 96, 96), 3),
    ("c7 128->128 @2x16^3", 2, 128, 128, (16, 16, 16), 7), ("c7 128->128 @4x12^3", 4, 128, 128, (12, 12, 12), 7), ("c7 256->128 @4x12^3", 4, 256, 128, (12, 12, 12), 7),
    ("c7 128->64 @4x24^3", 4, 128, 64, (24, 24, 24), 7), ("c7 256->128 @2x16^3", 2, 256, 128, (16, 16, 16), 7),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--filter", default="")
    ap.add_argument("--widths", action="store_true", help="the W = 96 / 192 shape families next to their 128-wide twins (SHAPES_W)")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: the clock the chip holds without data toggling (DVFS diagnostic)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.dtype == "fp32x3":
        import dose_prediction_amd
        dose_prediction_amd.set_compute_dtype("fp32x3")
    dev = torch.device("cuda:0")
    for name, N, ci, co, S, k in (SHAPES_W if a.widths else SHAPES):
        if a.filter and a.filter not in name:
            continue
        if isinstance(S, tuple):
            D_, H_, W_ = S
        else:
            D_ = H_ = W_ = S
        x = torch.randn((N, D_, H_, W_, ci), device=dev).to(dt)
        gy = torch.randn((N, D_, H_, W_, co), device=dev).to(dt)
        w = (torch.randn((co, ci, k, k, k), device=dev) * (ci * k ** 3) ** -0.5).requires_grad_(True)
        if a.zeros:
            x.zero_(); gy.zero_(); w.data.zero_()
        fl = 2.0 * N * D_ * H_ * W_ * ci * co * k ** 3
        line = f"{name:28s}"
        if a.what in ("fwd", "all"):
            with torch.no_grad():
                ms = timeit(lambda: ops.conv3d(x, w, None, 1, k // 2, 1))
            line += f"  fwd {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF"
        if a.what in ("wgrad", "all"):
            def wg():
                gw = torch.zeros(w.shape, dtype=torch.float32, device=dev)
                taps = k ** 3
                wse = _lib.lib().dp_conv3d_wgrad_tiled_ws_elems(ci, co, k, 1, k // 2, 1, 1, W_)
                ws = ops._zero_scratch(dev, wse)          # zero-in / zero-out scratch contract (dp_scratch_contract)
                _lib.call("dp_conv3d_wgrad_tiled", x.data_ptr(), ci, gy.data_ptr(), co, gw.data_ptr(), ws.data_ptr(), N, D_, H_, W_,
                          ci, co, k, ci * taps, taps, 1, 1 if dt == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream)
            ms = timeit(wg)
            line += f"  wgrad {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
