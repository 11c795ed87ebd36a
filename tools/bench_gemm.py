#!/usr/bin/env python3
"""Micro-benchmark of dp_gemm_nt on the ViT token-matrix shapes of DOSE-PYFER (B=2 x 512 tokens, hidden 768, mlp 3072).
usage: python tools/bench_gemm.py [--dtype bf16|fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dose_prediction_amd import ops  # noqa: E402

# (name, M, N, K, batch, out_f32)
SHAPES = [
    ("proj  1024x768x768", 1024, 768, 768, 1, False), ("qkv   1024x2304x768", 1024, 2304, 768, 1, False),
    ("fc1   1024x3072x768", 1024, 3072, 768, 1, False), ("fc2   1024x768x3072", 1024, 768, 3072, 1, False),
    ("qkv.dgrad 1024x768x2304", 1024, 768, 2304, 1, False), ("crop96 2304x768x768", 2304, 768, 768, 1, False), ("crop96 2304x768x3072", 2304, 768, 3072, 1, False),
    ("M256  256x768x768", 256, 768, 768, 1, False), ("M4096 4096x768x768", 4096, 768, 768, 1, False),
    ("dW    768x768x1024 f32", 768, 768, 1024, 1, True), ("dWfc  3072x768x1024 f32", 3072, 768, 1024, 1, True),
    ("QK^T  12x512x512x128", 512, 512, 128, 12, False), ("PV    12x512x128x512", 512, 128, 512, 12, False),
    ("tconv 65536x256x64", 65536, 256, 64, 1, False), ("tconv 8192x512x128", 8192, 512, 128, 1, False),
]


def timeit(fn, iters=20):
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
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    for name, M, N, K, nb, f32 in SHAPES:
        pad = int(os.environ.get("PAD", "0"))          # extra elements per row: tests address-interleave (channel) effects of the row pitch
        A = torch.randn((nb, M, K + pad), device=dev).to(dt)
        B = torch.randn((nb, N, K + pad), device=dev).to(dt)
        C = torch.empty((nb, M, N), device=dev, dtype=torch.float32 if f32 else dt)
        ms = timeit(lambda: ops.gemm_nt(A, B, C, M=M, N=N, K=K, batch=(nb, 1), sa=(M * (K + pad), 0), sb=(N * (K + pad), 0), sc=(M * N, 0),
                                        lda=K + pad, ldb=K + pad, ldc=N))
        ref = timeit(lambda: torch.matmul(A[:, :, :K], B[:, :, :K].transpose(1, 2))) if not os.environ.get("NO_REF") else float("nan")
        fl = 2.0 * nb * M * N * K
        print(f"{name:28s} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF   (torch.matmul {ref * 1e3:8.1f} us {fl / ref / 1e9:7.1f} TF)", flush=True)


if __name__ == "__main__":
    main()
