#!/usr/bin/env python3
"""Micro-benchmark of the normalisation row streams at the 128^3 level (HIP events): GB/s of algorithmic traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dose_prediction_amd import ops  # noqa: E402


def timeit(fn, iters=50):
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
    dev = torch.device("cuda:0")
    for C, act in ((16, "relu"), (16, "mish"), (32, "relu")):
        x = torch.randn((2, 128, 128, 128, C), device=dev).bfloat16().requires_grad_(True)
        nbytes = x.numel() * 2
        with torch.no_grad():
            ms = timeit(lambda: ops.norm_act(x, "instance", act=act))
        y = ops.norm_act(x, "instance", act=act)
        g = torch.randn_like(y)
        def bwd():
            x.grad = None
            y.backward(g, retain_graph=True)
        msb = timeit(bwd)
        print(f"C={C} {act}: fwd (stats + apply) {ms * 1e3:7.1f} us = {3 * nbytes / ms / 1e6:6.0f} GB/s (2 reads + 1 write);  "
              f"bwd (partial + apply) {msb * 1e3:7.1f} us = {5 * nbytes / msb / 1e6:6.0f} GB/s (4 reads + 1 write)")


if __name__ == "__main__":
    main()
