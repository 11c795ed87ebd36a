#!/usr/bin/env python3
"""Split-K Linear of the patch-embedding class in the fp32x3 mode: atomic (default) against the fixed-order path and float64, in several
shapes incl. the G7 subset network's (4 token rows, K = 20480, 48 outputs, splitk 32) -- round 6: the process-dependent offset of the
atomic-mode golden gradients disappears with dp_set_deterministic(2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dose_prediction_amd.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "fp32x3")
g = torch.Generator().manual_seed(0)
# warm the allocator with junk so that fresh blocks are not zero pages
junk = [torch.full((1 << 22,), 3.0e3, device=dev) for _ in range(8)]
del junk
for rows, K, nout, splitk in ((4, 20480, 48, 32), (16, 20480, 48, 32), (1024, 102400, 768, 7), (4, 20480, 48, 8), (4, 4096, 48, 4)):
    x = torch.randn((1, rows, K), generator=g)
    w = torch.randn((nout, K), generator=g) * K ** -0.5
    b = 0.1 * torch.randn((nout,), generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    out = {}
    for det in (False, True, False):
        with dose_prediction_amd.config.deterministic_as(det):
            xd = x.to(dev).requires_grad_(True)
            wd = w.to(dev).requires_grad_(True)
            y = ops.linear(xd, wd, b.to(dev), splitk=splitk)
            y.sum().backward()
            torch.cuda.synchronize()
        e = float((y.detach().double().cpu() - ref).abs().max() / ref.abs().max())
        gw = float((wd.grad.double().cpu() - x.double().sum(1).expand(nout, K)).norm() / x.double().sum(1).expand(nout, K).norm())
        print(f"rows {rows} K {K} nout {nout} splitk {splitk} {'det   ' if det else 'atomic'}: forward max-rel {e:.2e}   dW rel-L2 {gw:.2e}", flush=True)
