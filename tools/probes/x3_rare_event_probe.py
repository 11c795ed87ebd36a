"""Rare discrete events inside single fp32x3 convolutions: the same conv3d forward + backward repeated many times; a result further
than 1e-5 (relative L2) from the first repetition's is an event (atomics reorder sums at the 1e-7 level only).
    python tools/probes/x3_rare_event_probe.py [repetitions]      (terms from the DOSE_HIP_X3_* variables; default here: three everywhere)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import ops  # noqa: E402

c = dose_prediction_amd.config
if not os.environ.get("ONE"):
    c.set_x3_dgrad_terms(3); c.set_x3_wgrad_terms(3); c.set_x3_linear_wgrad_terms(3)
dose_prediction_amd.set_compute_dtype(os.environ.get("MODE", "fp32x3"))
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = torch.Generator().manual_seed(3)
shapes = [(2, 8, 4, 32, 16, 16, 7), (2, 8, 4, 32, 16, 16, 3), (2, 4, 4, 32, 16, 16, 7), (2, 4, 4, 32, 16, 16, 3), (2, 5, 4, 32, 16, 16, 3), (2, 16, 8, 16, 8, 16, 7),
          (1, 16, 16, 16, 16, 32, 7), (1, 32, 32, 8, 16, 16, 3)]
for (N, Cin, Cout, D, H, W, k) in shapes:
    x = torch.randn((N, D, H, W, Cin), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, k, k, k), generator=g) * (Cin * k ** 3) ** -0.5).to(dev)
    r = torch.randn((N, D, H, W, Cout), generator=g).to(dev)
    ref = None
    ev = [0, 0, 0]
    worst = [0.0, 0.0, 0.0]
    for it in range(reps):
        xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.conv3d(xa, wa, None, 1, k // 2, 1)
        y.backward(r)
        cur = [y.detach(), xa.grad, wa.grad]
        if ref is None:
            ref = [t.clone() for t in cur]
            continue
        for i in range(3):
            d = float((cur[i] - ref[i]).norm() / ref[i].norm())
            worst[i] = max(worst[i], d)
            ev[i] += d > 1e-5
    print(f"conv3d {Cin}->{Cout} k{k} {N}x{D}x{H}x{W}: events (y, gx, gw) {ev} of {reps - 1}; largest difference {['%.1e' % v for v in worst]}")
