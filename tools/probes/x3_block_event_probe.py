"""Rare discrete events in the multi-scale block (blocks.conv_3_1: 3^3 || 7^3 branches -> normalise into the mixer's input -> 1^3
conv -> IN -> act) in the fp32x3 mode: forward + backward repeated on fixed data; every gradient compared with the first repetition's.
    python tools/probes/x3_block_event_probe.py [repetitions]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import blocks  # noqa: E402

c = dose_prediction_amd.config
if not os.environ.get("ONE"):
    c.set_x3_dgrad_terms(3); c.set_x3_wgrad_terms(3); c.set_x3_linear_wgrad_terms(3)
dose_prediction_amd.set_compute_dtype(os.environ.get("MODE", "fp32x3"))
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
torch.manual_seed(5)
for (ca, cb, cout, shape, act) in [(4, 4, 4, (2, 32, 16, 16), "mish"), (8, 0, 4, (2, 32, 16, 16), "mish"), (8, 8, 8, (2, 16, 8, 16), "relu"), (16, 16, 16, (1, 16, 16, 32), "mish")]:
    blk = blocks.conv_3_1(ca + cb, cout, act).to(dev).train()
    a = torch.randn(shape + (ca,), device=dev)
    b = torch.randn(shape + (cb,), device=dev) if cb else None
    r = torch.randn(shape + (cout,), device=dev)
    ref, ev, worst, names = None, 0, 0.0, set()
    for it in range(reps):
        blk.zero_grad(set_to_none=True)
        aa = a.clone().requires_grad_(True)
        bb = b.clone().requires_grad_(True) if cb else None
        y = blk((aa, bb) if cb else aa)
        y.backward(r)
        cur = {"y": y.detach(), "ga": aa.grad}
        if cb:
            cur["gb"] = bb.grad
        cur.update({k: p.grad for k, p in blk.named_parameters() if p.grad is not None})
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            continue
        hit = False
        for k in cur:
            d = float((cur[k] - ref[k]).norm() / (ref[k].norm() + 1e-30))
            if d > 1e-5 and ref[k].norm() > 1e-6:
                hit = True
                names.add(k)
                worst = max(worst, d)
        ev += hit
    print(f"conv_3_1 {ca}+{cb}->{cout} {shape} {act}: events {ev} of {reps - 1}; largest difference {worst:.1e}; tensors {sorted(names)[:8]}")
