#!/usr/bin/env python3
"""Where does the fp32x3 atomic-mode backward pass of the G7 subset network first leave the deterministic one?  (round 6: process-dependent
4e-2 offsets of the atomic-mode golden gradients that dp_set_deterministic(2) -- split-K GEMMs unsplit -- removes.)  Records the gradient
arriving at every leaf module's output (full backward hooks) under mask 2 and under the default, and prints, in backward order, the modules
whose incoming gradient differs by more than 5e-3, plus the parameter gradients that differ."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
from helpers import load_golden, pcg_state_dict  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402

dev = torch.device("cuda:0")
dose_prediction_amd.set_compute_dtype("fp32x3")
g = load_golden("g7_subset_multi")


def run(mask):
    dose_prediction_amd.config.set_deterministic(mask)
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8, act="mish",
                          mode_multi_dec=True, multiS_conv=True)
    net.load_state_dict(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
    net.to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    outs = net(x)
    # every autograd node of the graph, in a fixed traversal order; a hook on each records the gradients it RECEIVES (grad_outputs) and the
    # order in which the engine runs it
    nodes, seen, stack = [], set(), [o.grad_fn for o in outs if o.grad_fn is not None]
    while stack:
        n = stack.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        nodes.append(n)
        for nf, _ in n.next_functions:
            stack.append(nf)
    rec, order = {}, []
    for i, n in enumerate(nodes):
        def hook(gi, go, i=i, n=n):
            order.append(i)
            rec[i] = (n.name(), [t.detach().float().clone() if torch.is_tensor(t) else None for t in go])
        n.register_hook(hook)
    torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
    torch.cuda.synchronize()
    dose_prediction_amd.config.set_deterministic(False)
    return rec, order, {}, {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


ra, oa, fa, pa = run(2)
rb, ob, fb, pb = run(0)
rc, oc, fc, pc = run(0)
print("gradients RECEIVED by the autograd nodes, in the engine's execution order: mask 2 vs atomic   |   atomic vs atomic (second pass)")
shown = 0
for i in oa:
    if i not in rb or i not in rc:
        continue
    name, ta = ra[i]
    pairs = [(a, b, c) for a, b, c in zip(ta, rb[i][1], rc[i][1]) if a is not None and b is not None and c is not None and a.shape == b.shape == c.shape]
    if not pairs:
        continue
    e = max(rel(b, a) for a, b, c in pairs)
    e2 = max(rel(c, b) for a, b, c in pairs)
    lo, hi = int(os.environ.get("PROBE_FROM", "-1")), int(os.environ.get("PROBE_TO", "-1"))
    if (lo <= oa.index(i) <= hi) or (e > 5e-3 and shown < 30 and lo < 0):
        print(f"  #{oa.index(i):4d} {name:34s} {e:.2e}   |   {e2:.2e}   shapes {[tuple(a.shape) for a, b, c in pairs]}")
        shown += 1
bad = sorted(((rel(pb[k], pa[k]), k) for k in pa), reverse=True)[:8]
print("parameter gradients, mask 2 vs atomic:", "; ".join(f"{k} {v:.1e}" for v, k in bad))
