#!/usr/bin/env python3
"""Where the fp32x3 step's non-matrix-core leftovers come from: every C-ABI call of the named entry points in one DOSE-PYFER training
step (side streams off), timed with HIP events and attributed to the innermost calling line of this package plus the module line above it.
    python tools/x3_leftover_sites.py [entry-point-substring ...]       default: split_rows pointwise gemm_nt conv3d( generic )"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dose_prediction_amd import _lib, losses, synth  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402

pats = [a for a in sys.argv[1:] if not a.startswith("--")] or ["dp_split_rows", "dp_pointwise", "dp_gemm_nt", "dp_conv3d", "dp_act", "dp_add", "dp_cat"]
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--dtype", "fp32x3", "--no-side-stream"]
args = bench.parse()
dev = torch.device("cuda:0")
shape = (128, 128, 128)
net = bench.build_model(args, shape, dev)
params = [p for p in net.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)
x, gt = synth.dose_input(2, shape).to(dev), synth.dose_target(2, shape).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
log = []
orig = _lib.call


def traced(name, *a):
    if not any(p in name for p in pats):
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = orig(name, *a)
    e1.record()
    fr = [f for f in traceback.extract_stack()[:-1] if "dose_prediction_amd" in f.filename]
    inner = fr[-1] if fr else None
    outer = next((f for f in reversed(fr) if "/ops.py" not in f.filename), None)
    ints = [int(v) for v in a if isinstance(v, int) and 0 < v < (1 << 40)][:8]
    log.append((name, f"{os.path.basename(inner.filename)}:{inner.lineno}" if inner else "?",
                f"{os.path.basename(outer.filename)}:{outer.lineno}" if outer else "(autograd)", ints, e0, e1))
    return rc


_lib.call = traced
from dose_prediction_amd import ops  # noqa: E402
step()
torch.cuda.synchronize()
_lib.call = orig
agg = collections.defaultdict(lambda: [0, 0.0, None])
for name, inner, outer, ints, e0, e1 in log:
    k = (name, inner, outer)
    agg[k][0] += 1
    agg[k][1] += e0.elapsed_time(e1)
    agg[k][2] = ints
tot = collections.Counter()
for (name, inner, outer), (n, ms, ints) in agg.items():
    tot[name] += ms
print("per entry point (ms of event time, includes launch gaps for tiny kernels):")
for name, ms in tot.most_common():
    print(f"  {ms:7.3f} ms  {name}")
print()
for (name, inner, outer), (n, ms, ints) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{ms:7.3f} ms {n:3d}x  {name:26s} {inner:14s} <- {outer:22s} {ints}")
