#!/usr/bin/env python3
"""Which dp_split_rows launches remain in one fp32x3 training step of DOSE-PYFER at 2 x 128^3 (1.64 ms, 83 launches in round 5)?  Logs every
ops.split_rows call with shape, parts and the two innermost callers."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import losses, ops, synth  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import Model  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402

dose_prediction_amd.set_compute_dtype("fp32x3")
dev = torch.device("cuda:0")
S = (128, 128, 128)
net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8, num_heads=6, act="mish", mode_multi_dec=True, multiS_conv=True)
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
net.to(dev).train()
x, gt = synth.dose_input(2, S).to(dev), synth.dose_target(2, S).to(dev)
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, amsgrad=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    opt.step()


step(); step()
torch.cuda.synchronize()
log = collections.Counter()
orig = ops.split_rows


def spy(a, ca, b, cb, cp, parts, pattern):
    fr = traceback.extract_stack(limit=5)[:-1]
    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(fr[-3:]))
    mb = a.numel() * 4 / 1e6 + (0 if b is None else b.numel() * 4 / 1e6)
    log[(tuple(a.shape), None if b is None else tuple(b.shape), ca, cb, cp, parts, where)] += 1
    spy.mb += mb
    return orig(a, ca, b, cb, cp, parts, pattern)


spy.mb = 0.0
ops.split_rows = spy
with torch.autograd.set_multithreading_enabled(False):
    step()
torch.cuda.synchronize()
print(f"{sum(log.values())} split_rows calls per step, {spy.mb:.0f} MB of fp32 rows read")
for k, n in sorted(log.items(), key=lambda kv: -(kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3] * kv[0][0][4] if len(kv[0][0]) == 5 else 0)):
    print(n, k)
