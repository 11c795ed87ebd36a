#!/usr/bin/env python3
"""Which Python lines of the step still launch ATen kernels (fills, adds, copies)?  One bf16 DOSE-PYFER training step under torch.profiler
with stacks; every CUDA kernel that is not one of this library's (name not starting with k_ / void k_) is attributed to the innermost
frame inside this repository.     python tools/aten_sites.py [mode]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dose_prediction_amd import losses, synth  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--dtype", mode]
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
nk = 0
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.kernels:
        ks = [k for k in ev.kernels if "k_" not in k.name.split("<")[0]]
        if not ks:
            continue
        frame = "?"
        for fr in (ev.stack or []):
            if ROOT in fr or "dose_prediction_amd" in fr or "bench.py" in fr or "tools/" in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        else:
            frame = (ev.stack or ["?"])[0] if ev.stack else "(autograd engine)"
        if frame == "(autograd engine)" or frame == "?":
            frame += " shapes " + str(ev.input_shapes)[:110]
        sites[(ev.name, frame[:170])] += len(ks)
        nk += len(ks)
print(f"{nk} ATen kernel launches in one {mode} step")
for (name, frame), n in sites.most_common(60):
    print(f"{n:4d}  {name:28s} {frame}")
