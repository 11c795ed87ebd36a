#!/usr/bin/env python3
"""What would running the optimizer step beside the frozen net_A forward of the NEXT step give?  (net_A's weights are frozen in the
benchmarked configuration, so its forward does not depend on the update.)  The Adam launch goes to its own stream; the caller's
stream waits for it in a forward pre-hook of net_B.     python tools/probes/adam_overlap_probe.py [dtype]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
side = torch.cuda.Stream()
state = {"overlap": False, "ev": None, "held": None}


def pre_hook(mod, inp):
    if state["ev"] is not None:
        torch.cuda.current_stream().wait_event(state["ev"])
        state["ev"] = None


net.net_B.register_forward_pre_hook(pre_hook)


def step():
    opt.zero_grad(set_to_none=True)
    loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    if state["overlap"]:
        main = torch.cuda.current_stream()
        state["held"] = [p.grad for p in params]          # (the gradients must outlive zero_grad(set_to_none=True) of the next step)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            opt.step()
            ev = torch.cuda.Event()
            ev.record(side)
        state["ev"] = ev
    else:
        opt.step()


def timed(n=30):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(3):
    state["overlap"] = False
    a = timed()
    state["overlap"] = True
    b = timed()
    print(f"{mode}: optimizer on the caller's stream {a:6.2f} ms/step   beside the next net_A forward {b:6.2f} ms/step   ({b - a:+.2f})")
