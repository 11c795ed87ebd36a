#!/usr/bin/env python3
"""Is the transformer branch on the critical path of the DOSE-PYFER step?  Un-profiled measurements (HIP events, no rocprof):
 (1) duration of the ViT forward on its side stream, and how long the caller's stream then WAITS for it in _SideRun.final;
 (2) step time with an extra spin kernel of `ms` milliseconds put on the side stream in front of the ViT forward / the ViT backward:
     a branch that is hidden behind the main stream absorbs the delay, a branch on the critical path passes it on one to one.
     python tools/probes/vit_exposure_probe.py [dtype]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dose_prediction_amd import blocks, losses, synth  # noqa: E402
from dose_prediction_amd.models import dose_pyfer  # noqa: E402
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

# spin cycles per millisecond (torch.cuda._sleep counts device clock cycles)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
CYC_PER_MS = 20_000_000 / e0.elapsed_time(e1)

EV = {"vit": [], "wait": []}
DELAY = {"fwd": 0.0, "bwd": 0.0}
vit_forward = blocks.ViT.forward
side_final = dose_pyfer._SideRun.final


def forward_probe(self, xin, events=None):
    if DELAY["fwd"]:
        torch.cuda._sleep(int(DELAY["fwd"] * CYC_PER_MS))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    z, hidden = vit_forward(self, xin, events)
    b.record()
    EV["vit"].append((a, b))
    if DELAY["bwd"] and z.requires_grad:
        side = torch.cuda.current_stream()

        def hook(g):
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(DELAY["bwd"] * CYC_PER_MS))
            return g
        z.register_hook(hook)
    return z, hidden


def final_probe(self, *outs):
    if self.side is None:
        return side_final(self, *outs)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(self.main)
    r = side_final(self, *outs)
    b.record(self.main)
    EV["wait"].append((a, b))
    return r


blocks.ViT.forward = forward_probe
dose_pyfer._SideRun.final = final_probe


def step():
    opt.zero_grad(set_to_none=True)
    loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    opt.step()


def timed(n=20):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    EV["vit"].clear(); EV["wait"].clear()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    vit = sum(a.elapsed_time(b) for a, b in EV["vit"]) / max(1, len(EV["vit"]))
    wait = sum(a.elapsed_time(b) for a, b in EV["wait"]) / max(1, len(EV["wait"]))
    return ms, vit, wait


print(f"{mode}: spin calibration {CYC_PER_MS / 1e6:.2f} Mcycles per ms")
base = timed()
print(f"no delay          : step {base[0]:6.2f} ms   ViT forward on its stream {base[1]:5.2f} ms   caller's stream waits {base[2]:5.2f} ms in final()")
for where in ("fwd", "bwd"):
    for ms in (0.5, 1.0, 2.0):
        DELAY["fwd"] = DELAY["bwd"] = 0.0
        DELAY[where] = ms
        r = timed()
        print(f"+{ms:3.1f} ms before ViT {where}: step {r[0]:6.2f} ms (+{r[0] - base[0]:5.2f})   ViT forward {r[1]:5.2f} ms   wait {r[2]:5.2f} ms")
DELAY["fwd"] = DELAY["bwd"] = 0.0
r = timed()
print(f"no delay again    : step {r[0]:6.2f} ms")
