#!/usr/bin/env python3
"""Host-side cost of one DOSE-PYFER training step (the launch thread must stay ahead of the GPU): cProfile over 10 steps, top functions
by own time, with and without the data-parallel reducer (1-rank RCCL).  The autograd engine walks a CUDA graph on a worker thread of
its own, which cProfile does not see (the whole backward pass shows up as run_backward's own time): the second table profiles the same steps
with torch.autograd.set_multithreading_enabled(False), i.e. the backward functions on the calling thread.
usage: tools/host_profile.py [ddp]"""
import cProfile
import os
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
import dose_prediction_amd
from dose_prediction_amd import losses
from dose_prediction_amd.models import dose_pyfer
from dose_prediction_amd.optim import FusedAdam

ddp = len(sys.argv) > 1 and sys.argv[1] == "ddp"
dose_prediction_amd.set_compute_dtype("bf16")
dev = torch.device("cuda")
torch.manual_seed(0)
net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=(128, 128, 128),
                       num_layers=8, num_heads=6, act="mish", mode_multi_dec=True, multiS_conv=True)
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
net.to(dev).train()
if ddp:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    red = attach_gradient_allreduce(net)
x = torch.randn(2, 9, 128, 128, 128, device=dev)
gt = torch.cat((torch.rand(2, 1, 128, 128, 128, device=dev), (torch.rand(2, 1, 128, 128, 128, device=dev) > 0.3).float()), 1)
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)


PH = []


def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = net(x)
    t1 = time.perf_counter()
    loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    PH.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))


for _ in range(5):
    step()
torch.cuda.synchronize()
# host time per step with the GPU out of the way: enqueue one step, wait, repeat
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
print("host enqueue time per step (GPU idle at the start of each): " + " ".join(f"{t:.1f}" for t in ts) + " ms")
for ph in PH[-6:]:
    print("   forward %.2f  loss %.2f  backward %.2f  optimizer %.2f ms" % tuple(1e3 * v for v in ph))
if ddp:
    red.host_s.clear()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print("reducer host time per step (hook includes the launches it triggers): " + ", ".join(f"{k} {100 * v:.2f} ms" for k, v in red.host_s.items()),
          f"; buckets {len(red.buckets)}, parameters {len(red.params)}")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
print("\n==== the same with the backward pass on the calling thread (torch.autograd.set_multithreading_enabled(False)) ====")
from dose_prediction_amd import _lib
print("binding:", _lib.BINDING)
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print("host enqueue time per step, single-threaded backward: " + " ".join(f"{t:.1f}" for t in ts) + " ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(60)
st.sort_stats("cumtime").print_stats(60)
