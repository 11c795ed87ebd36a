#!/usr/bin/env python3
"""Where does the backward pass end on each stream?  How long does the caller's stream sit idle at the end of the backward pass waiting for the transformer branch (second stream)?
Runs DOSE-PYFER bench steps and prints, per step, elapsed(main reached the join, ViT stream reached the join)."""
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dose_prediction_amd
from dose_prediction_amd import ops, losses
from dose_prediction_amd.models import dose_pyfer
from dose_prediction_amd.optim import FusedAdam

dose_prediction_amd.set_compute_dtype("bf16")
dev = torch.device("cuda")
torch.manual_seed(0)
net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=(128, 128, 128),
                       num_layers=8, num_heads=6, act="mish", mode_multi_dec=True, multiS_conv=True)
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
net.to(dev).train()
if len(sys.argv) > 1 and sys.argv[1] == "ddp":
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29656")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    red = attach_gradient_allreduce(net)
x = torch.randn(2, 9, 128, 128, 128, device=dev)
gt = torch.cat((torch.rand(2, 1, 128, 128, 128, device=dev), (torch.rand(2, 1, 128, 128, 128, device=dev) > 0.3).float()), 1)
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
from dose_prediction_amd import streams
marks, tails, state = [], [], {"armed": False}
_orig = (ops.flush_deferred, ops.join_wgrad_stream)


def _probe():
    if state["armed"] and torch.autograd.Variable._execution_engine is not None:
        state["armed"] = False
        main = torch.cuda.current_stream()
        evs = []
        for s in [main] + list(streams.side_streams(dev, streams.root(main))):
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            evs.append(e)
        tails.append(evs)


def _w0(*a):
    _probe()
    return _orig[0](*a)


def _w1(*a):
    _probe()
    return _orig[1](*a)


ops.flush_deferred, ops.join_wgrad_stream = _w0, _w1
for i in range(12):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    opt.zero_grad(set_to_none=True)
    out = net(x)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
    state["armed"] = i >= 4
    loss.backward()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    opt.step()
    e3 = torch.cuda.Event(enable_timing=True); e3.record()
    marks.append((e0, e1, e2, e3))
torch.cuda.synchronize()
for (e0, e1, e2, e3) in marks[4:]:
    print(f"forward {e0.elapsed_time(e1):6.2f}  backward {e1.elapsed_time(e2):6.2f}  adam+packs {e2.elapsed_time(e3):5.2f} ms")
for (e0, e1, e2, e3), evs in zip(marks[4:], tails):
    print("since the start of the backward pass: " + "  ".join(f"{n} done {e1.elapsed_time(e):6.2f}" for n, e in zip(("main", "vit", "branch", "wgrad"), evs))
          + f"  joined {e1.elapsed_time(e2):6.2f} ms")
