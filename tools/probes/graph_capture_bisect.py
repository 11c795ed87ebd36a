#!/usr/bin/env python3
"""Which part of the step makes hipStreamEndCapture die when the 3x3x3 branch stream is kept inside a capture?
One configuration per PROCESS (a segfault ends it): DOSE-PYFER at 64^3 (same stream structure as 128^3, seconds instead of minutes).
    python tools/probes/graph_capture_bisect.py <phase> [size]      phase: fwd | fwd_nograd | fwdbwd | step
Stream switches through the environment: DOSE_HIP_CAPTURE_BRANCH=1 (keep the branch stream), DOSE_HIP_WGRAD_STREAM=0, DOSE_HIP_SIDE_STREAM=0
(transformer stream), DOSE_HIP_BRANCH_STREAM=0.  Prints "CAPTURE OK <nodes?>" and replays twice, or dies."""
import faulthandler
import os
import sys

import torch

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import losses, synth  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import Model  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "step"
S = (int(sys.argv[2]),) * 3 if len(sys.argv) > 2 else (64, 64, 64)
dose_prediction_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8, num_heads=6, act="mish",
            mode_multi_dec=True, multiS_conv=True)
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
net.to(dev).train()
x, gt = synth.dose_input(1, S).to(dev), synth.dose_target(1, S).to(dev)
params = [p for p in net.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, amsgrad=True, capturable=True)


def step():
    if phase == "fwd_nograd":
        with torch.no_grad():
            return net(x)[1][0].float().mean()
    out = net(x)
    loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
    if phase == "fwd":
        return loss
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if phase == "step":
        opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm-up done; capturing", phase, "branch stream in capture:", dose_prediction_amd.config.branch_stream_allowed.__doc__ is not None and os.environ.get("DOSE_HIP_CAPTURE_BRANCH", "0"), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    loss = step()
print("CAPTURE OK", flush=True)
g.replay(); g.replay()
torch.cuda.synchronize()
print("REPLAY OK loss", float(loss), flush=True)
