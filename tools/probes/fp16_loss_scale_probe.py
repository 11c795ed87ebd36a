#!/usr/bin/env python3
"""fp16 loss-scale probe (BASELINE.json configs[4]): gradients of one DOSE-PYFER GenLoss step in fp16 storage for several static
loss scales, against the fp32-mode gradients of the same step.  Reports the global relative L2 error of the parameter-gradient
vector, non-finite counts, and how much of the boundary gradient (10 / #masked voxels per element) lands in fp16's subnormal range."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import losses, synth  # noqa: E402
from dose_prediction_amd.models import dose_pyfer  # noqa: E402

S = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 64, 64)
dev = torch.device("cuda:0")
torch.manual_seed(4321)
net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8, num_heads=6,
                       act="mish").to(dev).train()
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
x, gt = synth.dose_input(1, S).to(dev), synth.dose_target(1, S).to(dev)
sd = {k: v.clone() for k, v in net.state_dict().items()}


def grads(dtype, scale):
    dose_prediction_amd.set_compute_dtype(dtype)
    dose_prediction_amd.set_loss_scale(scale)
    net.load_state_dict(sd)
    net.zero_grad(set_to_none=True)
    loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    out = {n: p.grad.detach().double() / scale for n, p in net.named_parameters() if p.grad is not None}
    return float(loss), out


l32, g32 = grads(torch.float32, 1.0)
keys = list(g32)
ref = torch.cat([g32[k].reshape(-1) for k in keys])
nmask = float((gt[:, 1:2] > 0).sum())
print(f"volume {S}: loss {l32:.5f}; boundary gradient per masked voxel = 10/{nmask:.0f} = {10 / nmask:.3e} "
      f"(fp16 min normal 6.10e-05, min subnormal 5.96e-08)")
for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    for scale in ((1.0,) if name == "bf16" else (1.0, 64.0, 1024.0, 16384.0, 262144.0)):
        l, g = grads(dt, scale)
        v = torch.cat([g[k].reshape(-1) for k in keys])
        bad = int((~torch.isfinite(v)).sum())
        v = torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)
        worst = max(((g[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)).item() for k in keys if g32[k].norm() > 1e-3 * ref.norm() / len(keys) ** 0.5)
        print(f"{name} loss-scale {scale:>9.0f}: loss {l:.5f}  grad rel-L2 err vs fp32 {((v - ref).norm() / ref.norm()).item():.3e}  "
              f"worst tensor {worst:.3e}  non-finite {bad}  exactly-zero fraction {float((v == 0).double().mean()):.4f} (fp32: {float((ref == 0).double().mean()):.4f})")
dose_prediction_amd.set_compute_dtype(torch.float32)
dose_prediction_amd.set_loss_scale(1.0)
