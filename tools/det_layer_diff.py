"""Where does the run-to-run difference of the G7 subset network's backward pass (deterministic switch OFF) come from?  Two passes are
recorded module by module (forward outputs through forward hooks, gradient of every module output through tensor hooks) and compared in
execution order: the first tensor that differs, and how the relative difference grows from there.   python tools/det_layer_diff.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
from helpers import load_golden, pcg_state_dict, rel_l2  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402

MODE = os.environ.get("PROBE_MODE", "fp32x3")
dev = torch.device("cuda:0")
g = load_golden("g7_subset_multi")
sd = pcg_state_dict(g["keys"], g["shapes"], g["seed"])
dose_prediction_amd.set_compute_dtype(MODE)
if os.environ.get("PROBE_TERMS") == "3":
    c = dose_prediction_amd.config
    c.set_x3_dgrad_terms(3); c.set_x3_wgrad_terms(3); c.set_x3_linear_wgrad_terms(3)
net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8,
                      act="mish", mode_multi_dec=True, multiS_conv=True)
net.load_state_dict(sd)
net = net.to(dev).train()
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
names = {m: n for n, m in net.named_modules()}
rec = None


def fhook(m, inp, out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    for j, o in enumerate(outs):
        if torch.is_tensor(o) and o.dtype.is_floating_point:
            key = f"{names[m]}#{j}"
            rec["f"].append((key, o.detach().float().clone()))
            if o.requires_grad:
                o.register_hook(lambda gr, key=key: rec["b"].append((key, gr.detach().float().clone())))


for m in net.modules():
    if len(list(m.children())) == 0 or type(m).__name__ in ("conv_3_1", "UnetResBlock", "TransformerBlock", "ModifiedUnetrUpBlock"):
        m.register_forward_hook(fhook)


def one():
    global rec
    rec = {"f": [], "b": []}
    net.load_state_dict(sd0)
    net.zero_grad(set_to_none=True)
    x = g["x"].to(dev).requires_grad_(True)
    outs = net(x)
    torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(4)])
    torch.cuda.synchronize()
    return x.grad.clone(), rec


dx0, r0 = one()
for it in range(200):
    dx1, r1 = one()
    d = rel_l2(dx1.cpu(), dx0.cpu())
    if d > float(os.environ.get("PROBE_THRESH", "1e-4")):
        break
print(f"mode {MODE}: pass {it + 1} differs from pass 0 in d/dx by {d:.3e}")
print("forward (execution order), relative L2 difference of module outputs; first 40 non-zero:")
n = 0
for (k0, a), (k1, b) in zip(r0["f"], r1["f"]):
    assert k0 == k1
    e = rel_l2(b.cpu(), a.cpu()) if a.shape == b.shape else float("nan")
    if e > 0 and n < 40:
        print(f"  fwd {k0:70s} {tuple(a.shape)}  {e:.3e}  max|a| {float(a.abs().max()):.3e}")
        n += 1
print("backward (execution order), relative L2 difference of the gradient of module outputs; first 60 non-zero:")
n = 0
for (k0, a), (k1, b) in zip(r0["b"], r1["b"]):
    if k0 != k1:
        print("  (hook order differs)", k0, k1)
        break
    e = rel_l2(b.cpu(), a.cpu()) if a.shape == b.shape else float("nan")
    if e > 0 and n < 60:
        print(f"  bwd {k0:70s} {tuple(a.shape)}  {e:.3e}")
        n += 1
