"""Which module's output differs bit-wise between two forward passes on identical inputs?  (G7 subset network, fp32x3 by default.)
python tools/probes/nondeterminism_probe.py [mode] [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
import test_models_gpu as M  # noqa: E402
from helpers import load_golden, pcg_state_dict  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dose_prediction_amd.set_compute_dtype(mode)
dev = torch.device("cuda:0")
g = load_golden("g7_subset_multi")
net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8, act="mish",
                      mode_multi_dec=True, multiS_conv=True)
M._load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
x = g["x"].to(dev)
runs = []
for r in range(reps):
    rec = []
    hooks = []
    for name, m in net.named_modules():
        if not list(m.children()):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: rec.append((name, out.detach().float().clone() if torch.is_tensor(out) else None))))
    net.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    outs = net(xg)
    torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    rec.append(("d/dx", xg.grad.detach().float().clone()))
    for k, p_ in net.named_parameters():
        if p_.grad is not None:
            rec.append(("grad " + k, p_.grad.detach().float().clone()))
    runs.append(rec)
base = runs[0]
for r in range(1, reps):
    first = None
    ndiff = 0
    for (n0, a), (n1, b) in zip(base, runs[r]):
        if a is None or b is None or a.shape != b.shape:
            continue
        if not torch.equal(a, b):
            ndiff += 1
            if first is None:
                first = (n0, float((a - b).abs().max()), float(a.abs().max()))
    worst = max(((float((a - b).norm() / (a.norm() + 1e-30)), n0) for (n0, a), (n1, b) in zip(base, runs[r]) if a is not None and b is not None and a.shape == b.shape), default=None)
    if os.environ.get("FORWARD_ONLY"):
        fwd = [(n0, float((a - b).abs().max()), float(a.abs().max())) for (n0, a), (n1, b) in zip(base, runs[r])
               if a is not None and b is not None and a.shape == b.shape and not n0.startswith(("grad ", "d/dx")) and not torch.equal(a, b)]
        if fwd:
            print(f"run {r}: {len(fwd)} forward tensors differ from run 0; first three: {fwd[:3]}")
        continue
    dx0, dxr = base[[n for n, _ in base].index("d/dx")][1], runs[r][[n for n, _ in runs[r]].index("d/dx")][1]
    dd = (dx0 - dxr).abs()
    print(f"   d/dx: rel-L2 of the difference {float(dd.norm() / dx0.norm()):.2e}; elements differing by more than 1e-5 of the maximum: "
          f"{int((dd > 1e-5 * dx0.abs().max()).sum())} of {dd.numel()}; by more than 1e-3: {int((dd > 1e-3 * dx0.abs().max()).sum())}")
    print(f"run {r} vs run 0: {ndiff} tensors differ; first: {first}; largest relative difference: {worst}")
print("compared", reps, "runs")
