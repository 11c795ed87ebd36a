"""Run-to-run noise of the gradients of the small golden DOSE-PYFER subset model (two identical plain runs): the floor any
equality test between two runs has to allow (fp32 atomics in the split-K / split-kd kernels, amplified by the normalised layers)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from helpers import load_golden, pcg_state_dict
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
dev = torch.device("cuda:0")
g = load_golden("g7_subset_multi")
def build():
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
    net.load_state_dict(pcg_state_dict(g["keys"], g["shapes"], g["seed"]), strict=True); return net.to(dev).train()
def grads(net):
    out = {}
    for step in range(2):
        net.zero_grad(set_to_none=True)
        sum((o * o).mean() for o in net(g["x"].to(dev) * (1.0 + 0.1 * step))).backward()
        torch.cuda.synchronize()
        out[step] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}
    return out
a, b = grads(build()), grads(build())
for step in a:
    norms = sorted(float(r.double().norm()) for r in a[step].values() if r is not None); floor = 5e-2 * norms[len(norms)//2]
    worst = sorted(((float((a[step][k].double()-b[step][k].double()).norm())/max(float(a[step][k].double().norm()), floor), k) for k in a[step] if a[step][k] is not None), reverse=True)[:4]
    print(step, worst)
