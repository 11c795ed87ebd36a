"""Uninitialised-memory hunt: before every pass the caching allocator's free blocks are filled with NaN (a big tensor is filled and
released), a fresh model is built and run forward + backward; any kernel that reads memory nobody wrote turns its output into NaN.
Reports the first leaf module whose output is not finite and every parameter / input gradient that is not.
    python tools/probes/poison_probe.py [mode] [passes] [multi|plain|pyfer]"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tag = sys.argv[3] if len(sys.argv) > 3 else "multi"
if os.environ.get("TERMS3"):
    c = dose_prediction_amd.config
    c.set_x3_dgrad_terms(3); c.set_x3_wgrad_terms(3); c.set_x3_linear_wgrad_terms(3)
dose_prediction_amd.set_compute_dtype(mode)
dev = torch.device("cuda:0")
g = load_golden(f"g7_subset_{tag}")
kw = dict(mode_multi_dec=True, multiS_conv=True) if tag == "multi" else dict(mode_multi_dec=False)
for r in range(reps):
    if not os.environ.get("NOPOISON"):
        poison = torch.full((1 << 28,), float("nan"), device=dev)        # 1 GiB of NaN back into the allocator's free list
        del poison
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8,
                          act="mish", **kw)
    M._load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
    bad = []
    rec = []
    hooks = [m.register_forward_hook(lambda mod, inp, out, name=name: bad.append(name) if torch.is_tensor(out) and not bool(torch.isfinite(out.float()).all()) else None)
             for name, m in net.named_modules() if not list(m.children())]
    if os.environ.get("COMPARE"):
        hooks += [m.register_forward_hook(lambda mod, inp, out, name=name: rec.append((name, out.detach().float().clone())) if torch.is_tensor(out) else None)
                  for name, m in net.named_modules() if not list(m.children())]
    x = g["x"].to(dev).requires_grad_(True)
    outs = net(x)
    torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    gbad = [k for k, p in net.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if not bool(torch.isfinite(x.grad).all()):
        gbad.append("d/dx")
    if os.environ.get("COMPARE"):
        if r == 0:
            first_rec = rec
        else:
            dif = [(n0, float((a - b).abs().max()), float(a.abs().max())) for (n0, a), (n1, b) in zip(first_rec, rec) if a.shape == b.shape and not torch.equal(a, b)]
            if dif:
                print(f"pass {r}: {len(dif)} forward tensors differ from pass 0; first three: {dif[:3]}")
    if bad or gbad:
        print(f"pass {r}: forward NaN first at {bad[:2]} ({len(bad)} modules); non-finite gradients: {len(gbad)} {gbad[:4]}")
print("done", reps, "passes")
