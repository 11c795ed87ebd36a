#!/usr/bin/env python3
"""Which parameter gradients differ between the two reduction modes in the fp32x3 mode on the golden networks (round 6: the atomic-mode
gates showed a SYSTEMATIC offset on g7_subset/multi and g1 -- median == max over 200 passes -- where the deterministic mode passes at 1e-2).
Prints, per mode, the five worst parameters (error against the golden gradient with the gate's floor) and the distance between the modes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
from helpers import load_golden, pcg_state_dict, sub  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402
from dose_prediction_amd.models.c3d import BaseUNet  # noqa: E402

dev = torch.device("cuda:0")


def errs(named, gold):
    norms = sorted(float(g.double().norm()) for g in gold.values())
    floor = 5e-2 * norms[len(norms) // 2]
    out = {}
    for k, g in gold.items():
        ours = named[k].grad.detach().cpu().reshape(-1)[: g.numel()].double()
        out[k] = (float((ours - g.reshape(-1).double()).norm()) / max(float(g.double().norm()), floor), float(g.double().norm()), floor, tuple(named[k].shape))
    return out


def run(case, det, mode):
    dose_prediction_amd.set_compute_dtype(mode)
    import contextlib
    if isinstance(det, bool):
        cm = dose_prediction_amd.config.deterministic_as(det)
    else:
        dose_prediction_amd.config.set_deterministic(det)
        cm = contextlib.nullcontext()
    with cm:
        if case == "subset":
            g = load_golden("g7_subset_multi")
            net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8, act="mish",
                                  mode_multi_dec=True, multiS_conv=True)
            net.load_state_dict(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
            net.to(dev).train()
            x = g["x"].to(dev).requires_grad_(True)
            outs = net(x)
            torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
        else:
            g = load_golden("g1_base_unet")
            net = BaseUNet(3, [-1, 4, 8, 8, 16, 16])
            net.load_state_dict(sub(g, "sd"))
            net.to(dev).train()
            x = g["x"].to(dev).requires_grad_(True)
            net(x).backward(g["r"].to(dev))
        torch.cuda.synchronize()
    named = dict(net.named_parameters())
    return errs(named, sub(g, "grad")), {k: p.grad.detach().clone() for k, p in named.items() if p.grad is not None}


CASES = os.environ.get("PROBE_CASES", "subset,g1").split(",")
MODES = [{"fp32x3": "fp32x3", "fp32": torch.float32}[m] for m in os.environ.get("PROBE_MODES", "fp32x3,fp32").split(",")]
for case in CASES:
    for mode in MODES:
        res = {}
        dets = (True, False)
        if os.environ.get("PROBE_MASK"):          # a mask of deterministic SITES instead of the full switch, atomic everywhere else (config.set_deterministic)
            dets = (int(os.environ["PROBE_MASK"]),)
        if os.environ.get("PROBE_ATOMIC_ONLY"):
            dets = (False,)
        for det in dets:
            e, grads = run(case, det, mode)
            res[det] = grads
            top = sorted(e.items(), key=lambda kv: -kv[1][0])[:5]
            print(f"{case} {mode} {('det' if det else 'atomic') if isinstance(det, bool) else 'mask %d' % det}: " + "; ".join(f"{k} {v[0]:.2e} (|g| {v[1]:.2e}, floor {v[2]:.2e}, {v[3]})" for k, v in top))
        if True in res and False in res:
            d = sorted(((float((res[True][k].double() - res[False][k].double()).norm() / max(float(res[True][k].double().norm()), 1e-30)), k) for k in res[True]), reverse=True)[:5]
            print(f"   det vs atomic, relative: " + "; ".join(f"{k} {v:.2e}" for v, k in d))
dose_prediction_amd.set_compute_dtype(torch.float32)
