"""Bisect harness for the rare (about 1 in 100 passes) discrete event in the fp32x3 backward pass of the G7 subset network: d/dx of
every pass is compared with the first pass's; a relative difference above 1e-3 is an event.  Features are knocked out by environment
variables (REUSE: one model object for all passes; NOSPLITSHARE, NOPRESPLIT, NOSTATS, NOCATNORM, NOGYSPLIT, SYNC: a device
synchronisation after every C-ABI call).     python tools/x3_event_bisect.py [passes]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
import test_models_gpu as M  # noqa: E402
from helpers import load_golden, pcg_state_dict  # noqa: E402
from dose_prediction_amd import ops, _lib  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402

c = dose_prediction_amd.config
c.set_x3_dgrad_terms(3); c.set_x3_wgrad_terms(3); c.set_x3_linear_wgrad_terms(3)
dose_prediction_amd.set_compute_dtype("fp32x3")
E = os.environ.get
if E("NOSPLITSHARE"):
    ops._split_conv_input = lambda xa, ca, xb, cb, cp: ops.split_rows(xa, ca, xb, cb, cp, 2, 0b10)
if E("NOPRESPLIT") or E("NOGYSPLIT"):
    _na = ops.norm_act

    def norm_act(*a, **k):
        k.pop("x3_split_for", None)
        return _na(*a, **k)
    ops.norm_act = norm_act
    import dose_prediction_amd.blocks as B
    B.ops.norm_act = norm_act
if E("NOSTATS"):
    L = _lib.lib()
    L.dp_conv3d_tiled_stat_blocks = lambda *a: 0
if E("NOCATNORM"):
    ops.norm_act_cat = lambda r3, r7, act=None: ops.cat((ops.norm_act(r3, "instance", act=act), ops.norm_act(r7, "instance", act=act)))
if E("SYNC"):
    _oc = _lib.call

    def call(name, *a):
        rc = _oc(name, *a)
        torch.cuda.synchronize()
        return rc
    _lib.call = call
    ops._lib.call = call
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
g = load_golden("g7_subset_multi")
kw = dict(mode_multi_dec=True, multiS_conv=True)


def build():
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6, num_layers=8, act="mish", **kw)
    return M._load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()


net = build() if E("REUSE") else None
ref, ev, vals = None, 0, set()
rs = None
for it in range(n):
    m = net if net is not None else build()
    m.zero_grad(set_to_none=True)
    x = g["x"].to(dev).requires_grad_(True)
    outs = m(x)
    trace = {}
    saved = {}
    if E("NODES"):
        seen, stack = set(), [o.grad_fn for o in outs]
        nodes = []
        while stack:
            nd = stack.pop()
            if nd is None or nd in seen:
                continue
            seen.add(nd)
            if type(nd).__name__ != "AccumulateGrad":
                nodes.append(nd)
            stack.extend(f for f, _ in nd.next_functions)
        base = min(nd._sequence_nr() for nd in nodes)
        for nd in nodes:
            def hook(gin, gout, nd=nd, key=(nd._sequence_nr() - base, type(nd).__name__)):
                trace[key] = ([t.double().abs().sum() for t in gout if t is not None], [t.double().abs().sum() for t in gin if t is not None])    # (device scalars: no sync)
            nd.register_hook(hook)
            if type(nd).__name__ in ("NormActBackward", "Conv3dBackward", "NormActCatBackward"):
                def pre(gout, nd=nd, key=(nd._sequence_nr() - base, type(nd).__name__)):
                    try:
                        saved[key] = [(tuple(t.shape), str(t.dtype), t.float().abs().sum().double()) for t in nd.saved_tensors if t is not None]
                    except Exception as e:      # (released)
                        saved[key] = repr(e)
                nd.register_prehook(pre)
    if rs is None:
        rs = [g[f"r{i}"].to(dev) for i in range(len(outs))]
    torch.autograd.backward(outs, rs)
    gx = x.grad
    if ref is None:
        ref = gx.clone()
        ref_trace = trace
        ref_saved = saved
        continue
    d = float((gx - ref).norm() / ref.norm())
    if d > 1e-3:
        ev += 1
        vals.add(round(d, 5))
        if E("NODES") and ev <= 2:
            torch.cuda.synchronize()
            print(f"pass {it}: {len(trace)} backward nodes (pass 0: {len(ref_trace)})")
            shown = 0
            for key in sorted(trace, reverse=True):          # backward order: highest sequence number first
                if key not in ref_trace:
                    print("   node", key, "not in pass 0")
                    continue
                def rel(u, v):
                    return max([abs(float(p) - float(q)) / (abs(float(p)) + 1e-30) for p, q in zip(u, v)] + [0.0])
                ro, ri = rel(ref_trace[key][0], trace[key][0]), rel(ref_trace[key][1], trace[key][1])
                if (ro > 1e-5 or ri > 1e-5) and shown < 6:
                    per = [f"{abs(float(p) - float(q)) / (abs(float(p)) + 1e-30):.1e}" for p, q in zip(ref_trace[key][1], trace[key][1])]
                    print(f"   node {key}: incoming gradients differ {ro:.2e}, produced gradients differ {ri:.2e} (each: {per})")
                    if key in saved and isinstance(saved[key], list) and isinstance(ref_saved.get(key), list):
                        print("      saved tensors (shape, dtype, relative difference of |sum|):",
                              [(a[0], a[1], f"{abs(float(a[2]) - float(b[2])) / (abs(float(b[2])) + 1e-30):.1e}") for a, b in zip(saved[key], ref_saved[key])])
                    elif key in saved:
                        print("      saved:", saved[key] if not isinstance(saved[key], list) else "list", ref_saved.get(key) if not isinstance(ref_saved.get(key), list) else "list")
                    shown += 1
print(f"{' '.join(k for k in ('REUSE', 'NOSPLITSHARE', 'NOPRESPLIT', 'NOSTATS', 'NOCATNORM', 'SYNC') if E(k)) or 'default'}: events {ev} of {n - 1} passes; differences {sorted(vals)}")
