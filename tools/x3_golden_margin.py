"""How close test_reference_goldens_in_x3_mode[g7_subset] runs to its thresholds: the worst output / gradient errors over N repetitions
(atomics and stream timing make the ReLU-gate flips differ from run to run).   python tools/x3_golden_margin.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dose_prediction_amd  # noqa: E402
import test_models_gpu as M  # noqa: E402
from helpers import rel_err, cmp_prefix, load_golden, pcg_state_dict, sub  # noqa: E402
from dose_prediction_amd.models.dose_pyfer import MainSubsetModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
from dose_prediction_amd import _lib  # noqa: E402
CALLS = []
if os.environ.get("CALLSEQ"):
    _orig_call = _lib.call

    def _logged(name, *a):
        CALLS.append((name,) + tuple(v if isinstance(v, int) and abs(v) < (1 << 31) else ("p" if isinstance(v, int) else type(v).__name__) for v in a))
        return _orig_call(name, *a)
    _lib.call = _logged
    import dose_prediction_amd.ops as _ops
    _ops._lib.call = _logged
cfg = dose_prediction_amd.config
for terms in ((3, 3, 3), (1, 1, 1))[: int(os.environ.get("NTERMS", "2"))]:
    cfg.set_x3_dgrad_terms(terms[0]); cfg.set_x3_wgrad_terms(terms[1]); cfg.set_x3_linear_wgrad_terms(terms[2])
    dose_prediction_amd.set_compute_dtype(os.environ.get("MODE", "fp32x3"))
    for tag, kw in (("multi", dict(mode_multi_dec=True, multiS_conv=True)), ("plain", dict(mode_multi_dec=False)))[: int(os.environ.get("NTAGS", "2"))]:
        nbad = 0
        wo = wgx = wg = 0.0
        for it in range(n):
            dev = torch.device("cuda:0")
            g = load_golden(f"g7_subset_{tag}")
            net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                                  num_layers=8, act="mish", **kw)
            M._load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
            x = g["x"].to(dev).requires_grad_(True)
            brec = []
            if os.environ.get("BHOOKS"):
                for name, m in net.named_modules():
                    if not list(m.children()):
                        m.register_full_backward_hook(lambda mod, gin, gout, name=name: brec.append((name, [t.detach().float().clone() for t in gout if t is not None],
                                                                                                       [t.detach().float().clone() for t in gin if t is not None])))
            outs = net(x)
            wo = max(wo, max(rel_err(o.cpu(), g[f"y{i}"]) for i, o in enumerate(outs)))
            torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
            wgx = max(wgx, cmp_prefix(x.grad.cpu(), g["gx"]))
            if os.environ.get("BHOOKS"):
                if it == 0:
                    ref_brec = brec
                else:
                    shown = 0
                    if it == 1:
                        print(f"   backward hooks fired: {len(brec)} (run 0: {len(ref_brec)}); order equal: {[b[0] for b in brec] == [b[0] for b in ref_brec]}")
                    for (n0, go0, gi0), (n1, go1, gi1) in zip(ref_brec, brec):
                        do = max([float((a - b).norm() / (a.norm() + 1e-30)) for a, b in zip(go0, go1)] + [0.0])
                        di = max([float((a - b).norm() / (a.norm() + 1e-30)) for a, b in zip(gi0, gi1)] + [0.0])
                        if (do > 1e-4 or di > 1e-4) and shown < 6:
                            print(f"   run {it}: backward of {n0} ({n1}): grad_output differs {do:.2e}, grad_input differs {di:.2e}")
                            shown += 1
            if os.environ.get("CALLSEQ"):
                seq = list(CALLS)
                del CALLS[:]
                if it == 0:
                    ref_seq = seq
                elif seq != ref_seq:
                    k = next((i for i, (a, b) in enumerate(zip(ref_seq, seq)) if a != b), min(len(seq), len(ref_seq)))
                    print(f"   run {it}: call sequence differs from run 0 at call {k} of {len(ref_seq)} / {len(seq)}: run 0 {ref_seq[k - 1:k + 2]}  this run {seq[k - 1:k + 2]}")
            cur = {k: p_.grad.detach().clone() for k, p_ in net.named_parameters() if p_.grad is not None}
            if it == 0:
                ref_grads = cur
            elif os.environ.get("VERBOSE"):
                difs = [(float((cur[k] - ref_grads[k]).norm() / (ref_grads[k].norm() + 1e-30)), k) for k in cur]
                big = [(round(d, 5), k) for d, k in difs if d > 1e-3]
                if big:
                    print(f"   run {it}: {len(big)} of {len(difs)} gradient tensors differ from run 0 by > 1e-3; in decoder2 / out:",
                          [(round(d, 6), k) for d, k in difs if ("decoder2" in k or k.startswith("out") or "decoder.out" in k)])
            try:
                M._check_grads(net, sub(g, "grad"), tol=0.0)
            except AssertionError as e:
                wg = max(wg, e.args[0][1])
                nbad += e.args[0][1] > 2e-3 and terms == (3, 3, 3)
                if os.environ.get("VERBOSE") and e.args[0][1] > 2e-3:
                    print(f"   run {it}: worst parameter {e.args[0][0]} {e.args[0][1]:.2e}  gx {cmp_prefix(x.grad.cpu(), g['gx']):.2e}")
        print(f"terms {terms} {tag:6s}: worst of {n} runs: outputs {wo:.2e} (tol 1e-3)  gx {wgx:.2e} (tol 1e-2)  parameter gradients {wg:.2e} (tol 1e-2); runs above 2e-3: {nbad}")
