"""Per-parameter gradient error of the fp32 / fp32x3 (/ bf16) modes against the float64 oracle, DOSE-PYFER full width at 64^3."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import dose_prediction_amd
from dose_prediction_amd import synth
from dose_prediction_amd.models.dose_pyfer import Model
dev = torch.device("cuda:0")
torch.manual_seed(4321)
S = (64, 64, 64)
net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8, num_heads=6, act="mish")
for n, p in net.named_parameters():
    if "net_A" in n or "conv_out_A" in n:
        p.requires_grad = False
x = synth.dose_input(1, S)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
trainable = [k for k, p in net.named_parameters() if p.requires_grad]
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
for k in trainable:
    sd64[k].requires_grad_(True)
g = torch.Generator().manual_seed(99)
rs = [torch.randn((1, 1, 64 >> i, 64 >> i, 64 >> i), generator=g) for i in range(4)]
ref = oracle.dose_pyfer(sd64, x.double(), num_layers=8, num_heads=6, act="mish", training=True)[1]
torch.autograd.backward(ref, [r.double() for r in rs])
net.to(dev).train()
for mode in sys.argv[1:] or ("fp32", "fp32x3"):
    dose_prediction_amd.config.set_x3_wgrad_terms(3 if mode == "fp32x3w3" else 1)
    dose_prediction_amd.config.set_x3_dgrad_terms(3 if mode in ("fp32x3d3", "fp32x3w3") else 1)       # (fp32x3 = the defaults: 1 / 1)
    dose_prediction_amd.config.set_x3_linear_wgrad_terms(1 if mode == "fp32x3l1" else 3)
    dose_prediction_amd.set_compute_dtype("fp32x3" if mode.startswith("fp32x3") else mode)
    net.load_state_dict(sd)
    net.zero_grad(set_to_none=True)
    outs = net(x.to(dev))[1]
    torch.autograd.backward(outs, [r.to(dev) for r in rs])
    named = dict(net.named_parameters())
    keys = [k for k in trainable if sd64[k].grad is not None and named[k].grad is not None]
    num = den = 0.0
    rows = []
    for k in keys:
        a, b = named[k].grad.detach().double().cpu(), sd64[k].grad
        e2, n2 = float((a - b).norm() ** 2), float(b.norm() ** 2)
        num += e2; den += n2
        rows.append((e2, (e2 / max(n2, 1e-300)) ** 0.5, n2 ** 0.5, k))
    print(f"== {mode}: outputs {[('%.2e' % float((o.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())) for o, r in zip(outs, ref)]}  gradient rel-L2 {(num / den) ** 0.5:.3e}")
    # the Linear layers (2-D weights: transformer, patch embedding) on their own: ADVICE r3 asked for per-layer evidence before their
    # weight gradients may use one product (config.set_x3_linear_wgrad_terms)
    lin = [(e2, rel, nrm, k) for e2, rel, nrm, k in rows if sd64[k].dim() == 2]
    if lin:
        le, ln = sum(r[0] for r in lin), sum(r[2] ** 2 for r in lin)
        rels = sorted(r[1] for r in lin)
        print(f"   Linear weights ({len(lin)} tensors, linear wgrad terms {dose_prediction_amd.config.x3_linear_wgrad_terms()}): rel-L2 {(le / ln) ** 0.5:.3e}  "
              f"per-tensor rel median {rels[len(rels) // 2]:.2e} max {rels[-1]:.2e}")
    for e2, rel, nrm, k in sorted(rows, reverse=True)[:12]:
        print(f"   share {e2 / num:6.3f}  rel {rel:.2e}  |g| {nrm:.2e}  {k}")
