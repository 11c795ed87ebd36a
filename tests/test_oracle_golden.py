"""CPU: the oracle (oracle/) reproduces the golden vectors produced by the reference's own modules.
The goldens were produced by the reference run in float64 (stored as float32), and the oracle is run in
float64 here, so the tolerances below only have to cover the float32 storage rounding: a different
ALGORITHM (wrong eps, biased/unbiased variance, op order, key wiring ...) fails by orders of magnitude."""
import torch

import oracle
from helpers import load_golden, sub, pcg_state_dict, rel_err, cmp_prefix

torch.set_num_threads(8)
TOL = 1e-6      # forward, max-abs relative to max|ref|
GTOL = 1e-6     # gradients, relative L2


def _leafify(sd):
    out = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.double()
            out[k] = v.clone().requires_grad_(True) if "running" not in k else v
        else:
            out[k] = v
    return out


def _check_grads(sd, gold_grads, tol=GTOL):
    assert gold_grads, "fixture holds no gradients"
    for k, g in gold_grads.items():
        assert sd[k].grad is not None, k
        e = cmp_prefix(sd[k].grad, g)
        assert e < tol, (k, e)


def test_g1_base_unet():
    g = load_golden("g1_base_unet")
    sd = _leafify(sub(g, "sd"))
    x = g["x"].double().requires_grad_(True)
    y = oracle.base_unet(sd, "", x)
    assert rel_err(y, g["y"]) < TOL
    (y * g["r"]).sum().backward()
    assert cmp_prefix(x.grad, g["gx"]) < GTOL
    _check_grads(sd, sub(g, "grad"))


def test_g2_conv_3_1():
    g = load_golden("g2_conv_3_1")
    for act in ("relu", "mish"):
        sd = _leafify({"b." + k: v for k, v in sub(g, f"{act}/sd").items()})
        assert rel_err(oracle.conv_3_1(sd, "b", g["x"].double(), act, training=False), g[f"{act}/y_eval"]) < TOL
        x = g["x"].double().requires_grad_(True)
        bn = {}
        y = oracle.conv_3_1(sd, "b", x, act, training=True, bn_out=bn)
        assert rel_err(y, g[f"{act}/y_train"]) < TOL
        (y * g[f"{act}/r"]).sum().backward()
        assert cmp_prefix(x.grad, g[f"{act}/gx"]) < GTOL
        _check_grads(sd, {"b." + k: v for k, v in sub(g, f"{act}/grad").items()})
        after = sub(g, f"{act}/sd_after")
        for k, v in bn.items():
            assert rel_err(v.float(), after[k[2:]].float()) < 1e-5, k
        assert len(bn) == 6


def test_g2_conv_3_1_old():
    g = load_golden("g2_conv_3_1_old")
    sd = _leafify({"b." + k: v for k, v in sub(g, "sd").items()})
    assert rel_err(oracle.conv_3_1_old(sd, "b", g["x"].double(), training=False), g["y_eval"]) < TOL
    x = g["x"].double().requires_grad_(True)
    bn = {}
    y = oracle.conv_3_1_old(sd, "b", x, training=True, bn_out=bn)
    assert rel_err(y, g["y_train"]) < TOL
    (y * g["r"]).sum().backward()
    assert cmp_prefix(x.grad, g["gx"]) < GTOL
    _check_grads(sd, {"b." + k: v for k, v in sub(g, "grad").items()})
    after = sub(g, "sd_after")
    for k, v in bn.items():
        assert rel_err(v.float(), after[k[2:]].float()) < 1e-5, k


def test_g3_dual_dilated():
    g = load_golden("g3_dual_dilated")
    for act in ("relu", "mish"):
        sd = _leafify({"b." + k: v for k, v in sub(g, f"{act}/sd").items()})
        x = g["x"].double().requires_grad_(True)
        y = oracle.dual_dilated_block(sd, "b", x, act)
        assert rel_err(y, g[f"{act}/y"]) < TOL
        (y * g[f"{act}/r"]).sum().backward()
        assert cmp_prefix(x.grad, g[f"{act}/gx"]) < GTOL
        _check_grads(sd, {"b." + k: v for k, v in sub(g, f"{act}/grad").items()})


def test_g4_c3d_cascade():
    g = load_golden("g4_c3d_cascade")
    ya, yb = oracle.c3d_model(_leafify(sub(g, "sd")), g["x"].double())
    assert rel_err(ya, g["ya"]) < TOL and rel_err(yb, g["yb"]) < TOL


def test_g5_losses():
    g = load_golden("g5_loss")
    for freez in (True, False):
        pa = g["pa"].double().requires_grad_(True)
        pbs = [g[f"pb{i}"].double().requires_grad_(True) for i in range(4)]
        gt = g["gt"].double()
        l = oracle.gen_loss([pa, pbs], gt, 10, 1, casecade=True, freez=freez)
        tag = f"gen_freez{int(freez)}"
        assert abs(l.item() - g[tag].item()) < 1e-6 * abs(g[tag].item())
        l.backward()
        for i in range(4):
            assert cmp_prefix(pbs[i].grad, g[f"{tag}/gpb{i}"]) < 1e-6
        if not freez:
            assert cmp_prefix(pa.grad, g[f"{tag}/gpa"]) < 1e-5
        pb0 = g["pb0"].double().requires_grad_(True)
        l = oracle.loss_l1_masked([g["pa"].double(), pb0], gt, freez=freez)
        tag = f"l1_freez{int(freez)}"
        assert abs(l.item() - g[tag].item()) < 1e-6 * abs(g[tag].item())
        l.backward()
        assert cmp_prefix(pb0.grad, g[f"{tag}/gpb0"]) < 1e-5


def test_g5_loss_branches_huber_val_plain():
    """GenLoss(huber=True) in train and val mode, GenLoss val, Loss(casecade=False): loss.py:29-39, 100-103, 109-117."""
    g = load_golden("g5_loss")
    gt = g["gt"].double()
    pbs = [g[f"pb{i}"].double().requires_grad_(True) for i in range(4)]
    l = oracle.gen_loss([g["pa"].double(), pbs], gt, 10, 1, casecade=True, freez=True, huber=True)
    assert abs(l.item() - g["gen_huber"].item()) < 1e-6 * abs(g["gen_huber"].item())
    l.backward()
    for i in range(4):
        assert cmp_prefix(pbs[i].grad, g[f"gen_huber/gpb{i}"]) < 1e-6
    for tag, fn in (("gen_val_huber", lambda p: oracle.gen_loss_val(p, gt, huber=True)), ("l1_plain", lambda p: oracle.loss_l1_plain(p, gt))):
        pb0 = g["pb0"].double().requires_grad_(True)
        l = fn(pb0)
        assert abs(l.item() - g[tag].item()) < 1e-6 * abs(g[tag].item()), tag
        l.backward()
        assert cmp_prefix(pb0.grad, g[f"{tag}/gpb0"]) < 1e-6, tag
    assert abs(oracle.gen_loss_val(g["pb0"].double(), gt).item() - g["gen_val"].item()) < 1e-6 * abs(g["gen_val"].item())


def _subset_variant(tag, **kw):
    g = load_golden(f"g7_subset_{tag}")
    sd = _leafify(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
    x = g["x"].double().requires_grad_(True)
    bn = {}
    outs = oracle.main_subset_model(sd, "", x, 8, 6, "mish", True, bn_out=bn, **kw)
    for i, o in enumerate(outs):
        assert rel_err(o, g[f"y{i}"]) < TOL, (tag, i)
    sum((o * g[f"r{i}"]).sum() for i, o in enumerate(outs)).backward()
    assert cmp_prefix(x.grad, g["gx"]) < GTOL
    _check_grads(sd, sub(g, "grad"), tol=GTOL)
    for k, v in sub(g, "sd_after").items():
        assert rel_err(bn[k].float(), v.float()) < 1e-5, k


def test_g7_subset_multi():
    _subset_variant("multi", mode_multi_dec=True, multiS_conv=True)


def test_g7_subset_dual():
    _subset_variant("dual", mode_multi_dec=True, multiS_conv=False)


def test_g7_subset_plain():
    _subset_variant("plain", mode_multi_dec=False)


def test_g7_pyfer_model():
    g = load_golden("g7_pyfer_model")
    sd = _leafify(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
    with torch.no_grad():
        for training, suf in ((True, ""), (False, "_eval")):
            ya, ybs = oracle.dose_pyfer(sd, g["x"].double(), num_layers=4, num_heads=6, act="mish", training=training)
            assert rel_err(ya, g["ya" + suf]) < TOL
            for i, o in enumerate(ybs):
                assert rel_err(o, g[f"yb{i}{suf}"]) < TOL, (suf, i)


def test_g7_transeg():
    for tag in ("new", "old"):
        g = load_golden(f"g7_transeg_{tag}")
        sd = _leafify(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
        x = g["x"].double().requires_grad_(True)
        y = oracle.oar_transeg(sd, x, num_heads=12, training=True, old=(tag == "old"))
        assert rel_err(y, g["y"]) < TOL, tag
        (y * g["r"]).sum().backward()
        assert cmp_prefix(x.grad, g["gx"]) < GTOL
        _check_grads(sd, sub(g, "grad"), tol=GTOL)
