"""GPU parity of the HIP nn.Modules (through the C ABI) against the golden vectors produced by the reference's own
modules (tests/golden, fp64 reference runs) and against the CPU oracle.

Tolerance: the north-star's 1e-3 relative error (max|out-ref| / max|ref|) on network outputs for the fp32 (parity)
mode; gradients 2e-3 relative L2.  The bf16 (benchmark) mode is reported and only bounded loosely (it cannot meet an
fp32-class tolerance through ~100 normalised layers; see DESIGN.md "Precision")."""
import pytest
import torch

import oracle
from helpers import load_golden, sub, pcg_state_dict, rel_err, rel_l2, cmp_prefix

pytestmark = pytest.mark.gpu
OUT_TOL, GRAD_TOL = 1e-3, 2e-3


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _set(dtype):
    import dose_prediction_amd
    dose_prediction_amd.set_compute_dtype(dtype)


def _load(mod, sd):
    missing, unexpected = mod.load_state_dict(sd, strict=True), None
    return mod


def _check_grads(mod, gold, tol=GRAD_TOL):
    named = dict(mod.named_parameters())
    worst = ("", 0.0)
    # Some gradients are analytically (near-)zero -- a bias or a per-channel scale in front of a normalisation --
    # and consist of round-off only; errors are therefore measured against max(|gold|, 5e-2 * median gradient norm)
    # (the fp32 CPU oracle itself deviates from the fp64 goldens by the same amounts on those keys).
    norms = sorted(float(g.double().norm()) for g in gold.values())
    floor = 5e-2 * norms[len(norms) // 2]
    for k, g in gold.items():
        assert named[k].grad is not None, k
        ours = named[k].grad.detach().cpu().reshape(-1)[: g.numel()].double()
        e = float((ours - g.reshape(-1).double()).norm()) / max(float(g.double().norm()), floor)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst


def test_g1_base_unet():
    from dose_prediction_amd.models.c3d import BaseUNet
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g1_base_unet")
    net = _load(BaseUNet(3, [-1, 4, 8, 8, 16, 16]), sub(g, "sd")).to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < OUT_TOL
    y.backward(g["r"].to(dev))
    assert cmp_prefix(x.grad.cpu(), g["gx"]) < GRAD_TOL
    _check_grads(net, sub(g, "grad"))


@pytest.mark.parametrize("act", ["relu", "mish"])
def test_g2_conv_3_1(act):
    from dose_prediction_amd.blocks import conv_3_1
    from dose_prediction_amd.models.c3d import to_ndhwc, from_ndhwc
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g2_conv_3_1")
    blk = _load(conv_3_1(8, 4, act), sub(g, f"{act}/sd")).to(dev)
    blk.eval()
    y = from_ndhwc(blk(to_ndhwc(g["x"].to(dev))))
    assert rel_err(y.cpu(), g[f"{act}/y_eval"]) < OUT_TOL
    blk.train()
    x = g["x"].to(dev).requires_grad_(True)
    y = from_ndhwc(blk(to_ndhwc(x)))
    assert rel_err(y.cpu(), g[f"{act}/y_train"]) < OUT_TOL
    y.backward(g[f"{act}/r"].to(dev))
    assert cmp_prefix(x.grad.cpu(), g[f"{act}/gx"]) < GRAD_TOL
    _check_grads(blk, sub(g, f"{act}/grad"))
    after = sub(g, f"{act}/sd_after")
    for k, v in blk.state_dict().items():
        if "running" in k:
            assert rel_err(v.cpu().float(), after[k].float()) < 1e-4, k
        if "num_batches" in k:
            assert int(v) == int(after[k]), k


def test_g2_conv_3_1_old_and_dual():
    from dose_prediction_amd.blocks import conv_3_1_old, DualDilatedBlock
    from dose_prediction_amd.models.c3d import to_ndhwc, from_ndhwc
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g2_conv_3_1_old")
    blk = _load(conv_3_1_old(8, 4), sub(g, "sd")).to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    y = from_ndhwc(blk(to_ndhwc(x)))
    assert rel_err(y.cpu(), g["y_train"]) < OUT_TOL
    y.backward(g["r"].to(dev))
    assert cmp_prefix(x.grad.cpu(), g["gx"]) < GRAD_TOL
    _check_grads(blk, sub(g, "grad"))
    g = load_golden("g3_dual_dilated")
    for act in ("relu", "mish"):
        blk = _load(DualDilatedBlock(8, 4, act), sub(g, f"{act}/sd")).to(dev).train()
        x = g["x"].to(dev).requires_grad_(True)
        y = from_ndhwc(blk(to_ndhwc(x)))
        assert rel_err(y.cpu(), g[f"{act}/y"]) < OUT_TOL
        y.backward(g[f"{act}/r"].to(dev))
        assert cmp_prefix(x.grad.cpu(), g[f"{act}/gx"]) < GRAD_TOL
        _check_grads(blk, sub(g, f"{act}/grad"))


def test_g4_c3d_cascade():
    from dose_prediction_amd.models.c3d import Model
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g4_c3d_cascade")
    net = _load(Model(3, 1, [-1, 4, 4, 8, 8, 8], [-1, 4, 4, 8, 8, 8]), sub(g, "sd")).to(dev).train()
    ya, yb = net(g["x"].to(dev))
    assert rel_err(ya.cpu(), g["ya"]) < OUT_TOL and rel_err(yb.cpu(), g["yb"]) < OUT_TOL


@pytest.mark.parametrize("tag,kw", [("multi", dict(mode_multi_dec=True, multiS_conv=True)),
                                    ("dual", dict(mode_multi_dec=True, multiS_conv=False)),
                                    ("plain", dict(mode_multi_dec=False))])
def test_g7_subset(tag, kw):
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    dev = _dev()
    _set(torch.float32)
    g = load_golden(f"g7_subset_{tag}")
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                          num_layers=8, act="mish", **kw)
    assert list(net.state_dict().keys()) == list(g["keys"])
    _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    outs = net(x)
    for i, o in enumerate(outs):
        assert rel_err(o.cpu(), g[f"y{i}"]) < OUT_TOL, (tag, i)
    torch.autograd.backward(outs, [g[f"r{i}"].to(dev) for i in range(len(outs))])
    assert cmp_prefix(x.grad.cpu(), g["gx"]) < GRAD_TOL
    _check_grads(net, sub(g, "grad"))
    sd = net.state_dict()
    for k, v in sub(g, "sd_after").items():
        assert rel_err(sd[k].cpu().float(), v.float()) < 1e-4, k


def test_g7_pyfer_model():
    from dose_prediction_amd.models.dose_pyfer import Model
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g7_pyfer_model")
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 4, 8, 8, 16, 16], feature_size=4, img_size=(32, 16, 16), num_layers=4,
                num_heads=6, act="mish", mode_multi_dec=True, multiS_conv=True)
    assert list(net.state_dict().keys()) == list(g["keys"])
    _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev)
    x = g["x"].to(dev)
    with torch.no_grad():
        net.eval()
        ya, ybs = net(x)
        assert rel_err(ya.cpu(), g["ya_eval"]) < OUT_TOL
        for i, o in enumerate(ybs):
            assert rel_err(o.cpu(), g[f"yb{i}_eval"]) < OUT_TOL, i
        net.train()
        ya, ybs = net(x)
        assert rel_err(ya.cpu(), g["ya"]) < OUT_TOL
        for i, o in enumerate(ybs):
            assert rel_err(o.cpu(), g[f"yb{i}"]) < OUT_TOL, i


@pytest.mark.parametrize("tag", ["new", "old"])
def test_g7_transeg(tag):
    from dose_prediction_amd.models import oar_transeg
    dev = _dev()
    _set(torch.float32)
    g = load_golden(f"g7_transeg_{tag}")
    cls = oar_transeg.Model if tag == "new" else oar_transeg.TRANSEG
    net = cls(in_channels=1, out_channels=8, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=12,
              pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True, dropout_rate=0.0)
    assert list(net.state_dict().keys()) == list(g["keys"])
    _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < OUT_TOL
    # bit-exact OAR arg-max masks wherever the reference's own top-2 logit margin exceeds the fp32 noise floor
    ref = g["y"]
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()
    assert torch.equal(y.cpu().argmax(1)[safe], ref.argmax(1)[safe])
    assert safe.float().mean() > 0.98
    y.backward(g["r"].to(dev))
    assert cmp_prefix(x.grad.cpu(), g["gx"]) < GRAD_TOL
    # OldModels variant: BatchNorm->ReLU chains make a few gradients ill-conditioned in fp32 -- the fp32 CPU oracle itself
    # deviates from the fp64 golden by 6.1e-3 on decoder2...conv_7.conv.1.bias (measured) -- hence 1e-2 there.
    _check_grads(net, sub(g, "grad"), tol=GRAD_TOL if tag == "new" else 1e-2)


def test_bf16_mode_tracks_fp32():
    """bf16 benchmark mode: same graph, bf16 storage / MFMA; bounded loosely against the golden (reported in DESIGN.md)."""
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    dev = _dev()
    g = load_golden("g7_subset_multi")
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                          num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
    _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
    _set(torch.bfloat16)
    try:
        outs = net(g["x"].to(dev))
        for i, o in enumerate(outs):
            assert torch.isfinite(o).all()
            assert rel_l2(o.cpu(), g[f"y{i}"]) < 0.15, (i, rel_l2(o.cpu(), g[f"y{i}"]))
    finally:
        _set(torch.float32)
