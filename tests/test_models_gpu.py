"""GPU parity of the HIP nn.Modules (through the C ABI) against the golden vectors produced by the reference's own
modules (tests/golden, fp64 reference runs) and against the CPU oracle.

Tolerance: the north-star's 1e-3 relative error (max|out-ref| / max|ref|) on network outputs for the fp32 (parity)
mode; gradients 2e-3 relative L2.  The 16-bit modes are gated against the storage-rounding budget in
tests/test_precision_budget_gpu.py (HIP error <= 1.5 x the error of the oracle with emulated 16-bit storage)."""
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


GOLDEN_GATES = ("test_g1_", "test_g2_", "test_g4_", "test_g7_")


def pytest_generate_tests(metafunc):
    """Every reference-golden network gate (test_g*) runs in BOTH reduction modes: "det" = config.set_deterministic(True),
    the order-fixed reductions (slab scratch + k_conv_split_finish, batched split-K, gather trilinear backward); "atomic" = the default the
    benchmark times (fp32 atomics in the split-kd convolutions, split-K GEMMs, LayerNorm / trilinear backward).  Same tolerances in both:
    in the exact-fp32 mode the run-to-run spread of the atomic paths is 1e-6 .. 5e-4 on these networks (tests/golden/grad_bands.json,
    mode "fp32": the float64 oracle with one fp32 rounding injected at every stored tensor), far inside the 2e-3 gate -- no retry."""
    if metafunc.function.__name__.startswith(GOLDEN_GATES):
        metafunc.parametrize("_reduction_mode", ["det", "atomic"], indirect=True)


@pytest.fixture(autouse=True)
def _reduction_mode(request):
    import dose_prediction_amd
    mode = getattr(request, "param", None)
    on = mode == "det" and torch.cuda.is_available()
    if on:
        dose_prediction_amd.config.set_deterministic(True)
    yield mode
    if on:
        dose_prediction_amd.config.set_deterministic(False)


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
    return worst


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
    # OldModels variant: BatchNorm->ReLU chains make a few gradients ill-conditioned in fp32 -- the fp32 CPU oracle itself
    # deviates from the fp64 golden by 6.1e-3 on decoder2...conv_7.conv.1.bias (measured) -- hence 1e-2 there.  The same gate event reaches
    # the input gradient: 200 exact-fp32 passes with the default (atomic) reductions have a median of 9e-7 and a maximum of 1.4e-3
    # (tests/golden/x3_atomic_spread.json, mode "fp32": one ReLU gate at a pre-activation within round-off of zero), so the input-gradient
    # gate of the old variant is 1e-2 as well (the new variant: 2.2e-4 in every pass, gate 2e-3).
    assert cmp_prefix(x.grad.cpu(), g["gx"]) < (GRAD_TOL if tag == "new" else 1e-2)
    _check_grads(net, sub(g, "grad"), tol=GRAD_TOL if tag == "new" else 1e-2)


def test_transeg_default_conv_patch_embedding():
    """oar_transeg.Model's own constructor default pos_embed="conv" (oar_transeg.py:28; MONAI PatchEmbeddingBlock: Conv3d with
    kernel = stride = patch, flatten, transpose): state_dict keys patch_embeddings.{weight,bias} in MONAI's layout, forward and
    backward against the oracle's restatement (torch conv3d), fp32 and fp32x3."""
    import dose_prediction_amd
    from dose_prediction_amd.models import oar_transeg
    dev = _dev()
    torch.manual_seed(3)
    net = oar_transeg.Model(in_channels=1, out_channels=8, img_size=(32, 16, 16), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=12)
    keys = list(net.state_dict().keys())
    assert "vit.patch_embedding.patch_embeddings.weight" in keys and "vit.patch_embedding.patch_embeddings.1.weight" not in keys
    assert tuple(net.state_dict()["vit.patch_embedding.patch_embeddings.weight"].shape) == (48, 1, 16, 16, 16)
    sd64 = {k: (v.detach().double().requires_grad_(v.dtype.is_floating_point and "running" not in k) if v.dtype.is_floating_point else v)
            for k, v in net.state_dict().items()}
    x = torch.randn((2, 1, 32, 16, 16), generator=torch.Generator().manual_seed(5))
    r = torch.randn((2, 8, 32, 16, 16), generator=torch.Generator().manual_seed(6))
    ref = oracle.oar_transeg(sd64, x.double(), num_heads=12, training=True)
    ref.backward(r.double())
    gw = sd64["vit.patch_embedding.patch_embeddings.weight"].grad
    net.to(dev).train()
    for mode in (torch.float32, "fp32x3"):
        _set(mode)
        try:
            net.zero_grad(set_to_none=True)
            y = net(x.to(dev))
            assert rel_err(y.cpu(), ref.detach()) < OUT_TOL, mode
            y.backward(r.to(dev))
            got = net.vit.patch_embedding.patch_embeddings.weight.grad
            # (2e-2: this gradient runs back through twelve transformer layers and the whole decoder of a tiny random network, where a
            # handful of ReLU gates within round-off of zero move it by 0.1-0.6 % from run to run (split-kd atomics); a wrong axis
            # order in the permuted weight view would be an O(1) error)
            assert got.shape == gw.shape and rel_l2(got.cpu(), gw) < 2e-2, mode
        finally:
            _set(torch.float32)
    with pytest.raises(ValueError):
        oar_transeg.Model(in_channels=1, out_channels=8, img_size=(32, 16, 16), pos_embed="sincos")


def test_cascade_glue():
    """TRANSEG -> arg-max -> one-hot -> axis reversal -> cat(ptv, oars, ct) -> PYFER -> mask/clip x70
    (train_light_linked_model.py:143-173) against the oracle.  The dose comparison feeds the ORACLE dose network with the
    HIP path's own masks, so a (legitimate) arg-max flip on a near-tie cannot hide or fake an error."""
    import dose_prediction_amd
    from dose_prediction_amd import cascade
    from dose_prediction_amd.models import dose_pyfer, oar_transeg
    dev = _dev()
    _set(torch.float32)
    torch.manual_seed(3)
    S = (32, 32, 32)
    seg = oar_transeg.Model(1, 8, S, feature_size=4, hidden_size=48, mlp_dim=96, num_heads=12, pos_embed="perceptron").to(dev).eval()
    dose = dose_pyfer.Model(9, 1, [-1, 4, 8, 8, 16, 16], feature_size=4, img_size=S, num_layers=4, num_heads=6).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    ct = torch.randn((1, 1) + S, generator=g)
    ptv = (torch.rand((1, 1) + S, generator=g) > 0.7).float()
    mask = (torch.rand((1, 1) + S, generator=g) > 0.4).float()
    dose_gy, labels = cascade.cascade_forward(seg, dose, ct.to(dev), ptv.to(dev), mask.to(dev))
    # stage 1: segmentation logits and masks
    sd_seg = {k: v.detach().cpu() for k, v in seg.state_dict().items()}
    with torch.no_grad():
        ref_logits = oracle.oar_transeg(sd_seg, ct, num_heads=12, training=False)
    top2 = ref_logits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref_logits.abs().max()
    lab = labels.cpu().long()
    assert torch.equal(lab[safe], ref_logits.argmax(1)[safe]) and safe.float().mean() > 0.98
    # stage 2: dose network on the HIP path's own masks
    onehot = torch.nn.functional.one_hot(lab, 8).permute(0, 4, 1, 2, 3).float()          # [1,8,D,H,W]
    oars = onehot[0].permute(0, 3, 2, 1).unsqueeze(0)[:, 1:]                               # lines 158, 165
    structures = torch.cat((ptv, oars, ct.permute(0, 1, 4, 3, 2)), dim=1)                 # lines 163, 167
    sd_dose = {k: v.detach().cpu() for k, v in dose.state_dict().items()}
    with torch.no_grad():
        ref = oracle.dose_pyfer(sd_dose, structures, num_layers=4, num_heads=6, act="mish", training=False)[1][0]
    ref_gy = oracle.dose_postprocess(ref, mask)
    assert rel_err(dose_gy.cpu(), ref_gy) < OUT_TOL
    # the training-time variant of the glue hands the same 9-channel tensor to the trainer (fp32 mode: exact)
    st, lab2 = cascade.cascade_structures(seg, ct.to(dev), ptv.to(dev))
    assert torch.equal(lab2.cpu(), labels.cpu()) and torch.equal(st.cpu(), structures)
    # ... and the staged (NDHWC) form fed to forward_staged() gives the same dose as the NCDHW form through forward()
    stg, _ = cascade.cascade_structures(seg, ct.to(dev), ptv.to(dev), staged=True)
    with torch.no_grad():
        a, b = dose(st)[1][0], dose.forward_staged(stg)[1][0]
    assert rel_err(a.cpu(), b.cpu()) < 1e-5          # (split-kd atomics: not bitwise)


def test_activation_checkpointing():
    """BASELINE.json configs[4] trains with activation checkpointing: torch.utils.checkpoint around a decoder block must give the
    same outputs and gradients as the plain HIP path (custom autograd Functions are re-run in the backward)."""
    from torch.utils.checkpoint import checkpoint
    from dose_prediction_amd.models.base_blocks import ModifiedUnetrUpBlock
    from dose_prediction_amd.models.c3d import to_ndhwc, from_ndhwc
    dev = _dev()
    _set(torch.float32)
    torch.manual_seed(1)
    blk = ModifiedUnetrUpBlock(3, 16, 8, 2, act="mish").to(dev).train()
    x = torch.randn(1, 16, 4, 6, 8, device=dev, requires_grad=True)
    skip = torch.randn(1, 8, 8, 12, 16, device=dev, requires_grad=True)
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(use_ckpt):
        blk.load_state_dict(sd0)                       # same BatchNorm buffers at the start of both runs
        blk.zero_grad()
        for t in (x, skip):
            t.grad = None
        a, b = to_ndhwc(x), to_ndhwc(skip)
        y = checkpoint(blk, a, b, use_reentrant=False) if use_ckpt else blk(a, b)
        out = from_ndhwc(y)
        out.square().mean().backward()
        return out.detach().clone(), x.grad.clone(), skip.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}

    o1, gx1, gs1, gp1 = run(False)
    o2, gx2, gs2, gp2 = run(True)
    assert rel_err(o2.cpu(), o1.cpu()) < 1e-6 and rel_l2(gx2.cpu(), gx1.cpu()) < 1e-4 and rel_l2(gs2.cpu(), gs1.cpu()) < 1e-4
    for k in gp1:
        assert rel_l2(gp2[k].cpu(), gp1[k].cpu()) < 1e-3, k


def test_losses_and_dose_metrics_against_reference_goldens():
    """GenLoss / Loss (Train/loss.py) and the validation dose score (train_light_pyfer.py:166-172,
    evaluate_openKBP.py:42-48) through dp_masked_l1_fwd/_bwd and dp_dose_postprocess, against the values and gradients the
    reference's own loss module produced (fixture g5_loss) and against the oracle's post-processing."""
    from dose_prediction_amd import losses, ops
    dev = torch.device("cuda:0")
    g = load_golden("g5_loss")
    gt = g["gt"].to(dev)
    for freez in (True, False):
        pa = g["pa"].to(dev).requires_grad_(True)
        pbs = [g[f"pb{i}"].to(dev).requires_grad_(True) for i in range(4)]
        l = losses.gen_loss([pa, pbs], gt, 10.0, 1.0, casecade=True, freez=freez)
        tag = f"gen_freez{int(freez)}"
        assert abs(l.item() - g[tag].item()) < 2e-5 * abs(g[tag].item()), (l.item(), g[tag].item())
        l.backward()
        for i in range(4):
            assert cmp_prefix(pbs[i].grad.cpu(), g[f"{tag}/gpb{i}"]) < 2e-5
        if not freez:
            assert cmp_prefix(pa.grad.cpu(), g[f"{tag}/gpa"]) < 2e-5
        pb0 = g["pb0"].to(dev).requires_grad_(True)
        l = losses.l1_loss([g["pa"].to(dev), pb0], gt, freez=freez)
        tag = f"l1_freez{int(freez)}"
        assert abs(l.item() - g[tag].item()) < 2e-5 * abs(g[tag].item())
        l.backward()
        assert cmp_prefix(pb0.grad.cpu(), g[f"{tag}/gpb0"]) < 2e-5
    # the remaining branches of loss.py: Huber (train / val), validation L1, Loss(casecade=False)
    pbs = [g[f"pb{i}"].to(dev).requires_grad_(True) for i in range(4)]
    l = losses.gen_loss([g["pa"].to(dev), pbs], gt, 10.0, 1.0, casecade=True, freez=True, huber=True)
    assert abs(l.item() - g["gen_huber"].item()) < 2e-5 * abs(g["gen_huber"].item())
    l.backward()
    for i in range(4):
        assert cmp_prefix(pbs[i].grad.cpu(), g[f"gen_huber/gpb{i}"]) < 2e-5
    for tag, fn in (("gen_val_huber", lambda p: losses.gen_loss(p, gt, mode="val", huber=True)),
                    ("l1_plain", lambda p: losses.l1_loss(p, gt, casecade=False))):
        pb0 = g["pb0"].to(dev).requires_grad_(True)
        l = fn(pb0)
        assert abs(l.item() - g[tag].item()) < 2e-5 * abs(g[tag].item()), tag
        l.backward()
        assert cmp_prefix(pb0.grad.cpu(), g[f"{tag}/gpb0"]) < 2e-5, tag
    assert abs(losses.gen_loss(g["pb0"].to(dev), gt, mode="val").item() - g["gen_val"].item()) < 2e-5 * abs(g["gen_val"].item())
    # dose score: post-processing + masked MAE in Gy, ragged length (not a multiple of 4) and an empty mask
    torch.manual_seed(7)
    for n in (5003, 8192 * 3 + 1):
        pred, dose = torch.randn(n) * 0.5 + 0.3, torch.rand(n)
        mask = (torch.rand(n) > 0.6).float()
        ref = 70.0 * oracle.dose_mae(oracle.dose_postprocess(pred, mask) / 70.0, dose, mask)
        got = ops.dose_score(pred.to(dev), dose.to(dev), mask.to(dev))
        assert abs(got.item() - ref.item()) < 2e-5 * abs(ref.item())
        pp = ops.dose_postprocess(pred.to(dev), mask.to(dev))
        assert torch.equal(pp.cpu(), oracle.dose_postprocess(pred, mask))
    z = ops.masked_l1(torch.randn(100, device=dev), torch.randn(100, device=dev), torch.zeros(100, device=dev))
    assert z.item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sliding_window_inference(dtype):
    """sliding_window_inference(ct, roi, sw_batch 4, predictor) of train_light_linked_model.py:152-153 for a
    volume larger than the segmentation crop (48 x 32 x 40 with a 32^3 crop: 2 x 1 x 2 overlapping windows per image, batch 2,
    so the last window group is ragged) against the oracle's restatement of MONAI's algorithm driving the oracle network;
    window layout against hand-computed origins; roi == volume degenerates to the plain forward."""
    from dose_prediction_amd import cascade
    from dose_prediction_amd.models import oar_transeg
    dev = _dev()
    assert cascade.window_starts((128, 128, 128), (96, 96, 96)) == [[0, 32]] * 3          # interval 72: windows at 0 and 128-96
    assert cascade.window_starts((192, 192, 128), (96, 96, 96)) == [[0, 72, 96], [0, 72, 96], [0, 32]]
    assert cascade.window_starts((96, 96, 96), (96, 96, 96)) == [[0]] * 3
    _set(dtype)
    try:
        torch.manual_seed(11)
        roi, vol = (32, 32, 32), (48, 32, 40)
        seg = oar_transeg.Model(1, 8, roi, feature_size=4, hidden_size=48, mlp_dim=96, num_heads=12, pos_embed="perceptron").to(dev).eval()
        ct = torch.randn((2, 1) + vol, generator=torch.Generator().manual_seed(6))
        if dtype == torch.bfloat16:
            ct = ct.bfloat16().float()
        got = cascade.sliding_window_logits(seg, ct.to(dev), roi, sw_batch_size=3, overlap=0.25)
        got = got.float().permute(0, 4, 1, 2, 3).cpu()
        sd = {k: v.detach().cpu() for k, v in seg.state_dict().items()}
        with torch.no_grad():
            ref = oracle.sliding_window_inference(ct, roi, 3, lambda w: oracle.oar_transeg(sd, w, num_heads=12, training=False), overlap=0.25)
        assert rel_err(got, ref) < (OUT_TOL if dtype == torch.float32 else 0.15)
        one = cascade.sliding_window_logits(seg, ct[:, :, :32, :, :32].contiguous().to(dev), roi)
        direct = seg.forward_ndhwc(cascade.to_ndhwc(ct[:, :, :32, :, :32].contiguous().to(dev)))
        assert rel_err(one.float().cpu(), direct.float()[..., :one.shape[-1]].cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)   # split-kd atomics: not bitwise
    finally:
        _set(torch.float32)


def test_gradient_allreduce_single_rank_rccl_is_identity():
    """The bucketed gradient exchange on ONE rank over RCCL (torch.distributed 'nccl'): packing, in-place big buckets, AVG over a
    world of 1 and the .grad re-pointing must leave every gradient equal to the plain backward pass (to atomics round-off), twice in a row
    (bucket state resets), and parameters without gradients stay without.  World sizes > 1 are covered on CPU (gloo, test_ddp_cpu)."""
    import os
    import torch.distributed as dist
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    dev = _dev()
    g = load_golden("g7_subset_multi")

    def build():
        net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                              num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
        return _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()

    def grads(net):
        out = {}
        for step in range(2):
            net.zero_grad(set_to_none=True)
            sum((o * o).mean() for o in net(g["x"].to(dev) * (1.0 + 0.1 * step))).backward()
            torch.cuda.synchronize()
            out[step] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}
        return out

    ref = grads(build())
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        net = build()
        red = attach_gradient_allreduce(net, bucket_mb=0.05)    # small buckets: several packed ones and chunked big tensors
        got = grads(net)
        from dose_prediction_amd import ops
        n_direct = sum(1 for p in net.parameters() if p.grad is not None and p.data_ptr() in ops.GRAD_DEST)
        assert n_direct > 20, n_direct          # conv / linear weight gradients are produced inside their bucket slots
        red.close()
        assert not ops.GRAD_DEST
    finally:
        dist.destroy_process_group()
    for step in ref:
        norms = sorted(float(r.double().norm()) for r in ref[step].values() if r is not None)
        floor = 5e-2 * norms[len(norms) // 2]           # analytically zero gradients (a bias in front of a normalisation) are round-off only
        for k, r in ref[step].items():
            o = got[step][k]
            assert (r is None) == (o is None), (step, k)
            if r is not None:
                # not bitwise: the fp32 atomics of the split-K / split-kd kernels make two PLAIN runs differ by up to 7e-4 on these
                # gradients (tools/grad_noise.py); a lost, doubled or mis-scaled gradient would be an O(1) error
                e = float((o.double() - r.double()).norm()) / max(float(r.double().norm()), floor)
                assert e < 5e-3, (step, k, e)


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16", "fp32-rs_ag"])
def test_gradient_allreduce_two_ranks_matches_the_mean_of_local_gradients(grad_dtype):
    """World size 2 on the GPU (tests/ddp_gpu_worker.py: RCCL with one GPU per rank when the box has two, else gloo with both ranks on
    this box's one GPU; fp32 buckets incl. the in-place chunked exchange of tensors above the bucket size, and bf16 buckets): with bucket boundaries that fall
    between deferred Linear weights and their biases (such a bucket used to be exchanged before the grouped weight-gradient
    launch had written it), every gradient after the exchange equals the mean of the two ranks' local gradients, over three passes."""
    import os
    import subprocess
    import sys
    _dev()
    here = os.path.dirname(os.path.abspath(__file__))
    # "fp32-rs_ag": the same exchange as reduce-scatter + all-gather per bucket chunk (DOSE_DDP_ALGO=rs_ag, round 6)
    gd, _, algo = grad_dtype.partition("-")
    env = dict(os.environ, DDP_TEST_GRAD_DTYPE=gd, DOSE_DDP_ALGO=algo or "allreduce")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", {"fp32": "29547", "bf16": "29548"}.get(grad_dtype, "29549"), os.path.join(here, "ddp_gpu_worker.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DDP_GPU_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_two_ranks_gloo_on_one_gpu():
    """`python bench.py --gpus 2` (no launcher: bench.py starts its own torch.distributed.run child) with the gloo backend, both ranks
    on the one GPU of the test box: the N > 1 path (rank env, bucketed exchange, barrier + max-over-ranks timing, one JSON line
    from rank 0) cannot rot between rounds.  RCCL itself needs >= 2 GPUs; the driver's 8-GPU run covers that."""
    import json
    import os
    import subprocess
    import sys
    _dev()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DOSE_DDP_BACKEND="gloo")
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "32", "--batch", "1", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0 and res["config"]["global_batch"] == 2
    # the line says how many ranks took part in a collective on which backend (round 6: `rccl_ranks_seen`; "nccl" on a multi-GPU box)
    seen = res["rccl_ranks_seen"]
    assert seen["ranks_counted"] == 2 and seen["rank_sum"] == seen["rank_sum_expected"] == 1 and seen["backend"] == "gloo"
    # a launcher / flag mismatch is an error, not a silent 1-GPU run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--size", "32", "--steps", "1"],
                         env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)


def test_vit_backward_is_issued_before_the_128_cube_branch():
    """SURVEY H5 / VERDICT r1 9b: the patch-embedding weight (48 % of the gradient bytes) must not be the LAST gradient of the
    backward pass: its post-accumulate hook fires before skip1's first convolution's, and the side-stream forward equals the
    single-stream forward."""
    import dose_prediction_amd
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    dev = _dev()
    _set(torch.float32)
    g = load_golden("g7_subset_multi")
    outs = {}
    for side in (True, False):
        dose_prediction_amd.config.set_vit_side_stream(side)
        try:
            net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                                  num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
            _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
            order = []
            for name, p in net.named_parameters():
                p.register_post_accumulate_grad_hook(lambda _p, _n=name: order.append(_n))
            o = net(g["x"].to(dev))
            torch.autograd.backward(o, [g[f"r{i}"].to(dev) for i in range(4)])
            torch.cuda.synchronize()
            assert order.index("encoder.vit.patch_embedding.patch_embeddings.1.weight") < order.index("encoder.skip1.layer.conv1.conv.weight")
            outs[side] = ([t.detach().clone() for t in o], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            dose_prediction_amd.config.set_vit_side_stream(True)
    for a, b in zip(outs[True][0], outs[False][0]):
        assert rel_err(a.cpu(), b.cpu()) < 1e-5
    for n in outs[True][1]:
        assert cmp_prefix(outs[True][1][n].cpu(), outs[False][1][n].cpu()) < 1e-4, n


def test_captured_step_with_side_streams_replays_the_eager_steps():
    """A whole training step (forward, loss, backward, capturable FusedAdam + pack refresh) captured into a HIP graph WITH the side
    streams (config.set_capture_side_streams, round 4: forks / joins become graph edges) and replayed three times gives the parameters
    three eager steps give from the same start; the capture really contains work of several streams (the eager twin runs on ONE stream)."""
    import dose_prediction_amd
    from dose_prediction_amd import losses, synth
    from dose_prediction_amd.models.dose_pyfer import Model
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    _set(torch.float32)
    S = (32, 32, 32)

    def mk():
        torch.manual_seed(77)
        net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 8, 16, 16, 32, 32], feature_size=8, img_size=S, num_layers=4, num_heads=6, act="mish").to(dev).train()
        for n, p in net.named_parameters():
            if "net_A" in n or "conv_out_A" in n:
                p.requires_grad = False
        params = [p for p in net.parameters() if p.requires_grad]
        return net, FusedAdam(params, lr=1e-3, weight_decay=3e-5, amsgrad=True, capturable=True)
    x, gt = synth.dose_input(1, S).to(dev), synth.dose_target(1, S).to(dev)

    def step(net, opt):
        opt.zero_grad(set_to_none=True)
        loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
        loss.backward()
        opt.step()
        return loss
    try:
        # eager twin, everything on one stream
        for sw in (dose_prediction_amd.config.set_branch_stream, dose_prediction_amd.config.set_vit_side_stream, dose_prediction_amd.config.set_wgrad_stream):
            sw(False)
        net_e, opt_e = mk()
        for _ in range(4):
            le = step(net_e, opt_e)
        torch.cuda.synchronize()
        for sw in (dose_prediction_amd.config.set_branch_stream, dose_prediction_amd.config.set_vit_side_stream, dose_prediction_amd.config.set_wgrad_stream):
            sw(True)
        net_g, opt_g = mk()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            step(net_g, opt_g)                     # eager warm-up on the capture stream (side streams chosen for it, packs built)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        opt_g.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        from dose_prediction_amd import streams
        with torch.cuda.graph(graph, stream=cap):
            lg = step(net_g, opt_g)
        used = streams._CHOSEN.get((dev.index, cap.cuda_stream))
        assert used and len({s.cuda_stream for s in used}) >= 2, "no side streams were chosen for the capture stream"
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        for sw in (dose_prediction_amd.config.set_branch_stream, dose_prediction_amd.config.set_vit_side_stream, dose_prediction_amd.config.set_wgrad_stream):
            sw(True)
    assert torch.isfinite(lg).all()
    assert abs(float(lg.detach()) - float(le.detach())) < 2e-3 * abs(float(le.detach())), (float(lg.detach()), float(le.detach()))
    # Adam turns round-off-level gradient differences (atomics, stream order) into lr-sized steps of random sign, so the two runs are
    # compared as whole update vectors: distance between the trained parameters / length of the eager run's own update (4 steps)
    net0, _ = mk()
    num = den = 0.0
    for (n, a), (_, b), (_, c) in zip(net_g.named_parameters(), net_e.named_parameters(), net0.named_parameters()):
        if a.requires_grad:
            num += (a.detach() - b.detach()).double().pow(2).sum().item()
            den += (b.detach() - c.detach()).double().pow(2).sum().item()
    assert den > 0 and (num / den) ** 0.5 < 0.25, (num, den)


def test_branch_streams_match_the_single_stream_result():
    """config.set_branch_stream (on by default since round 3): the 3x3x3 branch of every multi-scale block and the small up-sampling
    blocks of the encoder run on a further HIP stream beside the 7x7x7 branch / the 128^3 block.  Same outputs and gradients as with
    everything on one stream -- on the default stream and on a user stream, bf16 included, repeated so that a missing event or an
    allocator reuse across streams would show as a difference between the repetitions."""
    import contextlib
    import dose_prediction_amd
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    dev = _dev()
    g = load_golden("g7_subset_multi")
    user = torch.cuda.Stream()
    for dtype, otol, gtol in ((torch.float32, 1e-5, 1e-4), (torch.bfloat16, 2e-2, 5e-2)):
        _set(dtype)
        outs = {}
        try:
            for tag, branch, side, stream in (("ref", False, False, None), ("wg", False, False, None), ("on", True, True, None), ("on2", True, True, None),
                                              ("user", True, True, user)):
                dose_prediction_amd.config.set_branch_stream(branch)
                dose_prediction_amd.config.set_vit_side_stream(side)
                # (the convolutions' weight gradients on their own stream, config.set_wgrad_stream: everywhere but in the reference run)
                dose_prediction_amd.config.set_wgrad_stream(tag != "ref")
                net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                                      num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
                _load(net, pcg_state_dict(g["keys"], g["shapes"], g["seed"])).to(dev).train()
                x, rs = g["x"].to(dev), [g[f"r{i}"].to(dev) for i in range(4)]
                torch.cuda.synchronize()
                with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                    for _ in range(2):           # second pass: every buffer of the first one has been through the caching allocator
                        net.zero_grad(set_to_none=True)
                        o = net(x)
                        torch.autograd.backward(o, rs)
                torch.cuda.synchronize()
                outs[tag] = ([t.detach().clone() for t in o], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            dose_prediction_amd.config.set_branch_stream(True)
            dose_prediction_amd.config.set_vit_side_stream(True)
            dose_prediction_amd.config.set_wgrad_stream(True)
            _set(torch.float32)
        for tag in ("wg", "on", "on2", "user"):
            for a, b in zip(outs[tag][0], outs["ref"][0]):
                assert rel_err(a.cpu(), b.cpu()) < otol, (dtype, tag)
            assert outs[tag][1].keys() == outs["ref"][1].keys()
            for n in outs["ref"][1]:
                assert cmp_prefix(outs[tag][1][n].cpu(), outs["ref"][1][n].cpu()) < gtol, (dtype, tag, n)


def test_host_running_ahead_of_the_gpu_does_not_corrupt_pointer_tables():
    """The fused Adam step and the grouped transformer weight-gradient launch read pointer tables that the host writes into PINNED
    buffers and copies asynchronously.  bench.py never synchronises between steps, so the host runs several steps ahead of the GPU;
    a table rewritten before its copy has executed makes a kernel use the NEXT step's pointers.  Here the GPU is held back by a
    long sleep kernel while six optimizer steps are enqueued, and every step keeps its gradient tensors alive so that the addresses
    change from step to step; the result must equal the run that synchronises after every step."""
    from dose_prediction_amd import blocks
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    _set(torch.float32)
    x = torch.randn(2, 8, 8, 8, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def run(synced):
        torch.manual_seed(11)
        vit = blocks.ViT(3, (8, 8, 8), (4, 4, 4), hidden_size=96, mlp_dim=192, num_layers=3, num_heads=6, pos_embed="perceptron").to(dev)
        params = [p for p in vit.parameters()]
        opt = FusedAdam(params, lr=1e-3, amsgrad=True)
        keep, losses_ = [], []
        if not synced:
            torch.cuda._sleep(int(1.5e9))                  # ~0.7 s at 2.1 GHz: the host enqueues all six steps meanwhile
        for it in range(6):
            opt.zero_grad(set_to_none=True)
            out, hidden = vit(x)
            loss = out.float().square().mean() + sum(h.float().mean() for h in hidden)
            loss.backward()
            opt.step()
            keep.append([p.grad for p in params])           # addresses of the next step's gradients differ
            losses_.append(loss.detach())
            if synced:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return torch.stack(losses_).cpu(), [p.detach().cpu().clone() for p in params]

    l_ref, p_ref = run(True)
    l_run, p_run = run(False)
    assert torch.isfinite(l_run).all()
    assert torch.allclose(l_run, l_ref, rtol=1e-4, atol=1e-6), (l_run, l_ref)
    for a, b in zip(p_run, p_ref):
        assert rel_l2(a, b) < 1e-3


def test_side_streams_do_not_share_a_queue_with_the_callers_stream():
    """dose_prediction_amd.streams: the transformer / branch / weight-gradient streams are drawn from torch's pool and kept only if a
    launch on them can overtake a spin kernel enqueued earlier on the caller's stream (and on each other) -- HIP multiplexes streams
    onto four hardware queues, and two streams on one queue run in host order.  Here: three distinct streams per caller's stream, all
    accepted by the probe on a fresh process (more streams than queues would show as rejected candidates, which is legal), cached,
    the same set whichever of them asks (root()), and a second caller's stream gets its own set."""
    from dose_prediction_amd import streams
    dev = _dev()
    main = torch.cuda.current_stream(dev)
    got = streams.side_streams(dev, main)
    assert len(got) == 3 and len({s.cuda_stream for s in got} | {main.cuda_stream}) == 4
    assert streams.side_streams(dev, main) is got
    for role in (streams.ROLE_VIT, streams.ROLE_BRANCH, streams.ROLE_WGRAD):
        assert streams.side_stream(dev, main, role) == got[role]
        with torch.cuda.stream(got[role]):       # asked from ON a side stream: the set of the caller behind it
            assert streams.side_stream(dev, torch.cuda.current_stream(dev), streams.ROLE_WGRAD) == got[streams.ROLE_WGRAD]
    for s in got:
        assert streams._overlap(main, s) and streams._overlap(s, main)
    user = torch.cuda.Stream()
    other = streams.side_streams(dev, user)
    assert len(other) == 3 and user.cuda_stream not in {s.cuda_stream for s in other}
    torch.cuda.synchronize()


def test_weight_gradient_stream_accumulation_and_autograd_grad():
    """config.set_wgrad_stream (on by default): the convolutions' weight gradients are launched on a stream of their own and joined at
    the end of the backward pass.  (1) torch.autograd.grad() hands out tensors that are complete for the caller's stream; (2) a second
    backward pass WITHOUT zero_grad (gradient accumulation over micro-batches: the parameters already hold a .grad, AccumulateGrad adds
    on the caller's stream) gives exactly twice / the sum of the single-pass gradients -- that pass keeps the weight gradients on the
    caller's stream; (3) plain torch.optim.SGD reads the gradients right after backward().  All against the one-stream run, repeated
    so that an allocator reuse across streams would show."""
    import dose_prediction_amd
    from dose_prediction_amd.models.c3d import BaseUNet
    dev = _dev()
    _set(torch.float32)
    torch.manual_seed(21)
    net = BaseUNet(3, [-1, 8, 16, 16, 32, 32]).to(dev).train()
    xs = [torch.randn((2, 3, 16, 16, 32), generator=torch.Generator().manual_seed(30 + i)).to(dev) for i in range(2)]
    params = [p for p in net.parameters()]
    ref = {}
    try:
        for on in (False, True, True):
            dose_prediction_amd.config.set_wgrad_stream(on)
            # (1) autograd.grad
            gs = torch.autograd.grad(net(xs[0]).square().mean(), params)
            torch.cuda.synchronize()
            # (2) accumulation over two micro-batches
            net.zero_grad(set_to_none=True)
            for x in xs:
                net(x).square().mean().backward()
            torch.cuda.synchronize()
            acc = [p.grad.clone() for p in params]
            # (3) an optimizer that knows nothing about the stream
            w0 = [p.detach().clone() for p in params]
            opt = torch.optim.SGD(params, lr=0.5)
            net.zero_grad(set_to_none=True)
            net(xs[1]).square().mean().backward()
            opt.step()
            torch.cuda.synchronize()
            delta = [p.detach() - w for p, w in zip(params, w0)]
            with torch.no_grad():
                for p, w in zip(params, w0):
                    p.copy_(w)
            dose_prediction_amd.ops.invalidate_packs(params)          # (restored through copy_: the packed copies follow the version counters anyway)
            if not on:
                ref = {"gs": gs, "acc": acc, "delta": delta}
            else:
                for name, got in (("gs", gs), ("acc", acc), ("delta", delta)):
                    for a, b, p in zip(got, ref[name], params):
                        assert cmp_prefix(a.cpu(), b.cpu()) < 1e-4, (name, tuple(p.shape))
    finally:
        dose_prediction_amd.config.set_wgrad_stream(True)
