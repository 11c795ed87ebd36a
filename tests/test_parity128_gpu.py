"""End-to-end parity AT THE BENCHMARK SIZE (north_star: "outputs match the reference CPU path within 1e-3 rel on
the predicted dose map and bit-exact on OAR argmax masks").  The oracle (the CPU restatement of dose_pyfer.py:355-360 /
oar_transeg.py:171-185, pinned by tests/golden) runs ONE fp32 forward on a real 128^3 synthetic OpenKBP-like volume (about 12 s on
the GPU box's host cores) and the HIP networks with the same weights run the same volume through the C ABI.  These are the
production-only branches no 64^3 test reaches as a network: XCD renumbering over 8 depth ranges, 512-token attention, the split-K
patch-embedding GEMM at K = 102 400, the cc16 kernels at W = 128."""
import os

import pytest
import torch

import oracle
from helpers import rel_err

pytestmark = pytest.mark.gpu

FULL = (128, 128, 128)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_mode():
    import dose_prediction_amd
    yield
    dose_prediction_amd.set_compute_dtype(torch.float32)


@pytest.fixture(scope="module")
def pyfer_case():
    from dose_prediction_amd import synth
    from dose_prediction_amd.models.dose_pyfer import Model
    torch.manual_seed(4321)
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=FULL, num_layers=8, num_heads=6,
                act="mish", mode_multi_dec=True, multiS_conv=True)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    del net
    x, gt = synth.dose_input(1, FULL), synth.dose_target(1, FULL)
    torch.set_num_threads(min(os.cpu_count() or 8, 128))
    with torch.no_grad():
        out = oracle.dose_pyfer(sd, x, num_layers=8, num_heads=6, act="mish", training=True)
    return sd, x, gt, out[0].detach(), [o.detach() for o in out[1]]


def _hip_pyfer(sd, x, mode):
    import dose_prediction_amd
    from dose_prediction_amd.models.dose_pyfer import Model
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(mode)
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=FULL, num_layers=8, num_heads=6,
                act="mish", mode_multi_dec=True, multiS_conv=True)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    with torch.no_grad():
        out = net(x.to(dev))
    res = out[0].float().cpu(), [o.float().cpu() for o in out[1]]
    del net, out
    torch.cuda.empty_cache()
    return res


def test_pyfer_128_fp32x3_matches_oracle(pyfer_case):
    """The fast tolerance-meeting mode at the benchmark size: all four dose maps AND net_A's map within 1e-3 (max|d| / max|ref|)."""
    sd, x, gt, ref_a, ref = pyfer_case
    got_a, got = _hip_pyfer(sd, x, "fp32x3")
    errs = [rel_err(g, r) for g, r in zip(got, ref)]
    ea = rel_err(got_a, ref_a)
    mask = gt[:, 1:2] > 0
    mae = float(70.0 * (got[0] - ref[0]).abs()[mask].mean())
    print(f"[parity128] pyfer fp32x3: rel-err {['%.2e' % e for e in errs]}, net_A map {ea:.2e}, dose-MAE {mae:.2e} Gy")
    assert max(errs) < 1e-3, errs
    assert ea < 1e-3, ea
    assert mae < 7e-2     # 1e-3 of the 70 Gy prescription scale


def test_pyfer_128_exact_fp32_matches_oracle(pyfer_case):
    """The exact-fp32 MFMA mode at the benchmark size (two fp32 evaluations of one graph: ~1e-5)."""
    sd, x, gt, ref_a, ref = pyfer_case
    got_a, got = _hip_pyfer(sd, x, torch.float32)
    errs = [rel_err(g, r) for g, r in zip(got, ref)]
    print(f"[parity128] pyfer exact fp32: rel-err {['%.2e' % e for e in errs]}")
    assert max(errs) < 1e-3, errs
    assert rel_err(got_a, ref_a) < 1e-3


def test_pyfer_128_bf16_is_within_the_storage_budget_band(pyfer_case):
    """The benchmark mode at the benchmark size.  bf16 STORAGE cannot meet 1e-3; what is asserted is that the
    128^3 error sits in the band the 64^3 emulated-storage budget predicts (tests/test_precision_budget_gpu.py gates the 64^3 figure at
    1.5 x the emulated oracle: 4.8e-2 / 2.66 Gy) -- a production-only kernel branch computing something else would be O(1)."""
    sd, x, gt, ref_a, ref = pyfer_case
    got_a, got = _hip_pyfer(sd, x, torch.bfloat16)
    errs = [rel_err(g, r) for g, r in zip(got, ref)]
    mask = gt[:, 1:2] > 0
    mae = float(70.0 * (got[0] - ref[0]).abs()[mask].mean())
    print(f"[parity128] pyfer bf16: rel-err {['%.2e' % e for e in errs]}, dose-MAE {mae:.2e} Gy")
    assert max(errs) < 0.15, errs
    assert mae < 6.0


@pytest.fixture(scope="module")
def transeg_case():
    from dose_prediction_amd import synth
    from dose_prediction_amd.models import oar_transeg
    torch.manual_seed(8765)
    net = oar_transeg.Model(in_channels=1, out_channels=8, img_size=FULL, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
                            pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    del net
    x = synth.ct_input(1, FULL)
    with torch.no_grad():
        ref = oracle.oar_transeg(sd, x, num_heads=12, training=True)
    return sd, x, ref.detach()


def _hip_transeg(sd, x, mode):
    import dose_prediction_amd
    from dose_prediction_amd.models import oar_transeg
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(mode)
    net = oar_transeg.Model(in_channels=1, out_channels=8, img_size=FULL, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
                            pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    with torch.no_grad():
        got = net(x.to(dev)).float().cpu()
    del net
    torch.cuda.empty_cache()
    return got


@pytest.mark.parametrize("mode", ["fp32x3", torch.float32])
def test_transeg_128_argmax_exact_off_near_ties(transeg_case, mode):
    """OAR-TRANSEG at 128^3: logits within 1e-3 of the oracle and the arg-max masks bit-exact wherever the oracle's own top-2 margin
    exceeds 1e-3 of the logit range (two fp32 evaluations of one graph differ by ~1e-5 and may legitimately flip exact near-ties)."""
    sd, x, ref = transeg_case
    got = _hip_transeg(sd, x, mode)
    e = rel_err(got, ref)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()
    mism = got.argmax(1) != ref.argmax(1)
    print(f"[parity128] transeg {mode}: logit rel-err {e:.2e}, arg-max mismatches {int(mism.sum())} of {mism.numel()} "
          f"({int((mism & safe).sum())} off near-ties; {int((~safe).sum())} near-tie voxels)")
    assert e < 1e-3
    assert int((mism & safe).sum()) == 0
    assert int(mism.sum()) <= int((~safe).sum())


# ------------------------------------------------------------------------------------------------ backward at the benchmark size (round 6)
BWD_KEYS = [
    "net_B.dose_convertors.0.0.weight",                                         # the full-resolution head (1x1x1, 16 -> 1)
    "net_B.decoder.decoder1.conv_block.cov_.conv_7.0.conv.0.weight",           # 7^3 32 -> 16 at 128^3 on the virtual concat: k_wgrad_hk<7>
    "net_B.decoder.decoder1.conv_block.cov_.conv_7.0.conv.3.weight",           # 7^3 16 -> 16 at 128^3
    "net_B.decoder.decoder1.conv_block.cov_.conv_3.0.conv.0.weight",           # 3^3 32 -> 16 at 128^3: k_wgrad_hk3
    "net_B.decoder.decoder1.conv_block.cov_.conv.0.weight",                    # the 1x1x1 mixer: k_wgrad_rows
    "net_B.decoder.decoder1.transp_conv.conv.weight",                          # ConvTranspose 32 -> 16, 64^3 -> 128^3
    "net_B.decoder.decoder2.conv_block.cov_.conv_7.0.conv.0.weight",           # 7^3 64 -> 32 at 64^3
    "net_B.encoder.skip1.layer.conv1.conv.weight",                             # 3^3 25 -> 16 at 128^3 (the first layer of the 128^3 skip block)
    "net_B.encoder.vit.blocks.0.attn.qkv.weight",                              # the first transformer layer: grouped TN GEMM
    "net_B.encoder.vit.patch_embedding.patch_embeddings.1.weight",             # K = 102 400 split-K weight gradient, the LAST gradient of the pass
]


def oracle_backward_128(sd, x, gt, keys):
    """ONE fp32 oracle forward + GenLoss + backward of DOSE-PYFER on a 128^3 volume on the host (net_A frozen as in
    train_light_pyfer.py:85-88; only `keys` require gradients, so the CPU skips every other weight gradient): loss, gradients."""
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        sdg[k].requires_grad_(True)
    out = oracle.dose_pyfer(sdg, x, num_layers=8, num_heads=6, act="mish", training=True)
    loss = oracle.gen_loss(out, gt, 10, 1, casecade=True, freez=True)
    loss.backward()
    return float(loss.detach()), {k: sdg[k].grad.detach() for k in keys}


def test_pyfer_128_fp32x3_backward_matches_oracle(pyfer_case):
    """Backward parity AT THE BENCHMARK SIZE: one training step's gradients (GenLoss, network_trainer.py:200-213) of
    B = 1 x 128^3 in fp32x3 against ONE oracle backward() on the host, for ten weight-gradient tensors that together cover the
    production-only backward branches no 64^3 test reaches end to end: k_wgrad_hk<7> / k_wgrad_hk3 at the full plane count, the XCD
    renumbering of the 128^3 data-gradient launches that every one of these gradients has passed through, the K = 102 400 split-K
    patch-embedding gradient (the last gradient of the pass: everything upstream of it is in it), the grouped transformer launch.
    Tolerance 2.5e-2 relative L2 per tensor: the default fp32x3 backward is ONE bf16 product per contraction (the
    whole gradient vector sits at 1.07e-2 of float64 at 64^3, single Linear tensors at 1.6-1.85e-2, and the exact-fp32 mode itself at
    4.2e-3 -- ReLU / LeakyReLU gates of pre-activations within round-off of zero); measured here (round 6, two boxes): 2.5e-5 (the head)
    .. 1.0e-2 (the patch embedding, the last gradient of the pass), loss within 7e-7.  A wrong tile decode, a dropped halo or a lost
    split-K share is an O(1) error.  The loss value is compared too (1e-4)."""
    import dose_prediction_amd
    from dose_prediction_amd import losses
    from dose_prediction_amd.models.dose_pyfer import Model
    import time
    sd, x, gt, _, _ = pyfer_case
    dev = _dev()
    t0 = time.time()
    ref_loss, ref = oracle_backward_128(sd, x, gt, BWD_KEYS)
    t_or = time.time() - t0
    dose_prediction_amd.set_compute_dtype("fp32x3")
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=FULL, num_layers=8, num_heads=6,
                act="mish", mode_multi_dec=True, multiS_conv=True)
    net.load_state_dict(sd)
    for n, p in net.named_parameters():
        if "net_A" in n or "conv_out_A" in n:
            p.requires_grad = False
    net = net.to(dev).train()
    out = net(x.to(dev))
    loss = losses.gen_loss(out, gt.to(dev), 10.0, 1.0, casecade=True, freez=True)
    loss.backward()
    torch.cuda.synchronize()
    named = dict(net.named_parameters())
    errs = {}
    for k in BWD_KEYS:
        g, r = named[k].grad.detach().double().cpu().reshape(-1), ref[k].double().reshape(-1)
        errs[k] = float((g - r).norm() / r.norm())
    got_loss = float(loss.detach())
    print(f"[parity128] backward fp32x3 vs oracle ({t_or:.0f} s on the host): loss {got_loss:.6f} vs {ref_loss:.6f}; "
          + ", ".join(f"{k.split('net_B.')[-1]} {e:.2e}" for k, e in errs.items()))
    del net, out, loss
    torch.cuda.empty_cache()
    assert abs(got_loss - ref_loss) < 1e-4 * abs(ref_loss), (got_loss, ref_loss)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < 2.5e-2, worst
