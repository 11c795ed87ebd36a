"""BASELINE.json full-size checks (2 x 128^3, the bench configuration) through size-independent properties: the CPU oracle
needs minutes per layer at this size, so these tests use identities that hold for any size --

  * adjointness: <conv_w(x), y> = <x, dgrad_w(y)> = <w, wgrad(x, y)>   (the three kernels of a layer are one bilinear form),
  * two independent implementations of the same op (LDS-tiled vs generic gather kernel) agree,
  * linearity of the convolution in x,
  * InstanceNorm output statistics (mean 0, variance 1 per sample and channel) and idempotence,
  * arg-max / one-hot: exactly one hot channel per voxel, idempotent,
  * the masked-L1 mean of a constant offset, Adam with zero gradient and zero weight decay leaves parameters unchanged,
  * one full DOSE-PYFER training step is finite and a few Adam steps reduce the loss.

All through the C ABI on the GPU; the oracle is not used here."""
import pytest
import torch

from helpers import rel_l2, rel_err
import oracle

pytestmark = pytest.mark.gpu
S = (128, 128, 128)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rand(shape, seed, dev, dtype):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(dev).to(dtype)


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("cfg", [(16, 16, 7, torch.float32), (32, 16, 7, torch.bfloat16), (16, 32, 3, torch.bfloat16), (16, 16, 3, torch.float32)])
def test_conv_adjoint_identities_128(cfg):
    from dose_prediction_amd import ops
    dev = _dev()
    cin, cout, k, dtype = cfg
    x = _rand((2,) + S + (cin,), 1, dev, dtype).requires_grad_(True)
    w = (_rand((cout, cin, k, k, k), 2, dev, torch.float32) * (cin * k ** 3) ** -0.5).requires_grad_(True)
    y = ops.conv3d(x, w, None, 1, k // 2, 1)
    r = _rand(tuple(y.shape), 3, dev, dtype)
    y.backward(r)
    lhs = _dot(y.float(), r.float())
    tol = 2e-5 if dtype == torch.float32 else 1e-2          # bf16: y, gx are rounded to 2^-9 before the dot products
    assert abs(lhs - _dot(x.float(), x.grad.float())) < tol * abs(lhs) + tol * 1e3
    wq = w if dtype == torch.float32 else w.bfloat16().float()      # the kernels multiply by the bf16-packed weights
    assert abs(lhs - _dot(wq, w.grad)) < tol * abs(lhs) + tol * 1e3


def test_conv_tiled_matches_generic_and_is_linear_128():
    from dose_prediction_amd import ops
    dev = _dev()
    dtype = torch.bfloat16
    x1, x2 = _rand((2,) + S + (16,), 4, dev, dtype), _rand((2,) + S + (16,), 5, dev, dtype)
    w = _rand((16, 16, 3, 3, 3), 6, dev, torch.float32) * 0.05
    b = _rand((16,), 7, dev, torch.float32)
    with torch.no_grad():
        yt = ops.conv3d(x1, w, b, 1, 1, 1)
        ops.USE_TILED = False
        try:
            yg = ops.conv3d(x1, w, b, 1, 1, 1)
        finally:
            ops.USE_TILED = True
        assert rel_l2(yt.float().cpu(), yg.float().cpu()) < 2e-3          # both round an fp32 sum to bf16; only the order differs
        # linearity in x (no bias): conv(x1 + x2) = conv(x1) + conv(x2) up to the bf16 rounding of the three outputs
        xs = (x1.float() + x2.float()).to(dtype)
        lhs = ops.conv3d(xs, w, None, 1, 1, 1).float()
        rhs = ops.conv3d(x1, w, None, 1, 1, 1).float() + ops.conv3d(x2, w, None, 1, 1, 1).float()
        assert rel_l2(lhs.cpu(), rhs.cpu()) < 1e-2


def test_instance_norm_statistics_and_idempotence_128():
    from dose_prediction_amd import ops
    dev = _dev()
    x = _rand((2,) + S + (16,), 8, dev, torch.float32) * 3.0 + 1.5
    with torch.no_grad():
        y = ops.norm_act(x, "instance")
        m = y.double().mean(dim=(1, 2, 3))
        v = y.double().var(dim=(1, 2, 3), unbiased=False)
        assert m.abs().max().item() < 1e-5 and (v - 1).abs().max().item() < 1e-4
        y2 = ops.norm_act(y, "instance")
        assert rel_l2(y2.cpu(), y.cpu()) < 1e-5


def test_argmax_onehot_properties_128():
    from dose_prediction_amd import ops
    dev = _dev()
    logits = _rand((2,) + S + (8,), 9, dev, torch.bfloat16)
    out = torch.zeros((2,) + S + (16,), dtype=torch.bfloat16, device=dev)
    labels = ops.argmax_onehot(logits, out, choff=1, labels=True)
    assert torch.equal(labels.long(), logits.float().argmax(-1))                     # bit-exact (ties cannot occur: random floats)
    hot = out[..., 1:8].float()
    assert torch.equal(hot.sum(-1), (labels != 0).float())                          # class 0 (background) is dropped by the glue
    out2 = torch.zeros_like(out)
    labels2 = ops.argmax_onehot(torch.cat((1 - hot.sum(-1, keepdim=True), hot), -1).to(torch.bfloat16), out2, choff=1, labels=True)
    assert torch.equal(labels2, labels) and torch.equal(out2, out)                   # idempotent


def test_masked_l1_and_adam_identities_128():
    from dose_prediction_amd import ops
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    g = _rand((2, 1) + S, 10, dev, torch.float32)
    mask = (_rand((2, 1) + S, 11, dev, torch.float32) > 0.3).float()
    assert abs(ops.masked_l1(g + 0.25, g, mask).item() - 0.25) < 1e-6                  # |p - g| == 0.25 everywhere
    p = torch.nn.Parameter(_rand((1 << 22,), 12, dev, torch.float32))
    before = p.detach().clone()
    opt = FusedAdam([p], lr=1e-3, weight_decay=0.0, amsgrad=True)
    p.grad = torch.zeros_like(p)
    opt.step()
    assert torch.equal(p.detach(), before)


def test_pyfer_training_steps_128_bf16():
    import dose_prediction_amd
    from dose_prediction_amd import losses, synth
    from dose_prediction_amd.models import dose_pyfer
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(4321)
        net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8,
                               num_heads=6, act="mish").to(dev).train()
        for prm in list(net.net_A.parameters()) + list(net.conv_out_A.parameters()):
            prm.requires_grad_(False)
        opt = FusedAdam([q for q in net.parameters() if q.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
        x, gt = synth.dose_input(2, S).to(dev), synth.dose_target(2, S).to(dev)
        hist = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            out = net(x)
            assert out[0].shape == (2, 1) + S and [tuple(o.shape[2:]) for o in out[1]] == [(128,) * 3, (64,) * 3, (32,) * 3, (16,) * 3]
            loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
            loss.backward()
            assert all(torch.isfinite(q.grad).all() for q in net.parameters() if q.grad is not None)
            opt.step()
            hist.append(loss.item())
        assert all(h == h and h < 1e4 for h in hist) and hist[-1] < hist[0], hist
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_pyfer_noncubic_192x192x128_step():
    """BASELINE.json configs[4] geometry (192 x 192 x 128, 1 152 tokens, batch 1 per GPU) in bf16: one training step runs, every
    output has the pyramid shape of dose_pyfer.py:360 and all gradients are finite (the reference's GenLoss.downSample assumes a cube;
    the device loss resamples each axis on its own)."""
    import dose_prediction_amd
    from dose_prediction_amd import losses, synth
    from dose_prediction_amd.models import dose_pyfer
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(1)
        sz = (192, 192, 128)
        net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=sz, num_layers=8,
                               num_heads=6, act="mish").to(dev).train()
        x, gt = synth.dose_input(1, sz).to(dev), synth.dose_target(1, sz).to(dev)
        out = net(x)
        assert [tuple(o.shape[2:]) for o in out[1]] == [sz, (96, 96, 64), (48, 48, 32), (24, 24, 16)] and out[0].shape == (1, 1) + sz
        loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=False)
        loss.backward()
        assert loss.item() == loss.item()
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


# ------------------------------------------------------------------------------------------------ sampled oracle at production size
# The property tests above never compare with the oracle at 2 x 128^3, where the tiled kernels take their production branches (XCD
# renumbering, 32-bit plane offsets, grid-quantised weight gradient, split-kd, W16 tiles).  Here the ORACLE's convolution
# (oracle.conv3d, float64) is evaluated on CPU-cropped receptive fields of sampled output voxels / on sampled weight-gradient taps.
def _sample_conv_ref(x_cpu, w_cpu, bias_cpu, pos, k):
    """oracle.conv3d at sampled output voxels: x_cpu NCDHW float64 (already padded by k//2), pos = [(n, d, h, w)] -> [len(pos), Cout]."""
    import oracle
    out = []
    for n, d, h, w in pos:
        patch = x_cpu[n:n + 1, :, d:d + k, h:h + k, w:w + k]
        out.append(oracle.conv3d(patch, w_cpu, bias_cpu).reshape(-1))
    return torch.stack(out)


def _positions(N, D, H, W, count, seed):
    g = torch.Generator().manual_seed(seed)
    pos = [(int(torch.randint(0, N, (1,), generator=g)), int(torch.randint(0, D, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)),
            int(torch.randint(0, W, (1,), generator=g))) for _ in range(count)]
    # corners, edges and tile seams (8-row / 32-position tiles, XCD slice boundaries at D/8) always included
    edge = [0, 1, 7, 8, 15, 16, 31, 32, 63, 64]
    for a in edge:
        for n in range(N):
            for (d, h, w) in ((a % D, 0, 0), (0, a % H, W - 1), (D - 1, H - 1, a % W), (a % D, a % H, a % W), (D - 1 - a % D, a % H, W - 1 - a % W)):
                pos.append((n, d, h, w))
    return pos


SAMPLED = [
    # (N, D, H, W, Cin (split ca | 0), Cout, k, dtype)
    (2, 128, 128, 128, 16, 0, 16, 7, torch.bfloat16),      # 7^3 16->16 (tap-paired), the dominant launch
    (2, 128, 128, 128, 32, 16, 16, 7, torch.bfloat16),     # 7^3 32->16 over a virtual concat (decoder1 first conv)
    (2, 128, 128, 128, 16, 0, 16, 3, torch.bfloat16),      # 3^3 16->16 at the 128^3 level
    (2, 128, 128, 128, 16, 0, 16, 7, torch.float32),       # the same launch geometry in the fp32 parity mode
    (2, 16, 16, 16, 256, 0, 128, 7, torch.bfloat16),       # split-kd / W16 tiles (decoder4 at 16^3)
    (2, 32, 32, 32, 128, 64, 64, 3, torch.bfloat16),       # two N tiles, virtual concat, 32^3
    # BASELINE.json configs[2] (OAR-TRANSEG 128^3 bf16 batch 2): the two launches no other configuration has
    (2, 128, 128, 128, 1, 0, 16, 3, torch.bfloat16, 16),   # encoder1's first 3^3 conv: ONE input channel in a 16-channel padded row (oar_transeg.py:92-100)
    (2, 128, 128, 128, 16, 0, 8, 1, torch.bfloat16),       # the 1x1x1 output head 16 -> 8 classes + bias (base_blocks.py:151-162)
    # row lengths other than 128: the 96^3 crop (OARSegmentation/config.py:24; four sliding windows per launch) and the
    # 192-wide volume of BASELINE configs[4], on the 96-position tiles of k_conv_cc16w
    (4, 96, 96, 96, 16, 0, 16, 7, torch.bfloat16),
    (1, 128, 192, 192, 32, 16, 16, 7, torch.bfloat16),
    (4, 96, 96, 96, 16, 0, 16, 3, torch.bfloat16),
]


@pytest.mark.parametrize("cfg", SAMPLED)
def test_conv_sampled_oracle_full_size(cfg):
    """Forward, data gradient and weight gradient of the hot launches vs oracle.conv3d on >= 4096 sampled voxels / 256 taps."""
    from dose_prediction_amd import ops
    dev = _dev()
    N, D, H, W, cin, ca, cout, k, dtype = cfg[:9]
    cpad = cfg[9] if len(cfg) > 9 else cin       # row pitch of the input: channels >= cin are another tensor's data (must be ignored)
    big = D >= 128
    nsample = 4096 if big else 1024
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    g = torch.Generator().manual_seed(17)
    x = torch.randn((N, D, H, W, cin), generator=g).to(dtype)
    w = (torch.randn((cout, cin, k, k, k), generator=g) * (cin * k ** 3) ** -0.5)
    bias = 0.1 * torch.randn((cout,), generator=g)
    r = torch.randn((N, D, H, W, cout), generator=g).to(dtype)
    if cpad > cin:
        xd = torch.cat((x, torch.randn((N, D, H, W, cpad - cin), generator=g).to(dtype)), -1).to(dev).requires_grad_(True)
    else:
        xd = x.to(dev).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    if ca:
        xa, xb = xd[..., :ca].detach().contiguous().requires_grad_(True), xd[..., ca:].detach().contiguous().requires_grad_(True)
        y = ops.conv3d((xa, xb), wd, bd, 1, k // 2, 1)
    else:
        y = ops.conv3d(xd, wd, bd, 1, k // 2, 1)
    y.backward(r.to(dev))
    torch.cuda.synchronize()
    gx = torch.cat((xa.grad, xb.grad), -1) if ca else xd.grad
    if cpad > cin:
        assert float(gx[..., cin:].abs().max()) == 0.0       # the padding channels receive no gradient
        gx = gx[..., :cin]
    wq = w if dtype == torch.float32 else w.to(dtype).float()           # the kernels multiply by the 16-bit packed weights
    p = k // 2
    pos = _positions(N, D, H, W, nsample, 5)
    idx = tuple(torch.tensor(c) for c in zip(*pos))
    # forward
    xp = torch.nn.functional.pad(x.double().permute(0, 4, 1, 2, 3), (p,) * 6)
    ref = _sample_conv_ref(xp, wq.double(), bias.double(), pos, k)
    got = y.detach().cpu().double()[idx]
    scale = ref.abs().max()
    assert (got - ref).abs().max() <= (tol if dtype == torch.float32 else 2 ** -8) * scale, ("fwd", ((got - ref).abs().max() / scale).item())
    assert ((got - ref).norm() / ref.norm()).item() < tol
    # data gradient = convolution of r with the transposed, flipped weights
    rp = torch.nn.functional.pad(r.double().permute(0, 4, 1, 2, 3), (p,) * 6)
    wt = wq.double().transpose(0, 1).flip(2, 3, 4).contiguous()
    refg = _sample_conv_ref(rp, wt, None, pos, k)
    gotg = gx.detach().cpu().double()[idx]
    assert ((gotg - refg).norm() / refg.norm()).item() < tol, "dgrad"
    assert (gotg - refg).abs().max() <= (tol if dtype == torch.float32 else 2 ** -8) * refg.abs().max() * 1.01
    # weight gradient at sampled (co, ci, kd, kh, kw): full-volume reductions of shifted products, float64
    gtap = torch.Generator().manual_seed(6)
    taps = [(int(torch.randint(0, cout, (1,), generator=gtap)), int(torch.randint(0, cin, (1,), generator=gtap)),
             int(torch.randint(0, k, (1,), generator=gtap)), int(torch.randint(0, k, (1,), generator=gtap)),
             int(torch.randint(0, k, (1,), generator=gtap))) for _ in range(96 if big else 256)]
    taps += [(0, 0, 0, 0, 0), (cout - 1, cin - 1, k - 1, k - 1, k - 1), (cout - 1, 0, 0, k - 1, 0), (0, cin - 1, k // 2, k // 2, k // 2)]
    rd = r.double()
    refw = []
    for co, ci, kd, kh, kw in taps:
        sl = xp[:, ci, kd:kd + D, kh:kh + H, kw:kw + W]
        refw.append((sl * rd[..., co]).sum())
    refw = torch.stack(refw)
    gotw = torch.stack([wd.grad[t].double().cpu() for t in taps])
    wtol = 5e-5 if dtype == torch.float32 else 2e-3      # (fp32 products of 16-bit operands are exact; only the fp32 accumulation order differs)
    assert ((gotw - refw).norm() / refw.norm()).item() < wtol, ("wgrad", ((gotw - refw).norm() / refw.norm()).item())
    # bias gradient: column sums of r
    refb = rd.sum(dim=(0, 1, 2, 3))
    assert ((bd.grad.double().cpu() - refb).norm() / refb.norm()).item() < 1e-4


def test_c5_cascade_fp16_checkpointing_192x192x128():
    """BASELINE.json configs[4] as ONE configuration (per-GPU share: batch 1): frozen OAR-TRANSEG sliding-window inference (roi 96^3,
    overlap 0.25, sw_batch 4 -> 3 x 3 x 2 windows) -> arg-max / one-hot glue with axis reversal (train_light_linked_model.py:143-167)
    -> DOSE-PYFER in fp16 storage with loss scaling and activation checkpointing of the four decoder blocks -> GenLoss -> backward
    -> fused Adam; three steps.  Finite everywhere, the loss falls, checkpointing lowers the peak memory, BatchNorm counters advance
    once per step (not twice: the recomputation must not update the running statistics)."""
    import dose_prediction_amd
    from dose_prediction_amd import cascade, losses, synth
    from dose_prediction_amd.models import dose_pyfer, oar_transeg
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    vol, roi = (192, 192, 128), (96, 96, 96)
    dose_prediction_amd.set_compute_dtype(torch.float16)
    dose_prediction_amd.set_loss_scale(1024.0)
    try:
        assert [len(a) for a in cascade.window_starts(vol, roi)] == [3, 3, 2]
        torch.manual_seed(8765)
        seg = oar_transeg.Model(in_channels=1, out_channels=8, img_size=roi, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
                                pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True).to(dev).eval()
        torch.manual_seed(4321)
        net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=vol[::-1], num_layers=8,
                               num_heads=6, act="mish").to(dev).train()
        for n, p in net.named_parameters():
            if "net_A" in n or "conv_out_A" in n:
                p.requires_grad = False
        opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
        full = synth.dose_input(1, vol[::-1]).to(dev)                       # (dose-loader orientation W, H, D)
        ct = full[:, 8:9].permute(0, 1, 4, 3, 2).contiguous()              # the segmentation loader's orientation
        ptv = full[:, 0:1].contiguous()
        gt = synth.dose_target(1, vol[::-1]).to(dev)
        structures, labels = cascade.cascade_structures(seg, ct, ptv, roi_size=roi, sw_batch_size=4, overlap=0.25)
        assert structures.shape == (1, 9) + vol[::-1] and labels.shape == (1,) + vol
        # CT / PTV pass through (rounded once to the fp16 staging buffer)
        assert torch.equal(structures[:, 8], full[:, 8].half().float()) and torch.equal(structures[:, 0], full[:, 0].half().float())
        onehot = structures[:, 1:8]
        assert set(onehot.unique().tolist()) <= {0.0, 1.0} and float(onehot.sum(1).max()) <= 1.0
        bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm3d)]

        def step():
            opt.zero_grad(set_to_none=True)
            out = net(structures)
            loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
            opt.step()
            return loss.item()

        peaks, hist = {}, [step()]        # (first step: allocates the 3 x 650 MB of Adam state, which must not count against either mode)
        for ckpt in (False, True):
            dose_prediction_amd.set_activation_checkpointing(ckpt)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            n0 = int(bn[0].num_batches_tracked)
            for _ in range(1 if not ckpt else 3):
                hist.append(step())
            torch.cuda.synchronize()
            peaks[ckpt] = torch.cuda.max_memory_allocated() / 2 ** 30
            assert int(bn[0].num_batches_tracked) - n0 == (1 if not ckpt else 3)      # recomputation must not advance the BN counters
        print(f"[c5] losses {hist}; peak memory {peaks[False]:.2f} GiB without / {peaks[True]:.2f} GiB with decoder checkpointing")
        assert all(h == h and h < 1e4 for h in hist) and hist[-1] < hist[0], hist
        assert peaks[True] < peaks[False], peaks      # (the saving is the intermediates of decoder2..4: decoder1's recomputation sets the peak)
        assert all(torch.isfinite(p).all() for p in net.parameters())
    finally:
        dose_prediction_amd.set_activation_checkpointing(False)
        dose_prediction_amd.set_loss_scale(1.0)
        dose_prediction_amd.set_compute_dtype(torch.float32)


# ------------------------------------------------------------------------------------------------ BASELINE.json configs[2] and [3]
def _transeg(dev, shape, seed=8765):
    from dose_prediction_amd.models import oar_transeg
    torch.manual_seed(seed)
    # hyper-parameters: OARSegmentation/train_light_transeg.py:110-124 (oar_transeg.py:20-35)
    return oar_transeg.Model(in_channels=1, out_channels=8, img_size=shape, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
                             pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True).to(dev)


@pytest.mark.parametrize("mode", [torch.bfloat16, "fp32x3"])
def test_transeg_training_steps_at_the_reference_crop_96(mode):
    """OAR-TRANSEG as the reference itself trains it (OARSegmentation/config.py:24: IMAGE_SIZE = 96; train_light_transeg.py:113, 148): 96^3
    crops (a batch of 4, the sliding-window batch of config.py:31), DiceCE loss, backward, fused Adam -- the row lengths 96 / 48 / 24 / 12 / 6
    through every kernel family (96-position 7^3 tiles, least-padding wave columns, W16 tiles, generic kernels at 6^3) in one network:
    three steps, every gradient finite, the loss falls; in fp32x3 the first forward pass is also checked against the float64 oracle on one
    crop (logits within 1e-3, arg-max exact off near-ties)."""
    import dose_prediction_amd
    from dose_prediction_amd import synth
    from dose_prediction_amd.losses import DiceCELoss
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    S96 = (96, 96, 96)
    dose_prediction_amd.set_compute_dtype(mode)
    try:
        net = _transeg(dev, S96).train()
        x = synth.ct_input(4, S96).to(dev)
        if mode == "fp32x3":
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            with torch.no_grad():
                ref = oracle.oar_transeg(sd, x[:1].cpu(), num_heads=12, training=True)
                got = net(x[:1]).float().cpu()
            assert rel_err(got, ref) < 1e-3
            top2 = ref.topk(2, dim=1).values
            safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()
            assert int(((got.argmax(1) != ref.argmax(1)) & safe).sum()) == 0
            net.load_state_dict(sd)
        opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=1e-5)
        lab = torch.randint(0, 8, (4, 1) + S96, generator=torch.Generator().manual_seed(5678)).float().to(dev)
        loss_fn = DiceCELoss(to_onehot_y=True, softmax=True)
        hist = []
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            logits = net(x)
            assert logits.shape == (4, 8) + S96 and logits.dtype == torch.float32
            loss = loss_fn(logits, lab)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
            opt.step()
            hist.append(loss.item())
        print(f"[transeg 96^3 {mode}] DiceCE {hist}")
        assert all(h == h and h < 1e3 for h in hist) and hist[-1] < hist[0], hist
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_c3_transeg_training_steps_128_bf16():
    """BASELINE.json configs[2]: OAR-TRANSEG at 128^3, bf16 storage, batch 2 on one GPU -- forward (logits [2, 8, 128^3]) + DiceCE
    loss (train_light_transeg.py:148) + backward + fused Adam, four steps: every gradient finite, every parameter that the graph reaches receives one, the loss
    falls.  (The 1 -> 16 first convolution and the 16 -> 8 head are checked against the oracle at this size in
    test_conv_sampled_oracle_full_size; 12-head d = 64 attention at B = 2, N = 512 in test_ops_gpu.test_fused_attention_full_size.)"""
    import dose_prediction_amd
    from dose_prediction_amd import synth
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    try:
        net = _transeg(dev, S).train()
        opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=1e-5)
        x = synth.ct_input(2, S).to(dev)
        lab = torch.randint(0, 8, (2, 1) + S, generator=torch.Generator().manual_seed(5678)).float().to(dev)
        from dose_prediction_amd.losses import DiceCELoss
        seg_loss = DiceCELoss(to_onehot_y=True, softmax=True)          # the reference's loss: train_light_transeg.py:148
        hist = []
        for it in range(4):
            opt.zero_grad(set_to_none=True)
            logits = net(x)
            assert logits.shape == (2, 8) + S and logits.dtype == torch.float32
            loss = seg_loss(logits, lab)
            loss.backward()
            if it == 0:
                missing = [k for k, p in net.named_parameters() if p.grad is None]
                # parameters the reference graph never reaches: cls_token, and conv3 / norm3 of the Cin == Cout residual blocks
                assert all(("cls_token" in k) or (".conv3." in k) for k in missing), missing
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
            opt.step()
            hist.append(loss.item())
        print(f"[c3] DiceCE {hist}")
        assert all(h == h and h < 1e3 for h in hist) and hist[-1] < hist[0], hist
        assert not opt.found_inf()
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_c4_cascade_training_step_128_bf16():
    """BASELINE.json configs[3], the per-GPU share (batch 2 of the global 16): OAR-TRANSEG forward without gradients at 128^3 ->
    arg-max / one-hot glue with axis reversal, staged in place (train_light_linked_model.py:143-167) -> DOSE-PYFER forward_staged ->
    GenLoss -> backward -> fused Adam, bf16 storage.  One-hot validity, CT / PTV pass-through, finite gradients, falling loss; the
    segmentation network receives no gradient and its weights do not move."""
    import dose_prediction_amd
    from dose_prediction_amd import cascade, losses, synth
    from dose_prediction_amd.models import dose_pyfer
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    try:
        seg = _transeg(dev, S).eval()
        torch.manual_seed(4321)
        net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8,
                               num_heads=6, act="mish").to(dev).train()
        for n, p in net.named_parameters():
            if "net_A" in n or "conv_out_A" in n:
                p.requires_grad = False
        opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
        full = synth.dose_input(2, S).to(dev)
        ct = full[:, 8:9].permute(0, 1, 4, 3, 2).contiguous()              # the segmentation loader's axis order
        ptv = full[:, 0:1].contiguous()
        gt = synth.dose_target(2, S).to(dev)
        seg_before = [p.detach().clone() for p in list(seg.parameters())[:4]]
        # the fp32 view of the staged input (what LinkedNet hands to the dose network, lines 165-167)
        structures, labels = cascade.cascade_structures(seg, ct, ptv)
        assert structures.shape == (2, 9) + S and labels.shape == (2,) + S and labels.dtype == torch.int32
        assert int(labels.min()) >= 0 and int(labels.max()) <= 7
        assert torch.equal(structures[:, 8], full[:, 8].bfloat16().float()) and torch.equal(structures[:, 0], full[:, 0].bfloat16().float())
        onehot = structures[:, 1:8]
        assert set(onehot.unique().tolist()) <= {0.0, 1.0} and float(onehot.sum(1).max()) <= 1.0
        # one-hot channel c is exactly (label == c) in the dose loader's axis order
        lab_dose = labels.permute(0, 3, 2, 1)
        for c in range(1, 8):
            assert torch.equal(onehot[:, c - 1] > 0, lab_dose == c), c
        hist = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            staged, _ = cascade.cascade_structures(seg, ct, ptv, staged=True)
            assert staged.shape == (2,) + S + (16,) and staged.dtype == torch.bfloat16
            out = net.forward_staged(staged)
            loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
            opt.step()
            hist.append(loss.item())
        print(f"[c4] losses {hist}")
        assert all(h == h and h < 1e4 for h in hist) and hist[-1] < hist[0], hist
        assert all(p.grad is None for p in seg.parameters())
        assert all(torch.equal(a, b.detach()) for a, b in zip(seg_before, list(seg.parameters())[:4]))
        # the staged buffer the training step consumed holds the same nine channels as the fp32 view (CT / PTV exactly; the masks up to
        # arg-max near-ties: the segmentation forward -- fp32x3 inside the cascade, config.cascade_seg_mode() -- is not bitwise
        # reproducible, its 16^3 layers accumulate with fp32 atomics; in bf16 storage 0.27 % of the one-hot entries moved between two runs)
        st9 = staged[..., :9].permute(0, 4, 1, 2, 3).float()
        assert torch.equal(st9[:, 0], structures[:, 0]) and torch.equal(st9[:, 8], structures[:, 8])
        flips = float((st9[:, 1:8] != structures[:, 1:8]).float().mean())
        print(f"[c4] one-hot entries that differ between two segmentation passes: {flips:.2e}")
        assert flips < 2e-4
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_pyfer_128_one_rank_rccl_every_bucket_is_exchanged_inside_backward():
    """The production bucket layout (DOSE-PYFER's 162.6 M trainable parameters, 32 MiB buckets, the 78.6 M-element patch embedding
    reduced in place) on one rank over RCCL: from the second backward pass on EVERY bucket is launched from the gradient hooks, i.e.
    inside the backward pass (DESIGN section 7's overlap claim, so far asserted on gloo / CPU only), no deferred weight gradient is
    left pending when a bucket goes out, and the step still trains."""
    import os
    import torch.distributed as dist
    import dose_prediction_amd
    from dose_prediction_amd import losses, ops, synth
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    from dose_prediction_amd.models import dose_pyfer
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(4321)
        net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8,
                               num_heads=6, act="mish").to(dev).train()
        for n, p in net.named_parameters():
            if "net_A" in n or "conv_out_A" in n:
                p.requires_grad = False
        red = attach_gradient_allreduce(net, bucket_mb=32.0)
        nb = len(red.buckets)
        assert nb >= 10 and any(red.inplace)
        opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
        x, gt = synth.dose_input(2, S).to(dev), synth.dose_target(2, S).to(dev)
        hist = []
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            before = dict(red.stats)
            loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
            loss.backward()
            assert ops.deferred_pending() == 0
            if it >= 1:
                assert red.stats["launched_in_backward"] - before["launched_in_backward"] == nb, (it, before, red.stats, nb)
                assert red.stats["launched_at_end"] == before["launched_at_end"]
            opt.step()
            hist.append(loss.item())
        assert all(h == h for h in hist) and hist[-1] < hist[0], hist
        red.close()
    finally:
        dist.destroy_process_group()
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_pyfer_training_steps_128_fp32x3():
    """BASELINE.json configs[1] geometry in the fp32x3 mode (fp32 storage, split-bf16 matrix-core arithmetic): three full training
    steps at 2 x 128^3 -- the production branches of the x3 path (DP_X3 launches of k_conv_cc16 / k_conv_tiled with x_hi slabs swept
    twice, K-along-H weight gradients on 2 Cin x Cout problems, split operands written by the normalisation kernels, grouped
    Linear gradients with the K-stacked bias selector, packed copies written by Adam) run, stay finite and reduce the loss; the first
    step's loss equals the bf16 run's to storage precision (same weights, same data)."""
    import dose_prediction_amd
    from dose_prediction_amd import losses, synth
    from dose_prediction_amd.models import dose_pyfer
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    first = {}
    try:
        for mode in ("fp32x3", torch.bfloat16):
            dose_prediction_amd.set_compute_dtype(mode)
            torch.manual_seed(4321)
            net = dose_pyfer.Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8,
                                   num_heads=6, act="mish").to(dev).train()
            for prm in list(net.net_A.parameters()) + list(net.conv_out_A.parameters()):
                prm.requires_grad_(False)
            opt = FusedAdam([q for q in net.parameters() if q.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
            x, gt = synth.dose_input(2, S).to(dev), synth.dose_target(2, S).to(dev)
            hist = []
            for _ in range(3 if mode == "fp32x3" else 1):
                opt.zero_grad(set_to_none=True)
                out = net(x)
                loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
                loss.backward()
                assert all(torch.isfinite(q.grad).all() for q in net.parameters() if q.grad is not None)
                opt.step()
                hist.append(loss.item())
            first[str(mode)] = hist[0]
            if mode == "fp32x3":
                assert all(h == h and h < 1e4 for h in hist) and hist[-1] < hist[0], hist
                assert not opt.found_inf()
            del net, opt
            torch.cuda.empty_cache()
        a, b = first["fp32x3"], first[str(torch.bfloat16)]
        assert abs(a - b) < 5e-2 * abs(a), first
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)
