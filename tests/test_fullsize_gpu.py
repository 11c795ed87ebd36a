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

from helpers import rel_l2

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
