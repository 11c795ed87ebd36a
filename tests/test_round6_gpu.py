"""Round 6: ADVICE r5 regressions (ticket counters of captured launches, deterministic LayerNorm on wide rows, DiceCE labels outside the
class range, nested backward passes and the per-pass zero arena) on the GPU, through the C ABI."""
import warnings

import pytest
import torch

import oracle
from helpers import rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# ------------------------------------------------------------------------------------------------ DiceCE, labels outside [0, C)
@pytest.mark.parametrize("label_dtype", [torch.int64, torch.uint8, torch.float32])
def test_dice_ce_backward_matches_forward_for_labels_outside_the_class_range(label_dtype):
    """A label outside [0, C) contributes no cross-entropy term and matches no Dice class in the forward kernel; the backward kernel used to
    add the CE gradient for such voxels all the same.  Reference for the semantics the kernel implements: the oracle's formula
    with the CE sum restricted to the valid voxels (still divided by B * V) and an all-zero one-hot row for the others, in float64 with
    autograd.  (torch / MONAI raise on such labels; accepting them is a documented divergence, it must at least be self-consistent.)"""
    from dose_prediction_amd import ops
    dev = _dev()
    B, C, S = 2, 8, (6, 7, 9)
    z = rnd((B, C) + S, 1, 2.0)
    lab = torch.randint(0, C, (B, 1) + S, generator=torch.Generator().manual_seed(2))
    bad = torch.rand((B, 1) + S, generator=torch.Generator().manual_seed(3)) < 0.2
    lab_bad = torch.where(bad, torch.full_like(lab, 200), lab)
    zr = z.double().requires_grad_(True)
    p = torch.softmax(zr, 1)
    onehot = torch.nn.functional.one_hot(lab[:, 0], C).movedim(-1, 1).double() * (~bad).double()
    inter, den = (onehot * p).sum((2, 3, 4)), onehot.sum((2, 3, 4)) + p.sum((2, 3, 4))
    dice = (1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)).mean()
    logp = torch.log_softmax(zr, 1)
    ce = -(logp.gather(1, lab) * (~bad).double()).sum() / (B * S[0] * S[1] * S[2])
    ref = dice + ce
    (3.0 * ref).backward()
    zh = z.to(dev).requires_grad_(True)
    got = ops.dice_ce(zh, lab_bad.to(dev, label_dtype))
    (3.0 * got).backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 2e-6 * abs(float(ref.detach()))
    assert rel_l2(zh.grad.cpu(), zr.grad) < 2e-6 and rel_err(zh.grad.cpu(), zr.grad) < 1e-5
    # and the gradient of an ignored voxel sums to zero over the classes in BOTH halves (softmax Jacobian), with no -1 at a phantom class
    assert float(zh.grad.sum(1).abs().max()) < 1e-7


# ------------------------------------------------------------------------------------------------ folded finalize: counters of captured launches
def test_folded_finalize_inside_a_capture_uses_counters_of_its_own():
    """dp_stats_partial_finalize (one launch, last-block ticket) recorded into a HIP graph: the captured launch gets a counter from the
    capture region -- never handed out again, so no later eager launch can draw tickets from it while a replay is in flight --
    and every replay, interleaved with thousands of eager folded launches that wrap the eager ring, produces the eager result bit for bit."""
    from dose_prediction_amd import _lib, ops
    dev = _dev()
    L = _lib.lib()
    if not L.dp_ticket_enabled():
        pytest.skip("DP_NO_TICKET set")
    N, V, C = 2, 40000, 16
    x = rnd((N, V, C), 5).to(dev).to(torch.bfloat16)
    nblk = L.dp_stats_nblk(V)

    def run(stream, bufs=None):
        part = torch.empty((N, nblk, 2, C), dtype=torch.float32, device=dev) if bufs is None else bufs[0]
        mean = torch.empty((N, C), dtype=torch.float32, device=dev) if bufs is None else bufs[1]
        rstd = torch.empty((N, C), dtype=torch.float32, device=dev) if bufs is None else bufs[2]
        rc = _lib.call("dp_stats_partial_finalize", x.data_ptr(), C, N, V, C, part.data_ptr(), 0, 1e-5, mean.data_ptr(), rstd.data_ptr(), 0, 0, 0.1,
                       1, stream)
        return rc, (part, mean, rstd)
    rc, ref = run(torch.cuda.current_stream().cuda_stream)          # (also allocates the counters: a first use inside a capture declines)
    assert rc == 0
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bufs = tuple(torch.zeros_like(t) for t in ref)
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side):
        rc_cap, _ = run(torch.cuda.current_stream().cuda_stream, bufs)
    assert rc_cap == 0, "the folded form must stay available inside a capture (counters from the capture region)"
    for rep in range(3):
        g.replay()
        for _ in range(3000):                                        # > 8192 eager counters in total: the eager ring wraps under the replays
            rc, out = run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(bufs[1], ref[1]) and torch.equal(bufs[2], ref[2]), rep
        assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2]), rep
        bufs[1].zero_()
        bufs[2].zero_()


# ------------------------------------------------------------------------------------------------ deterministic switch, wide LayerNorm rows
def test_deterministic_switch_keeps_wide_layernorm_rows_working():
    """set_deterministic(True) with a hidden size above 1024 (the fixed-order LayerNorm backward kernel keeps a row in registers: C <= 1024):
    the backward pass falls back to the atomic dgamma / dbeta accumulation with ONE warning instead of raising, and the
    results match the oracle; C = 768 (the reference's hidden size) still takes the deterministic kernel (bit-identical passes)."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.float32)
    try:
        dose_prediction_amd.config.set_deterministic(True)
        ops._LN_DET_WARNED[0] = False
        for C in (1536, 768):
            x, a = rnd((2, 33, C), 1), rnd((2, 33, C), 2)
            gam, bet, r = 1 + 0.1 * rnd((C,), 3), 0.1 * rnd((C,), 4), rnd((2, 33, C), 5)
            xr, ar, gr, br = (t.double().requires_grad_(True) for t in (x, a, gam, bet))
            zr = torch.nn.functional.layer_norm(xr + ar, (C,), gr, br, 1e-5)
            zr.backward(r.double())
            outs = []
            for _ in range(2):
                xd, ad, gd, bd = (t.to(dev).requires_grad_(True) for t in (x, a, gam, bet))
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    s, z = ops.add_layer_norm(xd, ad, gd, bd)
                    z.backward(r.to(dev))
                outs.append((z.detach(), xd.grad, gd.grad, bd.grad, len(w)))
            z, gx, gg, gb, _ = outs[0]
            assert rel_err(z.cpu(), zr.detach()) < 1e-5 and rel_l2(gx.cpu(), xr.grad) < 1e-5
            assert rel_l2(gg.cpu(), gr.grad) < 1e-5 and rel_l2(gb.cpu(), br.grad) < 1e-5
            if C > 1024:
                assert outs[0][4] == 1 and outs[1][4] == 0, "one warning, once"
            else:
                assert outs[0][4] == 0
                assert all(torch.equal(p, q_) for p, q_ in zip(outs[0][:4], outs[1][:4])), "C <= 1024: the deterministic kernel"
    finally:
        dose_prediction_amd.config.set_deterministic(False)


# ------------------------------------------------------------------------------------------------ nested backward passes and the pass arena
def test_nested_backward_does_not_reset_the_pass_arena():
    """A backward pass NESTED in another (reentrant activation checkpointing; here: torch.autograd.backward inside a backward node) has a
    graph-task id of its own.  The per-pass zero arena / zero-bias pool used to take a differing id for a dead pass and start over -- a
    fresh allocation and fill on entry to and exit from every segment.  Now the nested pass draws from the outer pass's arena
    (same buffer object before, inside and after), and a pass that really died (it raised) is still detected when the next network
    forward starts."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.float32)
    C = 64
    gam, bet = (1 + 0.1 * rnd((C,), 1)).to(dev).requires_grad_(True), rnd((C,), 2).to(dev).requires_grad_(True)
    seen = {}

    class Nested(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            seen["outer_before"] = ops._PASS_ARENA[dev]["buf"]
            with torch.enable_grad():
                u = rnd((4, C), 9).to(dev).requires_grad_(True)
                v = ops.layer_norm(u, gam.detach().requires_grad_(True), bet.detach().requires_grad_(True))
                torch.autograd.backward(v, torch.ones_like(v))            # nested pass: LayerNorm backward asks the arena for dgamma / dbeta
            seen["nested_after"] = ops._PASS_ARENA[dev]["buf"]
            seen["armed_after_nested"] = ops._PASS_ARENA[dev]["armed"]
            return g
    x = rnd((4, C), 3).to(dev).requires_grad_(True)
    y = ops.layer_norm(Nested.apply(ops.layer_norm(x, gam, bet)), gam, bet)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    assert seen["outer_before"] is not None and seen["nested_after"] is seen["outer_before"], "the nested pass must not replace the arena"
    assert seen["armed_after_nested"]
    assert not ops._PASS_ARENA[dev]["armed"], "disarmed by the outer pass's end-of-backward callback"
    ref = torch.nn.functional.layer_norm(torch.nn.functional.layer_norm(x.detach().double().cpu(), (C,), gam.detach().double().cpu(), bet.detach().double().cpu()),
                                         (C,), gam.detach().double().cpu(), bet.detach().double().cpu())
    assert rel_err(y.detach().cpu(), ref) < 1e-5
    assert torch.isfinite(gam.grad).all() and float(gam.grad.abs().sum()) > 0

    # a pass that dies leaves the arena armed; the next network-entry conversion outside any backward pass marks it dead
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")
    gam.grad = bet.grad = None
    y = ops.layer_norm(Boom.apply(ops.layer_norm(x, gam, bet)), gam, bet)
    with pytest.raises(RuntimeError, match="boom"):
        y.backward(torch.ones_like(y))
    assert ops._PASS_ARENA[dev]["armed"], "the dead pass never ran its end-of-backward callback"
    stale = ops._PASS_ARENA[dev]["buf"]
    from dose_prediction_amd.models.c3d import to_ndhwc
    to_ndhwc(rnd((1, 3, 4, 4, 8), 4).to(dev))                              # a network forward begins: epoch moves on
    gam.grad = bet.grad = None
    y = ops.layer_norm(ops.layer_norm(x, gam, bet), gam, bet)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    assert not ops._PASS_ARENA[dev]["armed"]
    xr, gr, br = x.detach().double().cpu().requires_grad_(True), gam.detach().double().cpu().requires_grad_(True), bet.detach().double().cpu().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(torch.nn.functional.layer_norm(xr, (C,), gr, br), (C,), gr, br)
    yr.backward(torch.ones_like(yr))
    assert rel_l2(gam.grad.cpu(), gr.grad) < 1e-5 and rel_l2(bet.grad.cpu(), br.grad) < 1e-5, "a stale arena must not leak into the next pass"
    del stale


# ------------------------------------------------------------------------------------------------ fp32x3: producer-side splits (round 6)
def test_x3_trilinear_writes_the_convolution_operand_directly():
    """c3d.UpConv (c3d.py:25-38) in the fp32x3 mode without gradients (the frozen C3D of DOSE-PYFER): dp_trilinear_up2_fwd with DP_X3 writes the
    bf16 [hi | lo] operand of the 3x3x3 convolution itself -- no fp32 up-sampled tensor, no split pass.  The halves are those of the same
    fp32 accumulators, so the block's output is BIT-identical to the unfused path (forced by enabling gradients), and both match the float64
    oracle at the x3 tolerance; dp_split_rows is not called on the fused path."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    from dose_prediction_amd.blocks import UpConv
    dev = _dev()
    dose_prediction_amd.set_compute_dtype("fp32x3")
    try:
        torch.manual_seed(3)
        up = UpConv(32, 16).to(dev)
        x = rnd((2, 32, 5, 9, 16), 7)                                     # NCDHW
        xh = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
        calls = []
        orig = ops.split_rows
        ops.split_rows = lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1]
        try:
            with torch.no_grad():
                y_fused = up(xh)
            n_fused = len(calls)
            y_plain = up(xh)                                              # gradients enabled + trainable weights: the fp32 tensor + split path
            n_plain = len(calls) - n_fused
        finally:
            ops.split_rows = orig
        assert n_fused == 0 and n_plain == 1, (n_fused, n_plain)
        assert torch.equal(y_fused, y_plain.detach())
        conv, norm = up.conv[0], up.conv[1]
        u = torch.nn.functional.interpolate(x.double(), scale_factor=2, mode="trilinear", align_corners=True)
        r = torch.nn.functional.conv3d(u, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1)
        r = torch.relu(torch.nn.functional.instance_norm(r, weight=norm.weight.detach().double().cpu(), bias=norm.bias.detach().double().cpu(), eps=norm.eps))
        got = y_fused.permute(0, 4, 1, 2, 3).double().cpu()
        assert rel_err(got, r) < 1e-4, rel_err(got, r)
    finally:
        dose_prediction_amd.set_compute_dtype(torch.float32)


def test_x3_branches_on_two_streams_share_one_input_split():
    """conv_3_1 (blocks_MDUNet.py:132-157) in the fp32x3 mode with the 3x3x3 branch on its own stream (config.set_branch_stream): the virtual
    concatenation both branches read is split ONCE -- the second consumer waits for the event of the split on the stream that made it --
    and the block's output and gradients equal the single-stream run's bit for bit (deterministic reductions)."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    from dose_prediction_amd.blocks import conv_3_1
    dev = _dev()
    dose_prediction_amd.set_compute_dtype("fp32x3")
    cfg = dose_prediction_amd.config
    prev = cfg.branch_stream()
    try:
        torch.manual_seed(5)
        blk = conv_3_1(32, 16, "mish").to(dev).train()
        a, b = rnd((1, 6, 20, 32, 16), 1).to(dev), rnd((1, 6, 20, 32, 16), 2).to(dev)
        r = rnd((1, 6, 20, 32, 16), 3).to(dev)
        res = {}
        with cfg.deterministic_as(True):
            for on in (True, False):
                cfg.set_branch_stream(on)
                calls = []
                orig = ops.split_rows
                ops.split_rows = lambda *x_, **k: (calls.append(x_[6] if len(x_) > 6 else None), orig(*x_, **k))[1]
                try:
                    blk.zero_grad(set_to_none=True)
                    a_, b_ = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
                    y = blk((a_, b_))
                    n_fwd = len(calls)
                    y.backward(r)
                    torch.cuda.synchronize()
                finally:
                    ops.split_rows = orig
                res[on] = (y.detach().clone(), a_.grad.clone(), b_.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}, n_fwd)
        assert res[True][4] == res[False][4] == 1, "one forward split of the shared input, whichever stream made it"
        for i in range(3):
            assert torch.equal(res[True][i], res[False][i]), i
        for k in res[True][3]:
            assert torch.equal(res[True][3][k], res[False][3][k]), k
    finally:
        cfg.set_branch_stream(prev)
        dose_prediction_amd.set_compute_dtype(torch.float32)
