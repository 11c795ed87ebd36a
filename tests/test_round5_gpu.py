"""Round 5: the deterministic-reduction switch (config.set_deterministic: bit-identical passes on demand) and
OAR-TRANSEG's own training loss on device (monai DiceCELoss(to_onehot_y=True, softmax=True), train_light_transeg.py:148)."""
import pytest
import torch

import oracle
from helpers import load_golden, pcg_state_dict, rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


@pytest.fixture
def det():
    import dose_prediction_amd
    dose_prediction_amd.config.set_deterministic(True)
    yield
    dose_prediction_amd.config.set_deterministic(False)
    dose_prediction_amd.set_compute_dtype(torch.float32)


# ------------------------------------------------------------------------------------------------ DiceCE
@pytest.mark.parametrize("label_dtype", [torch.float32, torch.int64, torch.int32, torch.uint8])
@pytest.mark.parametrize("shape", [(2, 8, 9, 10, 11), (1, 8, 32, 32, 40), (3, 5, 4, 4, 5), (1, 12, 6, 7, 8)])
def test_dice_ce_matches_the_oracle(shape, label_dtype):
    """ops.dice_ce / losses.DiceCELoss against oracle.dice_ce_loss (fp64): value, Dice and CE parts, gradient of the logits; labels in
    every storage type the kernel reads, with and without the channel axis; a class that never occurs (its Dice term is pure smoothing)."""
    from dose_prediction_amd import ops
    from dose_prediction_amd.losses import DiceCELoss
    dev = _dev()
    B, C = shape[:2]
    z = rnd(shape, 1, 2.0)
    lab = torch.randint(0, C - 1, (B, 1) + shape[2:], generator=torch.Generator().manual_seed(2))       # class C-1 never occurs
    zr = z.double().requires_grad_(True)
    ref = oracle.dice_ce_loss(zr, lab)
    (3.0 * ref).backward()
    zh = z.to(dev).requires_grad_(True)
    got = DiceCELoss(to_onehot_y=True, softmax=True)(zh, lab.to(dev, label_dtype))
    (3.0 * got).backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 2e-6 * max(1.0, abs(float(ref.detach())))
    assert rel_l2(zh.grad.cpu(), zr.grad) < 2e-6 and rel_err(zh.grad.cpu(), zr.grad) < 1e-5
    got2 = ops.dice_ce(z.to(dev), lab[:, 0].to(dev, label_dtype), lambda_dice=0.25, lambda_ce=2.0)
    ref2 = oracle.dice_ce_loss(z.double(), lab, lambda_dice=0.25, lambda_ce=2.0)
    assert abs(float(got2) - float(ref2)) < 2e-6 * max(1.0, abs(float(ref2)))


def test_dice_ce_rejects_what_it_does_not_implement():
    from dose_prediction_amd.losses import DiceCELoss
    for kw in (dict(), dict(to_onehot_y=True), dict(to_onehot_y=True, softmax=True, sigmoid=True), dict(to_onehot_y=True, softmax=True, jaccard=True),
               dict(to_onehot_y=True, softmax=True, include_background=False), dict(to_onehot_y=True, softmax=True, batch=True)):
        with pytest.raises(ValueError):
            DiceCELoss(**kw)


def test_dice_ce_full_size_128_identities():
    """2 x 8 x 128^3 logits (the C3 step's loss): uniform logits give loss = (1 - 1/8 - ...) closed form; gradient sums to zero over the
    classes of every voxel (softmax Jacobian) and is finite."""
    from dose_prediction_amd import ops
    dev = _dev()
    B, C, S = 2, 8, (128, 128, 128)
    lab = torch.randint(0, C, (B, 1) + S, generator=torch.Generator().manual_seed(3)).float().to(dev)
    z = torch.zeros((B, C) + S, device=dev, requires_grad=True)
    loss = ops.dice_ce(z, lab)
    V = S[0] * S[1] * S[2]
    cnt = torch.stack([(lab[b] == c).sum() for b in range(B) for c in range(C)]).double().cpu()
    dice = (1.0 - (2.0 * cnt / C + 1e-5) / (cnt + V / C + 1e-5)).mean()
    want = float(dice) + float(torch.log(torch.tensor(float(C))))
    assert abs(float(loss) - want) < 1e-5 * want
    loss.backward()
    assert bool(torch.isfinite(z.grad).all())
    assert float(z.grad.sum(1).abs().max()) < 1e-9


# ------------------------------------------------------------------------------------------------ deterministic mode: operators
def _twice(fn):
    a = fn()
    b = fn()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    return a


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, "fp32x3"])
@pytest.mark.parametrize("cfg", [
    # (N, Cin, Cout, D, H, W, k, stride): split-kd volumes, K-along-W / K-along-H / generic weight gradients, pointwise, stride 2
    (1, 64, 64, 6, 6, 16, 3, 1), (1, 128, 80, 4, 4, 16, 7, 1), (2, 16, 16, 3, 33, 40, 7, 1), (1, 32, 16, 4, 40, 36, 7, 1),
    (1, 16, 16, 5, 9, 130, 3, 1), (1, 64, 40, 2, 33, 32, 7, 1), (1, 16, 32, 8, 8, 8, 3, 2), (1, 32, 16, 32, 32, 40, 1, 1),
    (1, 24, 40, 3, 20, 24, 3, 1)])
def test_deterministic_conv_matches_the_oracle_and_itself(cfg, dtype, det):
    """Every convolution path under config.set_deterministic(True): results still match the fp64 oracle at the mode's tolerance, and two
    runs are BIT-identical (forward, data gradient, weight gradient, bias gradient)."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(dtype)
    st = torch.float32 if dtype == "fp32x3" else dtype
    N, Cin, Cout, D, H, W, k, s = cfg
    q = (lambda t: t.to(st).float())
    x, w, b = q(rnd((N, Cin, D, H, W), 1)), q(rnd((Cout, Cin, k, k, k), 2, (Cin * k ** 3) ** -0.5)), rnd((Cout,), 3, 0.1)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = oracle.conv3d(xr, wr, br, s, k // 2, 1)
    r = q(rnd(yr.shape, 4))
    (yr * r.double()).sum().backward()

    def run():
        xh = ndhwc(x).to(dev, st).requires_grad_(True)
        wh, bh = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        yh = ops.conv3d(xh, wh, bh, s, k // 2, 1)
        yh.backward(ndhwc(r).to(dev, st))
        torch.cuda.synchronize()
        return yh.detach().clone(), xh.grad.clone(), wh.grad.clone(), bh.grad.clone()
    y, gx, gw, gb = _twice(run)
    tol = {torch.float32: 2e-5, torch.bfloat16: 6e-3, "fp32x3": 8e-3}[dtype]       # (fp32x3: one-product gradients by default)
    assert rel_l2(ncdhw(y).float().cpu(), yr) < (3e-5 if dtype == "fp32x3" else tol)
    assert rel_l2(ncdhw(gx).float().cpu(), xr.grad) < tol
    assert rel_l2(gw.cpu(), wr.grad) < tol and rel_l2(gb.cpu(), br.grad) < tol


def test_deterministic_layernorm_linear_splitk_trilinear_instance_affine(det):
    """The remaining order-dependent reductions under the switch: LayerNorm's dgamma / dbeta (plain and fused add + LayerNorm), the
    split-K patch-embedding GEMM, the trilinear up-sampling's backward, InstanceNorm(affine) dgamma / dbeta over N = 3 samples: equal to
    the oracle and bit-identical between runs."""
    import dose_prediction_amd
    from dose_prediction_amd import ops
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.float32)
    x, g, b, go = rnd((2, 300, 96), 1), 1 + 0.2 * rnd((96,), 2), 0.1 * rnd((96,), 3), rnd((2, 300, 96), 4)

    def ln():
        xh, gh, bh = (t.to(dev).requires_grad_(True) for t in (x, g, b))
        ops.layer_norm(xh, gh, bh).backward(go.to(dev))
        s, z = ops.add_layer_norm(xh, 0.5 * xh, gh, bh)
        (s * go.to(dev)).sum().backward(retain_graph=True)
        z.backward(go.to(dev))
        torch.cuda.synchronize()
        return xh.grad.clone(), gh.grad.clone(), bh.grad.clone()
    gx, gg, gb = _twice(ln)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, g, b))
    torch.nn.functional.layer_norm(xr, (96,), gr, br).backward(go.double())
    s = 1.5 * xr
    z = torch.nn.functional.layer_norm(s, (96,), gr, br)
    ((s * go.double()).sum() + (z * go.double()).sum()).backward()
    assert rel_l2(gx.cpu(), xr.grad) < 2e-5 and rel_l2(gg.cpu(), gr.grad) < 2e-5 and rel_l2(gb.cpu(), br.grad) < 2e-5

    t, w, bb = rnd((1, 64, 8192), 5), rnd((96, 8192), 6, 8192 ** -0.5), 0.1 * rnd((96,), 7)

    def lin():
        th, wh = t.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = ops.linear(th, wh, bb.to(dev), splitk=8)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        return y.detach().clone(), th.grad.clone(), wh.grad.clone()
    y, _, _ = _twice(lin)
    assert rel_l2(y.cpu(), torch.nn.functional.linear(t.double(), w.double(), bb.double())) < 2e-5

    u, gu = rnd((2, 6, 5, 7, 9), 8), rnd((2, 12, 10, 14, 9), 9)

    def tri():
        uh = u.to(dev).requires_grad_(True)
        ops.trilinear_up2(uh).backward(gu.to(dev))
        torch.cuda.synchronize()
        return (uh.grad.clone(),)
    (gu_h,) = _twice(tri)
    ur = ncdhw(u).double().requires_grad_(True)
    oracle.trilinear_up2(ur).backward(ncdhw(gu).double())
    assert rel_l2(ncdhw(gu_h).cpu(), ur.grad) < 2e-5

    v, ga, be, gv = rnd((3, 4, 6, 8, 16), 10), 1 + 0.2 * rnd((16,), 11), 0.1 * rnd((16,), 12), rnd((3, 4, 6, 8, 16), 13)

    def inorm():
        vh, gh, bh = (t_.to(dev).requires_grad_(True) for t_ in (v, ga, be))
        ops.norm_act(vh, "instance", gh, bh, act="relu").backward(gv.to(dev))
        torch.cuda.synchronize()
        return vh.grad.clone(), gh.grad.clone(), bh.grad.clone()
    gvh, ggh, gbh = _twice(inorm)
    vr, gar, ber = ncdhw(v).double().requires_grad_(True), ga.double().requires_grad_(True), be.double().requires_grad_(True)
    torch.relu(oracle.instance_norm(vr, gar, ber)).backward(ncdhw(gv).double())
    assert rel_l2(ncdhw(gvh).cpu(), vr.grad) < 2e-5 and rel_l2(ggh.cpu(), gar.grad) < 2e-5 and rel_l2(gbh.cpu(), ber.grad) < 2e-5


# ------------------------------------------------------------------------------------------------ deterministic mode: networks
def _subset_net(dev, g):
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                          num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
    net.load_state_dict(pcg_state_dict(g["keys"], g["shapes"], g["seed"]))
    return net.to(dev).train()


@pytest.mark.parametrize("mode", ["fp32x3", torch.float32, torch.bfloat16])
def test_backward_passes_of_the_g7_subset_network_are_bit_identical_under_the_switch(mode, det):
    """the G7 subset network (the one whose backward pass landed on a second result about once in a hundred runs,
    tools/x3_event_bisect.py) run N times forward + backward with config.set_deterministic(True): every output and every one of its
    gradients is bitwise equal to the first run's.  N = 1000 in fp32x3 (the mode the retry loop existed for), 200 in the others."""
    import dose_prediction_amd
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(mode)
    g = load_golden("g7_subset_multi")
    net = _subset_net(dev, g)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = g["x"].to(dev)
    first = None
    n = 1000 if mode == "fp32x3" else 200
    for it in range(n):
        net.load_state_dict(sd0)                    # (BatchNorm running buffers back to the start)
        net.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        outs = net(xi)
        sum((o * o).mean() for o in outs).backward()
        cur = [o.detach() for o in outs] + [xi.grad] + [p.grad for p in net.parameters() if p.grad is not None]
        if first is None:
            first = [t.clone() for t in cur]
        else:
            bad = [i for i, (a, b) in enumerate(zip(first, cur)) if not torch.equal(a, b)]
            assert not bad, f"run {it}: tensors {bad[:8]} differ from run 0 with the deterministic switch on"


def test_pyfer_training_step_is_bit_identical_under_the_switch_64(det):
    """DOSE-PYFER at production width on 2 x 64^3 in bf16 (every tuned kernel family: cc16 / tiled convolutions, K-along-H and marching
    weight gradients, row kernels, fused attention, grouped transformer weight gradients, side streams): two training steps from the
    same state give bit-identical losses, gradients and updated parameters."""
    import dose_prediction_amd
    from dose_prediction_amd import synth, losses
    from dose_prediction_amd.models.dose_pyfer import Model
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    dose_prediction_amd.set_compute_dtype(torch.bfloat16)
    S = (64, 64, 64)
    torch.manual_seed(4321)
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 16, 32, 64, 128, 256], feature_size=16, img_size=S, num_layers=8, num_heads=6, act="mish")
    for n_, p in net.named_parameters():
        if "net_A" in n_ or "conv_out_A" in n_:
            p.requires_grad = False
    net = net.to(dev).train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x, gt = synth.dose_input(2, S).to(dev), synth.dose_target(2, S).to(dev)

    def run():
        net.load_state_dict(sd0)
        params = [p for p in net.parameters() if p.requires_grad]
        opt = FusedAdam(params, lr=1e-4, weight_decay=3e-5, amsgrad=True)
        opt.zero_grad(set_to_none=True)
        loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
        loss.backward()
        grads = [p.grad.clone() for p in params if p.grad is not None]
        opt.step()
        torch.cuda.synchronize()
        return [loss.detach().clone()] + grads + [p.detach().clone() for p in params]
    a, b = run(), run()
    bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
    assert not bad, f"{len(bad)} of {len(a)} tensors differ between two deterministic steps (first: {bad[:8]})"


# ------------------------------------------------------------------------------------------------ finalize folded into the partial pass
@pytest.mark.parametrize("shape,batch_mode,dtype", [
    ((2, 64, 64, 64, 16), 0, torch.bfloat16), ((2, 64, 64, 64, 16), 1, torch.float32), ((4, 24, 24, 24, 64), 0, torch.float32),
    ((1, 16, 16, 16, 256), 1, torch.bfloat16), ((3, 9, 10, 11, 24), 0, torch.float16), ((2, 128, 128, 128, 16), 0, torch.bfloat16)])
def test_folded_finalize_is_bit_identical_to_the_two_launch_form(shape, batch_mode, dtype):
    """dp_stats_partial_finalize / dp_norm_act_bwd_partial_finalize (the last block of every statistics group runs the finalize: one
    launch) against dp_stats_partial + dp_stats_finalize / dp_norm_act_bwd_partial + dp_norm_bwd_finalize, 100 repetitions each on two
    streams at once (a stale read of another block's row, or two launches sharing a counter, would show as a differing bit)."""
    from dose_prediction_amd import _lib, ops
    L = _lib.lib()
    if not L.dp_ticket_enabled():
        pytest.skip("DP_NO_TICKET is set")
    dev = torch.device("cuda:0")
    N, C = shape[0], shape[-1]
    V = shape[1] * shape[2] * shape[3]
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(dev).to(dtype)
    gy = torch.randn(shape, generator=g).to(dev).to(dtype)
    gamma, beta = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)
    dtc, nblk, groups = ops._dt(x), L.dp_stats_nblk(V), (1 if batch_mode else N)
    P = lambda t: 0 if t is None else t.data_ptr()
    need_gb = bool(batch_mode) or N == 1      # (affine instance normalisation over several samples keeps the two-launch form)

    def forward(folded, stream):
        part = torch.empty((N, nblk, 2, C), dtype=torch.float32, device=dev)
        mean, rstd = torch.empty((groups, C), device=dev), torch.empty((groups, C), device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        fin = (batch_mode, 1e-5, P(mean), P(rstd), P(rm) if batch_mode else 0, P(rv) if batch_mode else 0, 0.1)
        if folded:
            assert _lib.call("dp_stats_partial_finalize", P(x), C, N, V, C, P(part), *fin, dtc, stream) == 0
        else:
            _lib.call("dp_stats_partial", P(x), C, N, V, C, P(part), dtc, stream)
            _lib.call("dp_stats_finalize", P(part), N, nblk, C, V, *fin, stream)
        return mean, rstd, rm, rv

    def backward(folded, stream, mean, rstd):
        part = torch.empty((N, nblk, 2, C), dtype=torch.float32, device=dev)
        s1, s2 = torch.empty((groups, C), device=dev), torch.empty((groups, C), device=dev)
        dg, db = (torch.empty(C, device=dev), torch.empty(C, device=dev)) if need_gb else (None, None)
        src = (P(x), C, P(gy), C, P(mean), P(rstd), 0 if batch_mode else C, P(gamma), P(beta), 0, 0, 1, N, V, C, P(part))
        fin = (batch_mode, P(s1), P(s2), P(dg), P(db))
        if folded:
            assert _lib.call("dp_norm_act_bwd_partial_finalize", *src, *fin, dtc, stream) == 0
        else:
            _lib.call("dp_norm_act_bwd_partial", *src, dtc, stream)
            _lib.call("dp_norm_bwd_finalize", P(part), N, nblk, C, *fin, stream)
        return [t for t in (s1, s2, dg, db) if t is not None]

    main = torch.cuda.current_stream()
    ref_f = forward(False, main.cuda_stream)
    ref_b = backward(False, main.cuda_stream, ref_f[0], ref_f[1])
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in ref_f + tuple(ref_b))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in streams:
        s.wait_stream(main)
    reps = 100 if V * C <= (1 << 24) else 20
    got = []
    for i in range(reps):
        s = streams[i & 1]
        with torch.cuda.stream(s):
            f = forward(True, s.cuda_stream)
            got.append((f, backward(True, s.cuda_stream, ref_f[0], ref_f[1])))
    torch.cuda.synchronize()
    for f, b in got:
        for a, r in zip(f, ref_f):
            assert torch.equal(a, r)
        for a, r in zip(b, ref_b):
            assert torch.equal(a, r)
    # an affine instance normalisation over several samples is declined (3), not served wrongly
    if not batch_mode and N > 1:
        part = torch.empty((N, nblk, 2, C), dtype=torch.float32, device=dev)
        t = [torch.empty((N, C), device=dev) for _ in range(2)] + [torch.empty(C, device=dev) for _ in range(2)]
        src = (P(x), C, P(gy), C, P(ref_f[0]), P(ref_f[1]), C, P(gamma), P(beta), 0, 0, 1, N, V, C, P(part))
        assert _lib.call("dp_norm_act_bwd_partial_finalize", *src, 0, *[P(v) for v in t], dtc, main.cuda_stream) == 3


def test_marching_3x3x3_experiment_kernel_passes_the_conv_parity_cases():
    """k_conv_cc16m (csrc/conv_cc16.hip: input-stationary 3^3 marching along depth; measured slower, hence opt-in through DP_CC16M=1,
    which the library reads once): the 3^3 long-row cases of test_conv3d, the epilogue-statistics and the virtual-concat tests run
    in a child process with the switch on, in all storage types and in the fp32x3 convolution test."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DP_CC16M="1", DOSE_TEST_CC16_3_ONLY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_ops_gpu.py"),
                        "-k", "test_conv3d or conv_epilogue_statistics or conv3d_virtual_concat"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
