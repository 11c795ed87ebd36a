"""GPU parity of every HIP op (forward + backward, through the C ABI) against the CPU oracle leaf ops.

fp32 mode: exact-fp32 MFMA -> tolerance 2e-5 (relative L2) / 1e-4 (max-abs relative to max|ref|).
bf16 mode: inputs are rounded to bf16 FIRST and the oracle is evaluated (in fp64) on the rounded inputs, so the only
differences are fp32 accumulation order and the final rounding of outputs to bf16 (2^-9): tolerance 6e-3 rel-L2.
fp16 mode: the same protocol with fp16 rounding (2^-12): tolerance 1e-3 rel-L2.
"""
import copy
import math
import os

import pytest
import torch

import oracle
from helpers import rel_l2, rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-5, 1e-4), torch.bfloat16: (6e-3, 2e-2), torch.float16: (1e-3, 4e-3)}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def q(t, dtype):
    """Quantise a CPU fp32 tensor to the compute dtype's grid (identity for fp32)."""
    return t.to(dtype).float()


def ndhwc(t):   # NCDHW -> NDHWC
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


def check(name, got, ref, dtype, scale=1.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name
    tl2, tmax = TOL[dtype]
    e2, em = rel_l2(got, ref), rel_err(got, ref)
    assert e2 < tl2 * scale and em < tmax * scale, f"{name}: rel_l2={e2:.3e} rel_max={em:.3e} ({dtype})"


DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # (N, Cin, Cout, D, H, W, k, stride, pad, dil, bias)
    (2, 16, 16, 6, 10, 20, 3, 1, 1, 1, True),
    (1, 9, 16, 8, 8, 8, 3, 1, 1, 1, True),
    (1, 25, 16, 5, 7, 9, 3, 1, 1, 1, False),
    (2, 32, 16, 9, 8, 17, 7, 1, 3, 1, True),
    (1, 16, 32, 8, 8, 8, 3, 2, 1, 1, True),
    (1, 8, 4, 12, 12, 12, 3, 1, 2, 2, True),
    (1, 8, 4, 12, 12, 12, 3, 1, 3, 3, True),
    (1, 32, 16, 6, 6, 6, 1, 1, 0, 1, True),
    (2, 16, 1, 4, 6, 8, 1, 1, 0, 1, True),
    (1, 64, 64, 6, 6, 6, 3, 1, 1, 1, False),
    (1, 128, 80, 4, 4, 4, 7, 1, 3, 1, True),
    (1, 4, 4, 5, 5, 5, 7, 1, 3, 1, True),
    # LDS-tiled kernel: tap-paired (Cout<=16), Cout<=32, Cout>32 (two N tiles / grid.y), ragged W / H / channel counts
    (1, 32, 32, 3, 20, 70, 7, 1, 3, 1, True),
    (1, 16, 64, 2, 18, 33, 3, 1, 1, 1, True),
    (1, 64, 128, 2, 9, 16, 7, 1, 3, 1, False),
    (2, 32, 16, 3, 9, 130, 7, 1, 3, 1, True),
    (1, 24, 40, 2, 7, 40, 3, 1, 1, 1, True),
    (1, 16, 16, 9, 17, 64, 7, 1, 3, 1, True),
    (1, 9, 16, 4, 11, 36, 3, 1, 1, 1, True),
    (1, 128, 64, 2, 6, 32, 3, 1, 1, 1, True),
    (2, 1, 16, 3, 9, 40, 3, 1, 1, 1, False),      # single input channel (OAR-TRANSEG encoder1: CT -> 16)
    (4, 128, 128, 12, 12, 12, 7, 1, 3, 1, True),  # 12^3 level of the 96^3 sliding-window crop (W < 16 tiles), batch of 4 windows
    (2, 64, 32, 12, 12, 12, 3, 1, 1, 1, False),
    (1, 16, 16, 10, 9, 12, 3, 1, 1, 1, True),     # W < 16 with the tap-paired (Cout <= 16) configuration
    (1, 3, 16, 2, 9, 32, 7, 1, 3, 1, True),
    (2, 64, 32, 8, 8, 10, 3, 2, 1, 1, True),      # few voxels, many channels, stride 2: im2col + GEMM forward
    (1, 40, 24, 6, 5, 7, 3, 1, 2, 2, False),      # same path with dilation and ragged channels
    (1, 128, 1, 8, 8, 8, 1, 1, 0, 1, True),       # pointwise head over few voxels: TN-GEMM weight gradient
    (1, 16, 1, 32, 32, 40, 1, 1, 0, 1, True),     # deep-supervision head at >= 32768 voxels: row-stream forward + wgrad
    (2, 32, 3, 16, 32, 40, 1, 1, 0, 1, True),
    (1, 64, 2, 16, 32, 64, 1, 1, 0, 1, False),
    (1, 32, 16, 32, 32, 40, 1, 1, 0, 1, True),    # conv_3_1 mixer (tiled k = 1 weight gradient)
    (1, 25, 16, 32, 32, 40, 1, 1, 0, 1, True),
    # the matrix-core row stream (k_pointwise_mfma: > 16 input channels, 16-bit storage): one / two 32-channel chunks x one / two
    # 16-row output tiles, a ragged number of output channels, a voxel count that is not a multiple of the 16-voxel tile
    (1, 64, 32, 32, 32, 40, 1, 1, 0, 1, True),
    (1, 64, 16, 32, 32, 40, 1, 1, 0, 1, False),
    (1, 32, 32, 33, 31, 35, 1, 1, 0, 1, True),
    (2, 32, 24, 16, 31, 35, 1, 1, 0, 1, True),
    (1, 64, 12, 32, 32, 40, 1, 1, 0, 1, True),
    (1, 16, 32, 32, 32, 40, 1, 1, 0, 1, True),    # (16 inputs: half a chunk, the input-gradient shape of the 32 -> 16 mixer)
    # k_wgrad_rows (streaming matrix-core weight gradient of the row GEMMs): 8 / 24 / 40 / 64 inputs, 8 .. 128 outputs, a row count
    # that is not a multiple of the 32-row slab
    (1, 8, 8, 32, 32, 40, 1, 1, 0, 1, True),
    (1, 24, 40, 33, 31, 35, 1, 1, 0, 1, False),
    (2, 40, 128, 16, 32, 40, 1, 1, 0, 1, True),
    (1, 64, 64, 32, 32, 40, 1, 1, 0, 1, True),
    # 7^3 weight gradient with K along H (conv_wgrad_hk.hip: <= 16 output channels, planes >= 32 x 32): exact / ragged tiles, two
    # input-channel tiles, a partial second channel tile, fewer than 16 output channels
    (1, 16, 16, 2, 32, 32, 7, 1, 3, 1, True),
    (1, 16, 16, 4, 40, 36, 7, 1, 3, 1, False),
    (2, 32, 16, 3, 33, 70, 7, 1, 3, 1, True),
    (1, 24, 8, 5, 64, 32, 7, 1, 3, 1, True),
    (1, 32, 32, 2, 32, 40, 7, 1, 3, 1, True),     # ... wider layers: several 16-channel output tiles on the grid
    (1, 64, 40, 2, 33, 32, 7, 1, 3, 1, False),
    # 3^3 weight gradient marching along depth (k_wgrad_hk3: all 27 taps per block, x planes in an LDS ring): one plane, a few
    # planes, depth segments with ragged ends, ragged tiles, several channel tiles
    (1, 16, 16, 1, 32, 32, 3, 1, 1, 1, True),
    (1, 16, 16, 5, 32, 32, 3, 1, 1, 1, False),
    (2, 32, 16, 7, 40, 36, 3, 1, 1, 1, True),
    (1, 24, 40, 3, 33, 64, 3, 1, 1, 1, True),
    (1, 16, 8, 19, 64, 64, 3, 1, 1, 1, False),
    # 3^3 forward / data gradient of the 16-channel layers on long rows (k_conv_cc16<3>: W >= 96): ragged H / W tiles, a partial
    # input chunk, fewer than 16 output channels, many depth slices
    (1, 16, 16, 5, 9, 130, 3, 1, 1, 1, True),
    (2, 9, 16, 4, 20, 100, 3, 1, 1, 1, True),
    (1, 16, 8, 37, 8, 128, 3, 1, 1, 1, False),
    (1, 16, 16, 4, 40, 96, 3, 1, 1, 1, True),
    # one-column tiles (16 < W <= 32) with two or four 32-channel output tiles and enough planes to skip the kd split: 8-row blocks,
    # two row groups x two channel-tile groups per block (tiled_geometry, round 3); ragged H / W, a second block row of channel tiles
    (2, 64, 64, 32, 30, 24, 3, 1, 1, 1, True),
    (2, 64, 128, 26, 32, 20, 3, 1, 1, 1, False),
    (4, 32, 64, 26, 12, 32, 7, 1, 3, 1, True),
    # row lengths other than 128 (k_conv_cc16w: 96-position tiles, waves = column half x row half, both depth slices per wave): the
    # 96^3 crop of the segmentation network and the 192-wide rows of BASELINE configs[4]; odd depth (a block with one live slice),
    # ragged rows, a ragged second tile (W = 160), partial chunks, fewer than 16 output channels
    (1, 16, 16, 3, 17, 96, 7, 1, 3, 1, True),
    (1, 32, 16, 2, 9, 192, 7, 1, 3, 1, True),
    (2, 9, 16, 3, 12, 160, 3, 1, 1, 1, True),
    (1, 16, 8, 5, 8, 96, 7, 1, 3, 1, False),
    (2, 16, 16, 1, 8, 192, 3, 1, 1, 1, True),
    (1, 25, 16, 4, 11, 96, 3, 1, 1, 1, False),
    (1, 16, 16, 2, 9, 160, 7, 1, 3, 1, True),     # 96 + 64: a ragged second 96-position tile
    (1, 32, 16, 3, 8, 130, 7, 1, 3, 1, False),    # 96 + 34 (two tiles of 96 beat two of 128)
    (1, 16, 12, 1, 12, 288, 7, 1, 3, 1, True),    # three whole tiles, one depth slice (every block has one dead slice), 12 output channels
    (2, 64, 32, 3, 24, 24, 7, 1, 3, 1, True),     # K-along-H weight gradient on a 24 x 24 plane (one zero-padded 32 x 32 tile): the 24^3 level of the 96^3 crop
    (1, 16, 16, 2, 24, 30, 7, 1, 3, 1, False),
    (1, 40, 24, 4, 27, 25, 7, 1, 3, 1, True),
])
def test_conv3d(cfg, dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    N, Cin, Cout, D, H, W, k, s, p, dl, has_b = cfg
    if os.environ.get("DOSE_TEST_CC16_3_ONLY") and not (k == 3 and s == 1 and Cout <= 16 and W >= 96):
        pytest.skip("DOSE_TEST_CC16_3_ONLY: only the 3^3 cases of the 16-channel long-row kernels")
    x = q(rnd((N, Cin, D, H, W), 1), dtype)
    w = q(rnd((Cout, Cin, k, k, k), 2, (Cin * k ** 3) ** -0.5), dtype)
    b = rnd((Cout,), 3, 0.1) if has_b else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    yr = oracle.conv3d(xr, wr, br, s, p, dl)
    r = q(rnd(yr.shape, 4), dtype)
    (yr * r.double()).sum().backward()
    xh = ndhwc(x).to(dev, dtype).requires_grad_(True)
    wh = w.to(dev).requires_grad_(True)
    bh = b.to(dev).requires_grad_(True) if has_b else None
    yh = ops.conv3d(xh, wh, bh, s, p, dl)
    yh.backward(ndhwc(r).to(dev, dtype))
    check("y", ncdhw(yh), yr, dtype)
    check("gx", ncdhw(xh.grad), xr.grad, dtype)
    check("gw", wh.grad, wr.grad, dtype)
    if has_b:
        check("gb", bh.grad, br.grad, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,ldx", [(25, 16, 64), (40, 24, 64), (33, 32, 64), (64, 32, 64), (16, 32, 16), (25, 16, 32), (40, 32, 40), (20, 16, 24),
                                          (132, 16, 256), (164, 32, 192), (200, 24, 224), (160, 16, 256)])   # (eight-chunk instance with 5-7 real chunks: ADVICE r4)
def test_pointwise_rows_ragged_channels_ignore_the_row_padding(cin, cout, ldx, dtype):
    """dp_pointwise_rows on rows wider than Cin (pitch a multiple of 32): the matrix-core path reads whole 32-channel chunks and must
    mask what lies beyond Cin -- the padding here holds NaN / Inf.  Output rows are slices of a wider buffer whose other columns stay."""
    from dose_prediction_amd import ops
    dev = _dev()
    rows, ldy = 32768 + 7, 40
    x = q(rnd((rows, cin), 1), dtype)
    w = q(rnd((cout, cin), 2, cin ** -0.5), dtype)
    b = rnd((cout,), 3, 0.1)
    xb = torch.full((rows, ldx), float("nan"), dtype=dtype, device=dev)
    xb[:, cin:cin + 2] = float("inf")
    xb[:, :cin] = x.to(dev, dtype)
    yb = torch.full((rows, ldy), 7.0, dtype=dtype, device=dev)
    ops.gemm_nt(xb, w.to(dev, dtype).contiguous(), yb, bias=b.to(dev), M=rows, N=cout, K=cin, lda=ldx, ldb=cin, ldc=ldy)
    yr = x.double() @ w.double().t() + b.double()
    check("y", yb[:, :cout], yr, dtype)
    assert bool((yb[:, cout:] == 7.0).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3d_channel_slice_and_padding(dtype):
    """Input is a channel slice of a wider buffer (pitch > C) with extra zero-padded channels."""
    from dose_prediction_amd import ops
    dev = _dev()
    x = q(rnd((1, 9, 6, 6, 6), 1), dtype)
    w = q(rnd((8, 9, 3, 3, 3), 2, 0.1), dtype)
    yr = oracle.conv3d(x.double(), w.double(), None, 1, 1, 1)
    buf = torch.zeros((1, 6, 6, 6, 40), dtype=dtype, device=dev)
    buf[..., 8:17] = ndhwc(x).to(dev, dtype)
    yh = ops.conv3d(buf[..., 8:24], w.to(dev), None, 1, 1, 1)     # 16 channels visible, 9 used
    check("y", ncdhw(yh), yr, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 48, 8, 2, 1, 1), (1, 32, 16, 4, 4, 4), (1, 768, 128, 2, 2, 2), (1, 6, 3, 3, 3, 3),
                                 # >= 32768 voxels: the one-launch forward (dp_tconv2x_fwd, pixel shuffle in the store) and the matrix-core row
                                 # kernel for the data gradient (K = 8 Cout up to 128): one / two / four 32-channel chunks, ragged Cin, a non-cubic
                                 # volume whose extents are not powers of two, Cout = 8 (8-byte stores), and a shape outside (Cout = 12)
                                 (2, 32, 16, 32, 32, 32), (1, 64, 16, 32, 32, 40), (1, 128, 16, 32, 32, 32), (2, 40, 8, 24, 28, 36),
                                 (1, 32, 12, 32, 32, 40), (1, 96, 32, 32, 32, 32), (1, 256, 24, 32, 32, 32)])
def test_conv_transpose(cfg, dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    N, Cin, Cout, D, H, W = cfg
    x = q(rnd((N, Cin, D, H, W), 1), dtype)
    w = q(rnd((Cin, Cout, 2, 2, 2), 2, Cin ** -0.5), dtype)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = oracle.conv_transpose3d_k2s2(xr, wr)
    r = q(rnd(yr.shape, 3), dtype)
    (yr * r.double()).sum().backward()
    xh, wh = ndhwc(x).to(dev, dtype).requires_grad_(True), w.to(dev).requires_grad_(True)
    yh = ops.conv_transpose2x(xh, wh)
    yh.backward(ndhwc(r).to(dev, dtype))
    check("y", ncdhw(yh), yr, dtype)
    check("gx", ncdhw(xh.grad), xr.grad, dtype)
    check("gw", wh.grad, wr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind,affine,act,res,training", [
    ("instance", True, "relu", False, True), ("instance", False, "mish", False, True), ("instance", False, "lrelu", True, True),
    ("instance", False, None, False, True), ("batch", True, "relu", False, True), ("batch", True, "relu", False, False)])
def test_norm_act(kind, affine, act, res, training, dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    N, C, D, H, W = 2, 16, 7, 9, 40        # V = 2520 > one stats block
    x = q(rnd((N, C, D, H, W), 1) * 1.5 + 0.3, dtype)
    gam = (1 + 0.2 * rnd((C,), 2)) if affine else None
    bet = 0.1 * rnd((C,), 3) if affine else None
    rs = q(rnd((N, C, D, H, W), 4), dtype) if res else None
    rm, rv = 0.1 * rnd((C,), 5), rnd((C,), 6).abs() + 0.5
    xr = x.double().requires_grad_(True)
    gr = gam.double().requires_grad_(True) if affine else None
    br = bet.double().requires_grad_(True) if affine else None
    rr = rs.double().requires_grad_(True) if res else None
    if kind == "instance":
        yr = oracle.instance_norm(xr, gr, br)
        new_rm = new_rv = None
    else:
        yr, new_rm, new_rv = oracle.batch_norm(xr, gr, br, rm.double(), rv.double(), training)
    if res:
        yr = yr + rr
    yr = oracle.activation(yr, act)
    r = q(rnd(yr.shape, 7), dtype)
    (yr * r.double()).sum().backward()
    xh = ndhwc(x).to(dev, dtype).requires_grad_(True)
    gh = gam.to(dev).requires_grad_(True) if affine else None
    bh = bet.to(dev).requires_grad_(True) if affine else None
    rh = ndhwc(rs).to(dev, dtype).requires_grad_(True) if res else None
    rmh, rvh = rm.to(dev), rv.to(dev)
    yh = ops.norm_act(xh, kind, gh, bh, rmh, rvh, training, rh, act)
    yh.backward(ndhwc(r).to(dev, dtype))
    check("y", ncdhw(yh), yr, dtype)
    check("gx", ncdhw(xh.grad), xr.grad, dtype, scale=2.0)
    if affine:
        check("ggamma", gh.grad, gr.grad, dtype)
        check("gbeta", bh.grad, br.grad, dtype)
    if res:
        check("gres", ncdhw(rh.grad), rr.grad, dtype)
    if kind == "batch" and training:
        check("running_mean", rmh, new_rm, torch.float32)
        check("running_var", rvh, new_rv, torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(300, 768, 96, True), (64, 48, 144, False), (17, 96, 48, True), (1024, 1000, 64, True)])
def test_linear(cfg, dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    rows, K, Nout, has_b = cfg
    x = q(rnd((2, rows // 2 if rows % 2 == 0 else rows, K), 1), dtype)
    w = q(rnd((Nout, K), 2, K ** -0.5), dtype)
    b = 0.1 * rnd((Nout,), 3) if has_b else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    yr = oracle.linear(xr, wr, br)
    r = q(rnd(yr.shape, 4), dtype)
    (yr * r.double()).sum().backward()
    for splitk in (1, 3):
        xh, wh = x.to(dev, dtype).requires_grad_(True), w.to(dev).requires_grad_(True)
        bh = b.to(dev).requires_grad_(True) if has_b else None
        yh = ops.linear(xh, wh, bh, splitk)
        yh.backward(r.to(dev, dtype))
        check("y", yh, yr, dtype)
        check("gx", xh.grad, xr.grad, dtype)
        check("gw", wh.grad, wr.grad, dtype)
        if has_b:
            check("gb", bh.grad, br.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_gelu_add(dtype, C=768):
    from dose_prediction_amd import ops
    dev = _dev()
    x = q(rnd((2, 37, C), 1) * 2 + 0.5, dtype)
    g, b = 1 + 0.2 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = oracle.gelu(oracle.layer_norm(xr, gr, br)) + xr
    r = q(rnd(yr.shape, 4), dtype)
    (yr * r.double()).sum().backward()
    xh, gh, bh = x.to(dev, dtype).requires_grad_(True), g.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    yh = ops.add(ops.gelu(ops.layer_norm(xh, gh, bh)), xh)
    yh.backward(r.to(dev, dtype))
    check("y", yh, yr, dtype, scale=2.0)
    check("gx", xh.grad, xr.grad, dtype, scale=2.0)
    check("ggamma", gh.grad, gr.grad, dtype, scale=2.0)
    check("gbeta", bh.grad, br.grad, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [768, 48, 1000])
def test_add_layernorm_fused(dtype, C):
    """Fused residual add + LayerNorm (dp_add_layernorm_fwd/_bwd): the sum and the normalised output are BIT-identical to the separate
    add and LayerNorm kernels (the sum is rounded to the storage type before the statistics); gradients against the fp64 oracle, with
    both outputs used (the residual-path gradient is added inside the LayerNorm backward)."""
    from dose_prediction_amd import ops
    dev = _dev()
    a, b = q(rnd((2, 37, C), 1) * 2 + 0.5, dtype), q(rnd((2, 37, C), 5), dtype)
    g, be = 1 + 0.2 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    ar, br_, gr, ber = (t.double().requires_grad_(True) for t in (a, b, g, be))
    sr = ar + br_
    zr = oracle.layer_norm(sr, gr, ber)
    r1, r2 = q(rnd(zr.shape, 4), dtype), q(rnd(zr.shape, 6), dtype)
    ((zr * r1.double()).sum() + (sr * r2.double()).sum()).backward()
    ah, bh = a.to(dev, dtype).requires_grad_(True), b.to(dev, dtype).requires_grad_(True)
    gh, beh = g.to(dev).requires_grad_(True), be.to(dev).requires_grad_(True)
    sh, zh = ops.add_layer_norm(ah, bh, gh, beh)
    with torch.no_grad():
        s0 = ops.add(ah, bh)
        z0 = ops.layer_norm(s0, gh, beh)
    assert torch.equal(sh, s0) and torch.equal(zh, z0)
    torch.autograd.backward([sh, zh], [r2.to(dev, dtype), r1.to(dev, dtype)])
    check("z", zh, zr, dtype, scale=2.0)
    check("ga", ah.grad, ar.grad, dtype, scale=2.0)
    assert torch.equal(ah.grad, bh.grad)
    check("ggamma", gh.grad, gr.grad, dtype, scale=2.0)
    check("gbeta", beh.grad, ber.grad, dtype, scale=2.0)
    # only one of the two outputs used
    for use in (0, 1):
        a2, b2 = a.to(dev, dtype).requires_grad_(True), b.to(dev, dtype).requires_grad_(True)
        out = ops.add_layer_norm(a2, b2, gh, beh)[use]
        out.backward(r1.to(dev, dtype))
        if use == 0:
            assert torch.equal(a2.grad, r1.to(dev, dtype))
        else:
            a3 = s0.clone().requires_grad_(True)
            ops.layer_norm(a3, gh, beh).backward(r1.to(dev, dtype))
            assert torch.equal(a2.grad, a3.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 512, 768, 3072), (1, 37, 96, 200), (3, 64, 64, 256)])
def test_mlp_fused_gelu_epilogues(cfg, dtype):
    """ops.mlp = linear2(GELU(linear1(x))) with the GELU in linear1's GEMM epilogue and its derivative in linear2's data-gradient
    epilogue (dp_gemm_nt_gelu).  Forward: the composed ops' values (the pre-activation is rounded to the storage type before the
    GELU, as a separate kernel would read it).  Gradients: against the fp64 oracle, and not worse than the composed ops."""
    from dose_prediction_amd import ops
    dev = _dev()
    B, N, H, M = cfg
    x = q(rnd((B, N, H), 1), dtype)
    w1, b1 = q(rnd((M, H), 2, H ** -0.5), dtype), 0.1 * rnd((M,), 3)
    w2, b2 = q(rnd((H, M), 4, M ** -0.5), dtype), 0.1 * rnd((H,), 5)
    r = q(rnd((B, N, H), 6), dtype)
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    yr = oracle.linear(oracle.gelu(oracle.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    (yr * r.double()).sum().backward()

    def run(fused):
        t = [x.to(dev, dtype).requires_grad_(True)] + [v.to(dev).requires_grad_(True) for v in (w1, b1, w2, b2)]
        y = ops.mlp(*t) if fused else ops.linear(ops.gelu(ops.linear(t[0], t[1], t[2])), t[3], t[4])
        y.backward(r.to(dev, dtype))
        ops.flush_deferred()
        torch.cuda.synchronize()
        return y, t

    yf, tf = run(True)
    yc, tc = run(False)
    # (same roundings as the composed ops; the compiler may contract the GELU polynomial differently in the two kernels, which can
    # flip the last bit of a few activations, each of which reaches a row of outputs)
    assert float((yf != yc).float().mean()) < 2e-2 and rel_l2(yf.cpu().double(), yc.cpu().double()) < 1e-3
    s = 3.0
    check("y", yf, yr, dtype, scale=s)
    for name, a, c, g in zip(("gx", "gw1", "gb1", "gw2", "gb2"), tf, tc, ref):
        check(name, a.grad, g.grad, dtype, scale=s)
        ef, ec = rel_l2(a.grad.detach().cpu().double(), g.grad), rel_l2(c.grad.detach().cpu().double(), g.grad)
        assert ef <= 1.5 * ec + 1e-6, (name, ef, ec)


@pytest.mark.parametrize("C", [48, 1000, 1536])
def test_layernorm_widths(C):
    test_layernorm_gelu_add(torch.float32, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 64, 6, 128), (1, 512, 12, 64), (2, 8, 6, 8), (1, 144, 12, 4), (2, 40, 6, 128), (1, 100, 12, 64),
                                 (1, 1, 6, 64)])
def test_attention(cfg, dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    B, N, heads, d = cfg
    H = heads * d
    x = q(rnd((B, N, H), 1), dtype)
    wqkv = q(rnd((3 * H, H), 2, H ** -0.5), dtype)
    wo, bo = q(rnd((H, H), 3, H ** -0.5), dtype), 0.1 * rnd((H,), 4)
    xr, wr, wor, bor = (t.double().requires_grad_(True) for t in (x, wqkv, wo, bo))
    yr = oracle.attention(xr, wr, wor, bor, heads)
    r = q(rnd(yr.shape, 5), dtype)
    (yr * r.double()).sum().backward()
    xh = x.to(dev, dtype).requires_grad_(True)
    wh, woh, boh = (t.to(dev).requires_grad_(True) for t in (wqkv, wo, bo))
    yh = ops.linear(ops.attention(ops.linear(xh, wh), heads), woh, boh)
    yh.backward(r.to(dev, dtype))
    s = 3.0 if dtype != torch.float32 else 1.0     # intermediate qkv / P / O roundings in the 16-bit modes
    check("y", yh, yr, dtype, scale=s)
    check("gx", xh.grad, xr.grad, dtype, scale=s)
    check("gwqkv", wh.grad, wr.grad, dtype, scale=s)
    check("gwo", woh.grad, wor.grad, dtype, scale=s)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 512, 12, 64), (2, 512, 6, 128), (1, 1152, 12, 64), (1, 1152, 6, 128), (3, 77, 6, 64)])
def test_fused_attention_full_size(cfg, dtype):
    """dp_attention_fwd/bwd at the token counts of BASELINE.json (512 @128^3, 1152 @192x192x128) and a ragged one, against the
    float64 softmax(q k^T d^-1/2) v of the stored 16-bit qkv and against the unfused GEMM + softmax path of the same library."""
    from dose_prediction_amd import ops
    dev = _dev()
    B, N, heads, d = cfg
    H = heads * d
    qkv = q(rnd((B, N, 3 * H), 11), dtype)
    go = q(rnd((B, N, H), 12), dtype)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, N, heads, d).permute(0, 2, 1, 3) for t in x.split(H, dim=2))
    yr = (torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv).permute(0, 2, 1, 3).reshape(B, N, H)
    (yr * go.double()).sum().backward()
    xf = qkv.to(dev, dtype).requires_grad_(True)
    yf = ops.FusedAttention.apply(xf, heads)
    yf.backward(go.to(dev, dtype))
    xu = qkv.to(dev, dtype).requires_grad_(True)
    yu = ops.Attention.apply(xu, heads)
    yu.backward(go.to(dev, dtype))
    check("y", yf, yr, dtype)
    check("gqkv", xf.grad, x.grad, dtype, scale=2.0)
    # the fused kernels keep P in fp32 until the second MFMA: never farther from the reference than the unfused path (+ slack)
    yr = yr.detach()
    ef, eu = rel_l2(yf.detach().double().cpu(), yr), rel_l2(yu.detach().double().cpu(), yr)
    gf, gu = rel_l2(xf.grad.double().cpu(), x.grad), rel_l2(xu.grad.double().cpu(), x.grad)
    assert ef <= 1.5 * eu + 1e-4 and gf <= 1.5 * gu + 1e-4, (ef, eu, gf, gu)


def test_fused_attention_rejects_unsupported():
    from dose_prediction_amd import _lib
    dev = _dev()
    t = torch.zeros(1024, device=dev)
    with pytest.raises(_lib.DoseHipError):
        _lib.call("dp_attention_fwd", t.data_ptr(), t.data_ptr(), t.data_ptr(), 96, t.data_ptr(), 32, t.data_ptr(), 1, 1, 4, 32, 1.0, 1, 0)
    with pytest.raises(_lib.DoseHipError):
        _lib.call("dp_attention_fwd", t.data_ptr(), t.data_ptr(), t.data_ptr(), 192, t.data_ptr(), 64, t.data_ptr(), 1, 1, 4, 64, 1.0, 0, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(1000, 16, 16, 9, 16), (777, 8, 8, 24, 24), (4096, 32, 48, 5, 8), (3, 16, 16, 16, 16)])
def test_cat_two_inputs(cfg, dtype):
    """ops.cat of two row tensors (dp_cat2_rows, one pass) incl. sources that are channel slices of wider buffers; bit-exact."""
    from dose_prediction_amd import ops
    dev = _dev()
    rows, ca, lda, cb, ldb = cfg
    A = q(rnd((rows, lda), 1), dtype).to(dev, dtype)
    B = q(rnd((rows, ldb), 2), dtype).to(dev, dtype)
    a, b = A[:, :ca].requires_grad_(True), B[:, :cb].requires_grad_(True)
    y = ops.cat((a, b))
    ref = torch.cat((A[:, :ca], B[:, :cb]), dim=1)
    assert torch.equal(y.reshape(rows, ca + cb), ref)
    g = q(rnd((rows, ca + cb), 3), dtype).to(dev, dtype)
    ga, gb = torch.autograd.grad(y, (a, b), g.view(y.shape))
    assert torch.equal(ga.reshape(rows, ca), g[:, :ca]) and torch.equal(gb.reshape(rows, cb), g[:, ca:])


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify_posemb(dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    x = q(rnd((2, 5, 32, 16, 16), 1), dtype)
    pos = 0.1 * rnd((1, 2, 5 * 4096), 2)
    xr, pr = x.double().requires_grad_(True), pos.double().requires_grad_(True)
    yr = oracle.patchify(xr) + pr
    r = q(rnd(yr.shape, 3), dtype)
    (yr * r.double()).sum().backward()
    xp = torch.zeros((2, 32, 16, 16, 8), dtype=dtype)
    xp[..., :5] = ndhwc(x).to(dtype)
    xh, ph = xp.to(dev).requires_grad_(True), pos.to(dev).requires_grad_(True)
    yh = ops.add_broadcast(ops.patchify(xh, 5, 16), ph)
    yh.backward(r.to(dev, dtype))
    check("y", yh, yr, dtype)
    check("gx", ncdhw(xh.grad[..., :5]), xr.grad, dtype)
    check("gpos", ph.grad, pr.grad, dtype)
    assert xh.grad[..., 5:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_trilinear_cat_layout(dtype):
    from dose_prediction_amd import ops
    dev = _dev()
    x = q(rnd((2, 12, 3, 4, 5), 1), dtype)
    s = q(rnd((2, 4, 6, 8, 10), 2), dtype)
    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    yr = torch.cat((oracle.trilinear_up2(xr), sr), dim=1)
    r = rnd(yr.shape, 3)
    (yr * r.double()).sum().backward()
    xh, sh = x.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    a = ops.ToNDHWC.apply(xh, 16, dtype)[..., :12]
    b = ops.ToNDHWC.apply(sh, 8, dtype)[..., :4]
    yh = ops.FromNDHWC.apply(ops.cat((ops.trilinear_up2(a), b)))
    yh.backward(r.to(dev))
    check("y", yh, yr, dtype)
    check("gx", xh.grad, xr.grad, dtype)
    check("gs", sh.grad, sr.grad, dtype)


@pytest.mark.parametrize("cfg", [(2, 32, 5, 9, 20), (1, 64, 3, 4, 16), (1, 8, 2, 33, 70), (1, 16, 7, 16, 32)])
def test_trilinear_lds_path_matches_the_gather_kernel_bit_for_bit(cfg, monkeypatch):
    """dp_trilinear_up2_fwd through LDS (16-bit storage, C <= 64: the C3D decoder's UpConv at the 128^3 / 64^3 levels, c3d.py:36) against
    the oracle, and bit-identical to the one-thread-per-piece gather kernel it replaces (same weights, same FMA order)."""
    import subprocess
    import sys
    from dose_prediction_amd import ops
    dev = _dev()
    N, C, D, H, W = cfg
    for dtype in (torch.bfloat16, torch.float16):
        x = q(rnd((N, C, D, H, W), 11), dtype)
        yr = oracle.trilinear_up2(x.double())
        xh = ndhwc(x).to(dev, dtype)
        y = ops.trilinear_up2(xh)
        check("y", ncdhw(y), yr, dtype)
        # the gather kernel (DP_TRILINEAR_LDS=0 is read once per process: run it in a child)
        code = ("import os, sys, torch; sys.path.insert(0, %r); os.environ['DP_TRILINEAR_LDS'] = '0'; from dose_prediction_amd import ops; "
                "x = torch.load(sys.argv[1]).cuda(); torch.save(ops.trilinear_up2(x).cpu(), sys.argv[2])") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            torch.save(xh.cpu(), os.path.join(td, "x.pt"))
            subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "x.pt"), os.path.join(td, "y.pt")])
            y0 = torch.load(os.path.join(td, "y.pt"))
        assert torch.equal(y.cpu().view(torch.int16), y0.view(torch.int16)), (cfg, dtype)


def test_argmax_onehot():
    from dose_prediction_amd import ops
    dev = _dev()
    lg = rnd((2, 5, 6, 7, 8), 1)
    lg[0, 0, 0, 0, :] = 0.5          # a tie: torch.argmax takes the first maximum
    ref = torch.nn.functional.one_hot(lg.argmax(-1), 8)[..., 1:].float()
    out = torch.zeros((2, 5, 6, 7, 16), device=dev)
    lab = ops.argmax_onehot(lg.to(dev), out, choff=1, labels=True)
    assert torch.equal(out[..., 1:8].cpu(), ref)
    assert torch.equal(lab.cpu().long(), lg.argmax(-1))
    assert out[..., 0].abs().max().item() == 0 and out[..., 8:].abs().max().item() == 0


@pytest.mark.parametrize("amsgrad", [True, False])
def test_fused_adam_matches_torch(amsgrad):
    """FusedAdam (dp_adam_multi) == torch.optim.Adam as NetworkTrainer.set_optimizer builds it (network_trainer.py:120-125)."""
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    shapes = [(16, 9, 3, 3, 3), (16,), (768, 3000), (5,), (1, 512, 768), (33333,)]
    ps = [rnd(s, 10 + i).to(dev) for i, s in enumerate(shapes)]
    a = [p.clone().requires_grad_(True) for p in ps]
    b = [p.clone().requires_grad_(True) for p in ps]
    kw = dict(lr=1e-3, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=amsgrad)
    oa, ob = torch.optim.Adam(a, **kw), FusedAdam(b, **kw)
    for it in range(4):
        for i, (x, y) in enumerate(zip(a, b)):
            g = rnd(x.shape, 100 * it + i).to(dev) * (0.1 if it == 2 else 1.0)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert rel_err(y.detach().cpu(), x.detach().cpu()) < 2e-6
    for x, y in zip(a, b):
        for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
            assert rel_err(ob.state[y][k].cpu(), oa.state[x][k].cpu()) < 2e-6, k


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # (N, Ca, Cb_logical, Cb_physical, Cout, D, H, W, k)
    (2, 16, 16, 16, 16, 3, 9, 40, 7), (1, 16, 9, 16, 16, 4, 10, 32, 3), (1, 32, 32, 32, 32, 2, 9, 33, 7),
    (1, 64, 64, 64, 64, 2, 6, 16, 3), (1, 16, 16, 16, 8, 2, 5, 20, 3), (1, 8, 8, 8, 16, 3, 5, 6, 3),
    (1, 16, 16, 16, 16, 2, 34, 40, 7),       # (K-along-H weight gradient, second input behind the channel split)
    (1, 16, 16, 16, 16, 4, 34, 40, 3),       # (the depth-marching 3^3 weight gradient with a split input)
    (2, 32, 32, 32, 64, 30, 29, 24, 3),      # (one-column tiles, two channel-tile groups per block, no kd split)
    (1, 16, 16, 16, 16, 5, 11, 96, 3),       # (W >= 96, Cout <= 16: the wave-private 3^3 kernel k_conv_w3, odd depth, ragged rows, split in and out)
    (2, 16, 9, 16, 16, 2, 8, 128, 3),        # (... second operand padded to 16 channels)
    (1, 16, 16, 16, 16, 2, 9, 96, 7),        # (the 25-slot 7^3 sweep over a split input; W = 96: k_conv_cc16w)
    (1, 16, 16, 16, 16, 3, 10, 192, 7), (2, 16, 9, 16, 16, 2, 8, 192, 3),      # (two 96-position tiles, split in and out)
    # the two operands with DIFFERENT row pitches (round 6: the slab staging multiplies a block-constant voxel index by the pitch of the operand
    # it reads -- k_conv_tiled's table -- or forms its column offsets per operand -- k_conv_cc16)
    (1, 32, 16, 16, 32, 2, 9, 40, 7), (1, 32, 16, 16, 16, 2, 9, 128, 3), (1, 16, 32, 32, 16, 3, 10, 128, 7), (2, 32, 9, 16, 64, 4, 12, 16, 3)])
def test_conv3d_virtual_concat(cfg, dtype):
    """conv3d((a, b)) == conv3d(cat(a, b)) of the oracle, forward and every gradient (dp_conv3d_tiled2 / dp_conv3d_wgrad_tiled2;
    the last config is too narrow for the tiled kernels and must take the materialised-cat fallback)."""
    from dose_prediction_amd import ops
    dev = _dev()
    N, Ca, Cb, Cbp, Cout, D, H, W, k = cfg
    a = q(rnd((N, Ca, D, H, W), 1), dtype)
    b = q(rnd((N, Cb, D, H, W), 2), dtype)
    w = q(rnd((Cout, Ca + Cb, k, k, k), 3, ((Ca + Cb) * k ** 3) ** -0.5), dtype)
    bias = 0.1 * rnd((Cout,), 4)
    ar, br, wr, biasr = (t.double().requires_grad_(True) for t in (a, b, w, bias))
    yr = oracle.conv3d(torch.cat((ar, br), dim=1), wr, biasr, 1, k // 2, 1)
    r = q(rnd(yr.shape, 5), dtype)
    (yr * r.double()).sum().backward()
    ah = ndhwc(a).to(dev, dtype).requires_grad_(True)
    bp = torch.zeros((N, D, H, W, Cbp), dtype=dtype)
    bp[..., :Cb] = ndhwc(b).to(dtype)
    bh = bp.to(dev).requires_grad_(True)
    wh, biash = w.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    yh = ops.conv3d((ah, bh), wh, biash, 1, k // 2, 1)
    yh.backward(ndhwc(r).to(dev, dtype))
    check("y", ncdhw(yh), yr, dtype)
    check("ga", ncdhw(ah.grad), ar.grad, dtype)
    check("gb", ncdhw(bh.grad[..., :Cb]), br.grad, dtype)
    if Cbp > Cb:
        assert bh.grad[..., Cb:].abs().max().item() == 0
    check("gw", wh.grad, wr.grad, dtype)
    check("gbias", biash.grad, biasr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(768, 96, 1024, 1), (100, 37, 200, 1), (64, 64, 2048, 4), (130, 70, 333, 2), (768, 12800, 300, 1),
                                 (700, 6000, 130, 2)])   # the last two: wide outputs (patch-embedding weight gradient shape class)
def test_gemm_tn(cfg, dtype):
    """dp_gemm_tn: C = A^T B with k-major operands (weight gradients of row-major layers), ragged tiles and split-K."""
    from dose_prediction_amd import _lib
    dev = _dev()
    M, N, K, sk = cfg
    lda, ldb = M + (8 - M % 8) % 8, N + (8 - N % 8) % 8
    A = q(rnd((K, lda), 1), dtype)
    B = q(rnd((K, ldb), 2), dtype)
    ref = A[:, :M].double().t() @ B[:, :N].double()
    Ad, Bd = A.to(dev, dtype), B.to(dev, dtype)
    C = (torch.zeros if sk > 1 else torch.empty)((M, N), dtype=torch.float32, device=dev)
    _lib.call("dp_gemm_tn", Ad.data_ptr(), lda, Bd.data_ptr(), ldb, C.data_ptr(), N, M, N, K, sk, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype],
              torch.cuda.current_stream().cuda_stream)
    assert rel_l2(C.cpu().double(), ref) < (2e-5 if dtype == torch.float32 else 2e-5)   # inputs are pre-rounded: fp32 accumulation in both modes


# ------------------------------------------------------------------------------------------------ packed weights follow the optimizer
def _all_pack_users(dev, dtype):
    """One parameter per pack layout with a closure that makes ops build (and use) its packed copies."""
    from dose_prediction_amd import ops
    mk = lambda shape, seed, s=0.2: torch.nn.Parameter((rnd(shape, seed) * s).to(dev))  # noqa: E731
    x16 = ndhwc(rnd((1, 16, 4, 9, 32), 70)).to(dev).to(dtype).requires_grad_(True)
    x24 = ndhwc(rnd((1, 24, 4, 9, 32), 71)).to(dev).to(dtype).requires_grad_(True)
    x9 = ndhwc(rnd((1, 9, 6, 6, 6), 72)).to(dev).to(dtype).requires_grad_(True)
    tok = rnd((2, 40, 96), 73).to(dev).to(dtype).requires_grad_(True)
    tok2 = rnd((2, 40, 100), 74).to(dev).to(dtype).requires_grad_(True)
    x16w = ndhwc(rnd((1, 16, 2, 9, 96), 75)).to(dev).to(dtype).requires_grad_(True)
    users = [
        (mk((16, 16, 7, 7, 7), 1, 0.05), lambda w: ops.conv3d(x16, w, None, 1, 3, 1)),        # tiled, tap-paired (Cout <= 16)
        (mk((16, 16, 7, 7, 7), 9, 0.05), lambda w: ops.conv3d(x16w, w, None, 1, 3, 1)),       # cc16 layout (W >= 96), 25 slots
        (mk((8, 16, 3, 3, 3), 10), lambda w: ops.conv3d(x16w, w, None, 1, 1, 1)),             # cc16 layout, 3^3
        (mk((40, 24, 3, 3, 3), 2), lambda w: ops.conv3d(x24, w, None, 1, 1, 1)),              # tiled, two N tiles, ragged channels
        (mk((12, 9, 3, 3, 3), 3), lambda w: ops.conv3d(x9, w, None, 2, 1, 1)),                # generic (stride 2): modes 0 and 1
        (mk((12, 9, 3, 3, 3), 4), lambda w: ops.conv3d(x9, w, None, 1, 2, 2)),                # generic dilated: modes 0 and 2
        (mk((8, 16, 1, 1, 1), 5), lambda w: ops.conv3d(x16, w, None, 1, 0, 1)),               # pointwise
        (mk((16, 12, 2, 2, 2), 6), lambda w: ops.conv_transpose2x(x16, w)),                   # ConvTranspose matrices
        (mk((64, 96), 7), lambda w: ops.linear(tok, w)),                                      # Linear, no padding (cast)
        (mk((30, 100), 8), lambda w: ops.linear(tok2, w)),                                    # Linear, padded both ways
    ]
    return users


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_multi_matches_single_tensor_packs(dtype):
    """ops.refresh_packs (ONE dp_pack_multi launch) rebuilds every packed copy bit-identically to the per-tensor pack entry
    points it replaces, for every layout (conv generic modes 0-2, tiled / tiled transposed+flipped, tconv, Linear +/- transposed)."""
    from dose_prediction_amd import ops
    dev = _dev()
    users = _all_pack_users(dev, dtype)
    for w, f in users:
        y = f(w)
        y.backward(torch.ones_like(y))
    n_entries = 0
    for w, f in users:
        with torch.no_grad():
            w.data.mul_(-1.7).add_(0.01)         # through .data: the version counter does not move, exactly like the fused optimizer
        stale = {k: e[1].clone() for k, e in w._dp_packs.items()}
        ops.refresh_packs([w])
        multi = {k: e[1].clone() for k, e in w._dp_packs.items()}
        assert set(multi) == set(stale) and len(multi) >= 1
        ops.invalidate_packs([w])                  # now the per-tensor builders
        y = f(w)
        y.backward(torch.ones_like(y))
        for k, e in w._dp_packs.items():
            if e[1].data_ptr() == w.data_ptr():
                continue
            n_entries += 1
            assert torch.equal(e[1].view(torch.uint8), multi[k].view(torch.uint8)), (k, dtype)
            assert not torch.equal(e[1].view(torch.uint8), stale[k].view(torch.uint8)), (k, dtype)
    assert n_entries >= 14


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_adam_step_reaches_the_packed_weights(dtype):
    """FusedAdam writes parameters through raw pointers; the forward after step() must use the NEW weights
    (== a forward with freshly packed weights), for conv / tconv / linear in 16-bit and fp32 storage."""
    from dose_prediction_amd import ops
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    users = _all_pack_users(dev, dtype)
    opt = FusedAdam([w for w, _ in users], lr=5e-2, amsgrad=True)
    before = []
    for w, f in users:
        y = f(w)
        y.float().pow(2).sum().backward()
        before.append(y.detach().float().clone())
    opt.step()
    for (w, f), y0 in zip(users, before):
        y1 = f(w).detach().float()
        ops.invalidate_packs([w])
        y2 = f(w).detach().float()
        # (not bit-equal in fp32: the split-kd convolution accumulates with fp32 atomics; the packs themselves are compared bit for
        #  bit in test_pack_multi_matches_single_tensor_packs)
        assert (y1 - y2).abs().max() <= 1e-5 * y2.abs().max(), "forward after FusedAdam.step() does not use freshly packed weights"
        assert (y1 - y0).abs().max() > 1e-3 * y0.abs().max(), "forward did not change after the optimizer step"


def test_fused_adam_state_dict_round_trips_with_torch_adam():
    """NetworkTrainer saves / restores optimizer_state_dict (network_trainer.py:340-363): FusedAdam must resume from a
    torch.optim.Adam checkpoint (tensor `step`s) and vice versa, and continue identically."""
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    shapes = [(8, 4, 3, 3, 3), (8,), (40, 24)]
    kw = dict(lr=1e-3, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)

    def run(cls_a, cls_b):
        pa = [rnd(s, 10 + i).to(dev).requires_grad_(True) for i, s in enumerate(shapes)]
        pb = [p.detach().clone().requires_grad_(True) for p in pa]
        oa, ob = cls_a(pa, **kw), cls_b(pb, **kw)
        for it in range(2):
            for i, p in enumerate(pa):
                p.grad = rnd(p.shape, 100 * it + i).to(dev)
            oa.step()
        ob.load_state_dict(copy.deepcopy(oa.state_dict()))      # (state_dict() hands out references to the live state tensors)
        with torch.no_grad():
            for x, y in zip(pa, pb):
                y.copy_(x)
        for it in range(2, 4):
            for i, (x, y) in enumerate(zip(pa, pb)):
                g = rnd(x.shape, 100 * it + i).to(dev)
                x.grad, y.grad = g.clone(), g.clone()
            oa.step()
            ob.step()
        for x, y in zip(pa, pb):
            assert rel_err(y.detach().cpu(), x.detach().cpu()) < 2e-6
        assert FusedAdam._step_value(ob.state[pb[0]]["step"]) == 4

    run(torch.optim.Adam, FusedAdam)
    run(FusedAdam, torch.optim.Adam)
    # without amsgrad the state carries no max_exp_avg_sq key (torch's Adam has none either)
    p = rnd((5, 3), 1).to(dev).requires_grad_(True)
    o = FusedAdam([p], lr=1e-3)
    p.grad = torch.ones_like(p)
    o.step()
    assert "max_exp_avg_sq" not in o.state[p]


def test_fused_adam_capturable_replays_correct_steps():
    """a step captured in a HIP graph must keep advancing Adam's bias corrections (device-side step count)."""
    from dose_prediction_amd.optim import FusedAdam
    dev = _dev()
    kw = dict(lr=1e-2, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)
    p0 = rnd((300, 70), 1).to(dev)
    g = rnd((300, 70), 2).to(dev)
    a = p0.clone().requires_grad_(True)
    b = p0.clone().requires_grad_(True)
    oa, ob = torch.optim.Adam([a], **kw), FusedAdam([b], capturable=True, **kw)
    a.grad, b.grad = g.clone(), g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ob.step()                       # eager warm-up step (allocates state)
    torch.cuda.current_stream().wait_stream(side)
    oa.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ob.step()
    oa.step()                           # the capture pass itself is not executed...
    graph.replay()                      # ...this replay is step 2
    for _ in range(3):
        oa.step()
        graph.replay()
    torch.cuda.synchronize()
    assert int(ob.state[b]["step"].item()) == 5
    assert rel_err(b.detach().cpu(), a.detach().cpu()) < 5e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # (N, Cin, ca (virtual-concat split or 0), Cout, D, H, W, k)
    (2, 16, 0, 16, 5, 19, 70, 7),       # tap-paired epilogue, ragged H and W tiles
    (1, 32, 16, 16, 4, 9, 130, 3),      # virtual concat, TWC = 4
    (2, 16, 0, 16, 5, 20, 96, 7),       # 96-position tiles (k_conv_cc16w): statistics per (d, 8-row tile, 96-position tile), odd depth
    (1, 32, 16, 16, 4, 13, 192, 3),     # ... two tiles per row, virtual concat, ragged H
    (2, 16, 0, 32, 50, 61, 40, 3),      # NPAIR == 1, one N tile (>= 400 blocks: no split-kd), ragged H
    (1, 24, 0, 72, 20, 78, 33, 3),      # two N tiles per block + a second channel block (grid.y), ragged Cout / H
    (4, 32, 0, 64, 100, 16, 16, 3),     # W16 tiles (two image rows per MFMA tile)
    (2, 64, 0, 64, 32, 27, 24, 3),      # one-column tiles, 8-row blocks with two channel-tile groups (no kd split): ragged H and W
    (2, 32, 16, 128, 26, 32, 32, 3),    # ... four channel tiles (two block rows), virtual concat
    (1, 128, 0, 64, 4, 8, 16, 7),       # split-kd volume: statistics fall back to the row pass
    (1, 12, 0, 8, 6, 6, 6, 3)])         # too small for the tiled kernel: generic convolution + row pass
def test_conv_epilogue_statistics(cfg, dtype):
    """conv3d(..., stats=True): the per-block (sum, sum of squares) rows written by the convolution epilogue (from its accumulators
    rounded to the storage type) give the InstanceNorm / BatchNorm statistics of the stored output; checked through norm_act against the oracle's
    conv -> norm -> act in float64, forward and backward, for both statistics modes.  (Mish, not ReLU: with millions of elements a few
    normalised values land within round-off of 0, where ReLU's derivative jumps and the comparison would measure that instead.)"""
    from dose_prediction_amd import ops
    dev = _dev()
    N, Cin, ca, Cout, D, H, W, k = cfg
    x = q(rnd((N, Cin, D, H, W), 1), dtype)
    w = q(rnd((Cout, Cin, k, k, k), 2, (Cin * k ** 3) ** -0.5), dtype)
    b = 0.3 * rnd((Cout,), 3)
    for kind in ("instance", "batch"):
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yr = oracle.conv3d(xr, wr, b.double(), 1, k // 2, 1)
        if kind == "instance":
            zr = oracle.activation(oracle.instance_norm(yr), "mish")
        else:
            zr = oracle.activation(oracle.batch_norm(yr, torch.ones(Cout).double(), torch.zeros(Cout).double(), torch.zeros(Cout).double(),
                                                     torch.ones(Cout).double(), True)[0], "mish")
        r = rnd(zr.shape, 4)
        zr.backward(r.double())
        xd = ndhwc(x).to(dev).to(dtype).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        if ca:
            xa, xb = xd[..., :ca].detach().contiguous().requires_grad_(True), xd[..., ca:].detach().contiguous().requires_grad_(True)
            y, st = ops.conv3d((xa, xb), wd, b.to(dev), 1, k // 2, 1, stats=True)
        else:
            y, st = ops.conv3d(xd, wd, b.to(dev), 1, k // 2, 1, stats=True)
        assert st.shape[0] == N and st.shape[2:] == (2, Cout) and not st.requires_grad
        V = D * H * W
        from dose_prediction_amd import _lib
        from_epilogue = _lib.lib().dp_conv3d_tiled_stat_blocks(N, D, H, W, Cin, Cout, k, Cout, ops._DT[dtype]) > 0
        assert from_epilogue == (D * H * W > 1000), "this configuration was meant to exercise the other statistics path"
        if from_epilogue:
            assert st.shape[1] == _lib.lib().dp_conv3d_tiled_stat_blocks(N, D, H, W, Cin, Cout, k, Cout, ops._DT[dtype])
        # both paths give the statistics of the tensor AS STORED (the epilogue rounds its fp32 accumulators to the storage type before
        # summing, ADVICE r2: mean / variance then belong to the values the normalisation kernels read): 1e-5 against sums of the
        # stored y; against the float64 oracle's unrounded output the rounding noise is ~ 2^-9 / sqrt(V)
        s = st.double().sum(1).cpu()
        ys = y.detach().double().cpu()
        st1, st2 = ys.sum(dim=(1, 2, 3)), (ys ** 2).sum(dim=(1, 2, 3))
        ref1, ref2 = yr.detach().sum(dim=(2, 3, 4)), (yr.detach() ** 2).sum(dim=(2, 3, 4))
        std = (ref2 / V - (ref1 / V) ** 2).clamp_min(1e-12).sqrt()
        assert ((s[:, 0] - st1).abs() / (V * std)).max() < 1e-5
        assert ((s[:, 1] - st2).abs() / st2).max() < 1e-5
        tol = 1e-5 if dtype == torch.float32 else 2e-3
        assert ((s[:, 0] - ref1).abs() / (V * std)).max() < tol
        assert ((s[:, 1] - ref2).abs() / ref2).max() < tol
        gam = torch.ones(Cout, device=dev)
        z = ops.norm_act(y, kind, gam if kind == "batch" else None, torch.zeros(Cout, device=dev) if kind == "batch" else None,
                         act="mish", stats=st)
        check(f"stats {kind} fwd", ncdhw(z), zr, dtype, scale=2.0)
        z.backward(ndhwc(r).to(dev).to(dtype))
        gx = torch.cat((xa.grad, xb.grad), -1) if ca else xd.grad
        check(f"stats {kind} gx", ncdhw(gx), xr.grad, dtype, scale=10.0)     # (conv -> norm -> relu chain: three rounded stages)
        check(f"stats {kind} gw", wd.grad, wr.grad, dtype, scale=10.0)
