"""Run-time configuration of the HIP path."""
import os

import torch

_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16}
_X3_NAMES = ("fp32x3", "f32x3", "x3", "bf16x3")
_env_mode = os.environ.get("DOSE_HIP_DTYPE", "fp32").lower()
_x3 = _env_mode in _X3_NAMES
_compute_dtype = torch.float32 if _x3 else _DTYPES[_env_mode]


def set_compute_dtype(dtype):
    """torch.float32: parity mode (exact-fp32 MFMA); torch.bfloat16: benchmark mode (bf16 MFMA, fp32 accumulate);
    torch.float16: fp16 storage + fp16 MFMA with fp32 accumulation (BASELINE.json configs[4]).
    "fp32x3": the FAST tolerance-meeting mode -- fp32 storage like the parity mode, but the 3x3x3 / 7x7x7 convolutions and the Linear
    layers run on the bf16 matrix cores with every fp32 operand split into two bf16 halves (x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo,
    fp32 accumulation; csrc/x3.hip).  compute_dtype() is torch.float32 in that mode and x3() is True."""
    global _compute_dtype, _x3
    x3 = False
    if isinstance(dtype, str):
        if dtype.lower() in _X3_NAMES:
            dtype, x3 = torch.float32, True
        else:
            dtype = _DTYPES[dtype.lower()]
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be float32, bfloat16, float16 or 'fp32x3'")
    _compute_dtype, _x3 = dtype, x3


def compute_dtype():
    """Storage type of the activations."""
    return _compute_dtype


def x3():
    """True in the "fp32x3" mode (fp32 storage, split-bf16 matrix-core arithmetic)."""
    return _x3


def compute_mode():
    """Name of the current mode: 'fp32', 'fp32x3', 'bf16' or 'fp16' (accepted by set_compute_dtype)."""
    if _x3:
        return "fp32x3"
    return {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[_compute_dtype]


class compute_mode_as:
    """Context: run a region in another mode (the cascade runs its no-grad segmentation network in 'fp32x3' so that the OAR masks
    are the reference's whatever storage type the dose network trains in)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = compute_mode()
        set_compute_dtype(self.mode)

    def __exit__(self, *a):
        set_compute_dtype(self.prev)


_loss_scale = 1.0
_checkpoint_decoder = False


def set_loss_scale(scale):
    """Static loss scale for 16-bit storage (fp16: BASELINE.json configs[4]).  The module boundary (ops.FromNDHWC.backward) multiplies
    the incoming fp32 gradient by `scale` while converting it to the storage type, so every stored activation gradient is `scale`
    times larger (out of the fp16 subnormal range); parameter gradients come out multiplied by `scale` and are divided again
    inside FusedAdam (grad_scale defaults to this value) or with ops.unscale_grads(params) before any other optimizer.  1.0 = off."""
    global _loss_scale
    if not (scale > 0):
        raise ValueError("loss scale must be positive")
    _loss_scale = float(scale)


def loss_scale():
    return _loss_scale


def effective_loss_scale():
    """The scale actually applied to the backward pass: set_loss_scale()'s value while the storage type is 16-bit, 1.0 in fp32 storage
    (there the boundary gradient is not scaled, so nothing may be divided out again).  The ONE place every consumer asks:
    ops.FromNDHWC / ToNDHWC, FusedAdam, ops.unscale_grads."""
    return _loss_scale if _compute_dtype in (torch.bfloat16, torch.float16) else 1.0


def set_activation_checkpointing(on):
    """Recompute the four pyramid decoder blocks in the backward pass (torch.utils.checkpoint around each ModifiedUnetrUpBlock /
    UnetrUpBlock of PyMSCDecoder): their 128^3..16^3 intermediates are the bulk of the saved activations (BASELINE.json configs[4])."""
    global _checkpoint_decoder
    _checkpoint_decoder = bool(on)


def activation_checkpointing():
    return _checkpoint_decoder


_bn_updates = True


class bn_buffer_updates:
    """Context: whether training-mode BatchNorm layers update running_mean / running_var / num_batches_tracked (switched off while
    an activation-checkpointed block is recomputed in the backward pass)."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global _bn_updates
        self.prev, _bn_updates = _bn_updates, self.on

    def __exit__(self, *a):
        global _bn_updates
        _bn_updates = self.prev


def bn_updates_enabled():
    return _bn_updates


_vit_side_stream = os.environ.get("DOSE_HIP_SIDE_STREAM", "1") != "0"


def set_vit_side_stream(on):
    """Run the ViT branch of ViTEncoder / OAR-TRANSEG (small-grid, latency-bound token kernels) on a second HIP stream so that it
    overlaps the independent full-chip 128^3 `skip1` / `encoder1` branch, forward and (through autograd's stream bookkeeping)
    backward.  On by default; DOSE_HIP_SIDE_STREAM=0 or set_vit_side_stream(False) serialises everything on the current stream."""
    global _vit_side_stream
    _vit_side_stream = bool(on)


def vit_side_stream():
    return _vit_side_stream


_cascade_seg_mode = os.environ.get("DOSE_HIP_CASCADE_SEG_MODE", "fp32x3")


def set_cascade_seg_mode(mode):
    """Mode in which the cascade (dose_prediction_amd.cascade) runs its no-grad OAR-TRANSEG forward, whatever mode the dose network
    trains in: 'fp32x3' (default: the OAR masks fed to DOSE-PYFER are the reference's, arg-max exact off near-ties), 'fp32', or None =
    the current mode (bf16 storage flips ~0.8 % of the arg-max voxels of a random-init network: profiles/r02_f_bench_line_transeg.json).
    Env DOSE_HIP_CASCADE_SEG_MODE ('same' = None)."""
    global _cascade_seg_mode
    if mode is not None and mode != "same":
        prev = compute_mode()
        set_compute_dtype(mode)           # validates the name
        set_compute_dtype(prev)
    _cascade_seg_mode = None if mode in (None, "same") else mode


def cascade_seg_mode():
    return None if _cascade_seg_mode in (None, "same") else _cascade_seg_mode


def _env_terms(name, default):
    v = os.environ.get(name, str(default))
    if v not in ("1", "3"):
        raise ValueError(f"{name} must be 1 or 3, got {v!r}")
    return int(v)


_x3_wgrad_terms = _env_terms("DOSE_HIP_X3_WGRAD_TERMS", 1)
_x3_linear_wgrad_terms = _env_terms("DOSE_HIP_X3_LINEAR_WGRAD_TERMS", 1)
_x3_dgrad_terms = _env_terms("DOSE_HIP_X3_DGRAD_TERMS", 1)


def set_x3_wgrad_terms(n):
    """fp32x3 mode: number of split products in the WEIGHT gradients.  1 (default): x_hi gy_hi -- the weight gradients (and only they)
    are formed from bf16-rounded operands with fp32 accumulation; the forward pass always uses the three products, so outputs keep
    their 1e-4 parity (data gradients: set_x3_dgrad_terms).  3: x_hi gy_hi + x_lo gy_hi + x_hi gy_lo like every other contraction of the mode
    (10.5 ms per DOSE-PYFER step more).  Why 1 is the default: a weight-gradient element is a sum over millions of voxels, the
    operand rounding is unbiased, and what it adds (~1.6e-3 relative per element) is below what separates the EXACT fp32 mode from
    float64 on the same gradients (4.2e-3, ReLU gates of pre-activations within round-off of zero).  Measured at production width
    (tools/probes/x3_grad_probe.py, tools/probes/x3_trajectory_probe.py): gradient vector vs float64 1.05e-2 (1 product) / 1.03e-2 (3 products);
    parameters after six Adam steps 7.6 % of the update away from an exact-fp32 run in BOTH cases (two exact-fp32 runs: 4.6 %,
    bf16: 44 %); the reference trainer's G6 sequence passes in both (tests/test_trainer_sequence.py)."""
    global _x3_wgrad_terms
    if n not in (1, 3):
        raise ValueError("x3 weight-gradient terms must be 1 or 3")
    _x3_wgrad_terms = n


def x3_wgrad_terms():
    return _x3_wgrad_terms


def set_x3_dgrad_terms(n):
    """fp32x3 mode: number of split products in the DATA gradients of the convolutions and Linear layers.  1 (default since round 4):
    gy_hi w_hi only (a DP_X1 launch: the bf16 kernels on the hi halves, fp32 result); 3: like the forward pass (+7 ms per DOSE-PYFER
    step).  The forward pass -- every output, loss and validation metric, i.e. what the north-star's 1e-3 / arg-max bar is stated on --
    always uses three products.  Why 1 is the default: measured at production width (tools/probes/x3_grad_probe.py, 64^3, all 148 trainable
    tensors against the float64 oracle) the gradient vector is 1.09e-2 away with one product and 1.04e-2 with three -- the floor is set
    by activation gates of pre-activations within round-off of zero, not by the backward arithmetic -- against 2.6e-1 in the bf16 mode,
    whose error comes from its FORWARD activations; six fused-Adam steps end 8.0 % of the update away from an exact-fp32 run with one
    product, 8.1 % with three (two exact-fp32 runs: 5.4 %; bf16: 40 %) (tools/probes/x3_trajectory_probe.py).  bench.py reports the
    three-product variants beside the default (fp32_mode.dgrad_three_products, .all_three_products)."""
    global _x3_dgrad_terms
    if n not in (1, 3):
        raise ValueError("x3 data-gradient terms must be 1 or 3")
    _x3_dgrad_terms = n


def x3_dgrad_terms():
    return _x3_dgrad_terms


def set_x3_linear_wgrad_terms(n):
    """fp32x3 mode: split products in the weight gradients of the LINEAR layers (transformer, patch embedding): 1 = x_hi gy_hi (default
    since round 4), 3 = all three.  Their contraction runs over 1-2 k token rows, not over millions of voxels, so the averaging argument
    of set_x3_wgrad_terms does not carry over -- round 3 therefore defaulted to 3 until there was per-layer evidence.  That
    evidence (tools/probes/x3_grad_probe.py, the 33 Linear weight tensors of DOSE-PYFER at production width, each against the float64 oracle):
    relative L2 error per tensor median 1.56e-2 / max 1.82e-2 with three products, 1.60e-2 / 1.85e-2 with one; all 33 together 1.81e-2
    vs 1.84e-2; six Adam steps end 8.1 % of the update away from the exact-fp32 run in both cases.  The gradient error of these layers is
    set upstream (activation gates flipping within round-off of zero), not by the operand rounding of their own contraction."""
    global _x3_linear_wgrad_terms
    if n not in (1, 3):
        raise ValueError("x3 weight-gradient terms must be 1 or 3")
    _x3_linear_wgrad_terms = n


def x3_linear_wgrad_terms():
    return _x3_linear_wgrad_terms


_branch_stream = os.environ.get("DOSE_HIP_BRANCH_STREAM", "1") != "0"


def set_branch_stream(on):
    """Run the 3x3x3 branch of every multi-scale block (blocks_MDUNet.conv_3_1: conv_3 || conv_7 on the same input) on a second HIP
    stream beside its 7x7x7 branch, and the 64^3 .. 16^3 up-sampling blocks of the encoder (skip2..4 / encoder2..4) beside the 128^3
    block, forward and (through autograd's stream bookkeeping) backward.  On by default (DOSE-PYFER step 26.4 -> 25.5 ms, OAR-TRANSEG
    24.0 -> 23.4); env DOSE_HIP_BRANCH_STREAM=0 / set_branch_stream(False) keeps those kernels on the caller's stream."""
    global _branch_stream
    _branch_stream = bool(on)


def branch_stream():
    return _branch_stream


_wgrad_stream = os.environ.get("DOSE_HIP_WGRAD_STREAM", "1") != "0"


def set_wgrad_stream(on):
    """Launch the convolutions' weight-gradient kernels on a HIP stream of their own (one per device).  Nothing inside the backward pass
    reads a weight gradient, so they leave the critical path dgrad -> normalisation backward -> dgrad ...: the MFMA-bound 7x7x7 /
    3x3x3 weight gradients then run beside the HBM-bound normalisation passes of the layers in front.  The stream is joined before
    anybody reads a gradient: at the end of the backward pass (autograd engine callback), in FusedAdam.step() and before the
    data-parallel reducer launches a bucket.  Not used for a parameter that already holds a .grad (accumulation over micro-batches
    adds on the caller's stream).  On by default; env DOSE_HIP_WGRAD_STREAM=0 / set_wgrad_stream(False) keeps them on the caller's
    stream."""
    global _wgrad_stream
    _wgrad_stream = bool(on)


def wgrad_stream():
    return _wgrad_stream


_capture_side_streams = os.environ.get("DOSE_HIP_CAPTURE_SIDE_STREAMS", "1") != "0"


def set_capture_side_streams(on):
    """Keep the weight-gradient stream (and, as before, the transformer's stream) while a step is being CAPTURED into a HIP graph: the
    forks (event recorded on the capturing stream, waited for on the side stream) and joins become dependency edges of the graph, so a
    replay runs them side by side without the launch thread.  The 3x3x3 branch / small-block stream is NOT kept: with its many fork /
    join pairs per pass hipStreamEndCapture of ROCm 7.2 dies with a segmentation fault (no HIP error is reported first; round 4,
    `bench.py --graph`), so branch_stream_allowed() stays False inside a capture.  The side streams are the ones chosen (and probed for
    hardware-queue collisions, streams.py) for the stream the capture runs on, so warm up eagerly ON that stream and pass it to
    torch.cuda.graph(..., stream=...).  On by default; env DOSE_HIP_CAPTURE_SIDE_STREAMS=0 / set_capture_side_streams(False): one stream."""
    global _capture_side_streams
    _capture_side_streams = bool(on)


def side_streams_allowed():
    """False only while the current stream is capturing AND capture-time side streams are switched off."""
    return _capture_side_streams or not torch.cuda.is_current_stream_capturing()


_capture_branch = os.environ.get("DOSE_HIP_CAPTURE_BRANCH", "0") == "1"      # (experiment: keep the branch stream inside a capture)


def branch_stream_allowed():
    """The 3x3x3 branch / small-block stream: never inside a capture (see set_capture_side_streams), unless DOSE_HIP_CAPTURE_BRANCH=1
    asks for the experiment (tools/r06_graph_repro.sh)."""
    return _capture_branch or not torch.cuda.is_current_stream_capturing()


_deterministic = 0


def set_deterministic(on):
    """Bit-reproducible passes on demand (the reference's CPU path is bit-repeatable, the parity gates run under this).
    By default a handful of reductions meet in fp32 atomics, whose order -- hence the last bit of the sum -- differs from run to run:
    split-kd convolutions of small volumes, split-K GEMMs (patch embedding), the tap-major scratch of the 7^3 / 3^3 weight-gradient
    kernels, the generic weight gradient, LayerNorm's dgamma / dbeta, the trilinear up-sampling's backward scatter.  With the switch
    on each of them takes a fixed-order path (csrc: dp_set_deterministic -- one scratch slab per kd / chunk share of a split-kd convolution
    and per voxel share of a weight-gradient kernel, added in share order by the finish / unpack pass; split-K GEMMs as a batched GEMM
    over K shares + an ordered sum; per-block partial rows + an fp64 combine for LayerNorm; a gather instead of the trilinear scatter);
    everything else (per-block statistics partials, the 3^3 marching weight gradient, the row-stream weight gradients, the grouped
    transformer weight-gradient launch, losses) was order-fixed already.  Same arithmetic, same tolerances, the same kernels and launch
    geometry: the DOSE-PYFER 128^3 bf16 step measures 23.6 ms with the switch on, 23.8 off (one box, round 5).
    Process-wide (the flag lives in libdose_hip.so); off by default, env DOSE_HIP_DETERMINISTIC=1 switches it on at import."""
    global _deterministic
    from . import _lib
    # (an int other than 0 / 1 is a mask of single sites, for experiments -- tools/probes/determinism_probe.py: 1 split-kd convolutions, 2 split-K
    # GEMMs, 4 tiled weight gradients, 8 generic weight gradient, 16 trilinear backward, 32 LayerNorm dgamma / dbeta)
    mask = 0x7fffffff if on is True or on == 1 else int(on or 0)
    _lib.lib().dp_set_deterministic(mask)
    _deterministic = mask


def deterministic(site=0x7fffffff):
    return bool(_deterministic & site)


class deterministic_as:
    """Context: run a region with set_deterministic(on)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _deterministic
        set_deterministic(self.on)

    def __exit__(self, *a):
        set_deterministic(self.prev)


if os.environ.get("DOSE_HIP_DETERMINISTIC", "0") == "1":
    set_deterministic(True)


def set_backward_on_calling_thread(on=True):
    """Run the backward pass on the thread that calls loss.backward() (torch.autograd.set_multithreading_enabled(not on)) instead of the
    autograd engine's per-device worker thread.  One process drives one GPU here and every backward node is a launch, so the worker
    thread buys nothing, and the hand-over costs the launch thread 1.2 ms of a 14-ms step (tools/host_profile.py, round 6: 14.0 -> 12.8 ms)
    -- host time that matters as soon as the data-parallel reducer's hooks share that thread.  A process-wide torch setting, so it is NOT
    switched at import: bench.py switches it on (config.backward_on_calling_thread in its line); a trainer does it once at start-up
    (INTEGRATION.md).  Same graph, same kernels, same results."""
    torch.autograd.set_multithreading_enabled(not on)


def backward_on_calling_thread():
    return not torch.autograd.is_multithreading_enabled()
