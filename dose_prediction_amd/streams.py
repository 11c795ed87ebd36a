"""The side streams of the hot path (transformer branch, 3x3x3 branch / small blocks, weight gradients), chosen so that they really run
beside the caller's stream and beside each other.

HIP multiplexes streams onto a handful of hardware queues (four per process by default; with GPU_MAX_HW_QUEUES=8 the DOSE-PYFER step
takes 35 instead of 25 ms, so that knob is no way out).  Two streams that land on the same queue execute in host-enqueue order: the
kernels of one wait behind everything enqueued earlier on the other, although no event says so (round 3 trace: the caller's stream sat
idle for the whole transformer backward, `tools/stream_timeline.py`).  Which streams share a queue depends on the order in which the
process first used its streams, so it cannot be planned; it can be MEASURED: a long spin kernel on one stream, a trivial launch on the
other, and the second finishes first only if the two do not share a queue.  `side_streams()` draws candidates from torch's stream pool
and keeps the first `n` that overlap with the caller's stream and with each other."""
import os

import torch

ROLE_VIT, ROLE_BRANCH, ROLE_WGRAD = 0, 1, 2
_CHOSEN = {}
_ROOT = {}
_PROBE = os.environ.get("DOSE_HIP_STREAM_PROBE", "1") != "0"
_SPIN = 400_000          # cycles of the spin kernel (~0.2 ms)
LAST_REPORT = []         # [(main stream id, [(candidate id, accepted)])] for tests / diagnosis


def _overlap(a, b):
    """True if a launch on b can finish while an earlier-enqueued kernel on a is still running."""
    ea = torch.cuda.Event(enable_timing=True)
    eb = torch.cuda.Event(enable_timing=True)
    for s in (a, b):                          # (the first launch on a stream creates its queue: milliseconds)
        with torch.cuda.stream(s):
            torch.cuda._sleep(100)
    torch.cuda.synchronize(a.device)
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN)
        ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(100)
        eb.record(b)
    torch.cuda.synchronize(a.device)
    return eb.elapsed_time(ea) > 0.0          # eb happened before ea


def side_streams(device, main, n=3):
    """n torch streams for the caller's stream `main` on `device` (cached per caller's stream)."""
    key = (device.index, main.cuda_stream)
    got = _CHOSEN.get(key)
    if got is not None and len(got) >= n:
        return got
    chosen, report = [], []
    if torch.cuda.is_current_stream_capturing():
        # no probing inside a capture (it synchronises): reuse the streams probed for another caller's stream of this device
        for (di, _h), prev in _CHOSEN.items():
            if di == device.index:
                chosen = [c for c in prev if c != main][:n]
                break
    elif _PROBE:
        spare = []
        for _ in range(16):
            c = torch.cuda.Stream(device=device)
            if c == main or c in chosen or c in spare:
                continue
            ok = all(_overlap(s, c) and _overlap(c, s) for s in [main] + chosen)
            report.append((c.cuda_stream, ok))
            (chosen if ok else spare).append(c)
            if len(chosen) >= n:
                break
        chosen += spare[:n - len(chosen)]          # fewer independent queues than roles: take what there is
    while len(chosen) < n:
        chosen.append(torch.cuda.Stream(device=device))
    LAST_REPORT.append((main.cuda_stream, report))
    _CHOSEN[key] = chosen
    return chosen


def root(stream):
    """The caller's stream a side stream was chosen for (the stream itself if it is not one of ours): code that runs ON a side stream
    (the 3x3x3 branch's backward, blocks inside the transformer branch) asks for the side streams of the same caller."""
    return _ROOT.get((stream.device.index, stream.cuda_stream), stream)


def side_stream(device, main, role):
    """Side stream `role` (ROLE_VIT / ROLE_BRANCH / ROLE_WGRAD) for the caller's stream behind `main`."""
    main = root(main)
    got = side_streams(device, main)
    for s in got:
        _ROOT.setdefault((s.device.index, s.cuda_stream), main)
    return got[role]
