#!/usr/bin/env python3
"""One training step of the default bench model (any mode) with EVERY C-ABI call bracketed by HIP events on one stream:
per entry point the count and total time, and for the entry points named on the command line every call with its integer arguments.

    python tools/list_calls.py fp32x3 dp_split_rows dp_gemm_nt        (GPU; side streams off so that a time is the kernel's own)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import _lib, losses, synth  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    names = set(sys.argv[2:])
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--dtype", mode, "--no-side-stream"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    shape = (128, 128, 128)
    net = bench.build_model(args, shape, dev)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)
    x, gt = synth.dose_input(2, shape).to(dev), synth.dose_target(2, shape).to(dev)
    rec = None
    orig = _lib.call

    def call(name, *a):
        if rec is None:
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = orig(name, *a)
        e1.record()
        rec.append((name, a, e0, e1, phase[0]))
        return rc
    _lib.call = call
    for m in list(sys.modules.values()):
        if getattr(m, "__name__", "").startswith("dose_prediction_amd") and hasattr(m, "_lib") and m is not _lib:
            pass        # modules call _lib.call through the module attribute: patched above
    phase = ["fwd"]
    for it in range(3):
        rec = [] if it == 2 else None
        opt.zero_grad(set_to_none=True)
        phase[0] = "fwd"
        out = net(x)
        loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
        phase[0] = "bwd"
        loss.backward()
        phase[0] = "opt"
        opt.step()
        torch.cuda.synchronize()
    tot = {}
    for name, a, e0, e1, ph in rec:
        ms = e0.elapsed_time(e1)
        t = tot.setdefault(name, [0, 0.0])
        t[0] += 1
        t[1] += ms
    print(f"mode {mode}: {len(rec)} calls, {sum(v[1] for v in tot.values()):.2f} ms between events")
    for name, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {ms:8.3f} ms {n:5d}  {name}")
    for name, a, e0, e1, ph in rec:
        if name in names:
            ints = [v for v in a if isinstance(v, int) and abs(v) < (1 << 31)]
            print(f"{ph} {name} {e0.elapsed_time(e1):7.3f} ms  {ints}")


if __name__ == "__main__":
    main()
