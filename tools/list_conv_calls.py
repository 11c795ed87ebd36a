#!/usr/bin/env python3
"""One bf16 training step of the default bench model; prints every tiled convolution launch (forward / data gradient) with its
shape, pitches and HIP-event time (the side stream is off so that the times are those of the kernels alone)."""
import os
import sys

import torch

os.environ.setdefault("DOSE_HIP_SIDE_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dose_prediction_amd import _lib, losses, synth  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--no-cpu-baseline"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    shape = (128, 128, 128)
    net = bench.build_model(args, shape, dev)
    x, gt = synth.dose_input(2, shape).to(dev), synth.dose_target(2, shape).to(dev)
    for it in range(3):
        _lib.PROFILE = [] if it == 2 else None
        out = net(x)
        loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True)
        loss.backward()
        torch.cuda.synchronize()
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, a, e0, e1 in _lib.PROFILE:
        ms = e0.elapsed_time(e1)
        if name == "dp_conv3d_tiled":            # x, ldx, wq, bias, y, ldy, ws, N, D, H, W, Cin, Cout, k
            print(f"tiled   {ms:7.3f} ms  ldx={a[1]} ldy={a[5]} N={a[7]} DHW={a[8:11]} Cin={a[11]} Cout={a[12]} k={a[13]}")
        elif name == "dp_conv3d_tiled2":         # x, ldx, x2, ldx2, csplit, wq, bias, y, ldy, y2, ldy2, osplit, ws, N, D, H, W, Cin, Cout, k
            print(f"tiled2  {ms:7.3f} ms  ldx={a[1]} x2={'y' if a[2] else 'n'} ldx2={a[3]} csplit={a[4]} ldy={a[8]} y2={'y' if a[9] else 'n'} N={a[13]} DHW={a[14:17]} Cin={a[17]} Cout={a[18]} k={a[19]}")
        elif name == "dp_conv3d_tiled_stats":    # x, ldx, x2, ldx2, csplit, wq, bias, y, ldy, ws, stat_part, N, D, H, W, Cin, Cout, k
            print(f"stats   {ms:7.3f} ms  ldx={a[1]} x2={'y' if a[2] else 'n'} ldx2={a[3]} csplit={a[4]} ldy={a[8]} N={a[11]} DHW={a[12:15]} Cin={a[15]} Cout={a[16]} k={a[17]}")
    _lib.PROFILE = None


if __name__ == "__main__":
    main()
