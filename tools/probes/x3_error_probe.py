"""Error of the fp32x3 mode next to the exact-fp32 mode on the golden networks and on single convolutions (vs float64)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from helpers import load_golden, sub, rel_err, rel_l2, cmp_prefix
import dose_prediction_amd
from dose_prediction_amd import ops
dev = torch.device("cuda:0")

def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale

for mode in ("fp32", "fp32x3"):
    dose_prediction_amd.set_compute_dtype(mode)
    for (N, Cin, Cout, D, H, W, k) in ((2, 16, 16, 6, 10, 20, 3), (2, 32, 16, 3, 9, 130, 7), (1, 64, 64, 6, 6, 16, 3)):
        x = rnd((N, Cin, D, H, W), 1) * 1.3 + 0.2
        w = rnd((Cout, Cin, k, k, k), 2, (Cin * k ** 3) ** -0.5)
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yr = oracle.conv3d(xr, wr, None, 1, k // 2, 1)
        r = rnd(yr.shape, 4)
        (yr * r.double()).sum().backward()
        xh = x.permute(0, 2, 3, 4, 1).contiguous().to(dev).requires_grad_(True)
        wh = w.to(dev).requires_grad_(True)
        yh = ops.conv3d(xh, wh, None, 1, k // 2, 1)
        yh.backward(r.permute(0, 2, 3, 4, 1).contiguous().to(dev))
        print(mode, (Cin, Cout, k), "y %.2e gx %.2e gw %.2e" % (rel_l2(yh.permute(0, 4, 1, 2, 3).cpu(), yr.detach()),
              rel_l2(xh.grad.permute(0, 4, 1, 2, 3).cpu(), xr.grad), rel_l2(wh.grad.cpu(), wr.grad)))
    from dose_prediction_amd.models.c3d import BaseUNet
    g = load_golden("g1_base_unet")
    net = BaseUNet(3, [-1, 4, 8, 8, 16, 16])
    net.load_state_dict(sub(g, "sd"))
    net = net.to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    y = net(x)
    y.backward(g["r"].to(dev))
    print(mode, "G1 y rel_err %.2e  gx %.2e" % (rel_err(y.cpu(), g["y"]), cmp_prefix(x.grad.cpu(), g["gx"])))
