import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from dose_prediction_amd import ops
dev = torch.device("cuda:0")
N, Cin, Cout, D, H, W, k = 4, 32, 64, 100, 16, 16, 3
g = torch.Generator().manual_seed(1)
x = torch.randn((N, Cin, D, H, W), generator=g)
w = torch.randn((Cout, Cin, k, k, k), generator=g) * (Cin * 27) ** -0.5
r = torch.randn((N, Cout, D, H, W), generator=g)
nd = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
nc = lambda t: t.permute(0, 4, 1, 2, 3).contiguous()
def rep(name, got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    e = (got - ref).abs()
    i = int(e.argmax())
    idx = []
    for s in reversed(ref.shape):
        idx.append(i % s); i //= s
    print(f"{name:28s} rel_max {float(e.max() / ref.abs().max()):.3e} rel_l2 {float((got - ref).norm() / ref.norm()):.3e} at {tuple(reversed(idx))}  n_bad {(e > 1e-3 * ref.abs().max()).sum().item()}", flush=True)
# 1. conv fwd / dgrad / wgrad alone
xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
yr = oracle.conv3d(xr, wr, None, 1, 1, 1)
yr.backward(r.double())
xd = nd(x).to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
y = ops.conv3d(xd, wd, None, 1, 1, 1)
y.backward(nd(r).to(dev))
rep("conv fwd", nc(y), yr); rep("conv dgrad", nc(xd.grad), xr.grad); rep("conv wgrad", wd.grad, wr.grad)
# 2. with stats
xd2 = nd(x).to(dev).requires_grad_(True)
y2, st = ops.conv3d(xd2, wd, None, 1, 1, 1, stats=True)
rep("conv fwd (stats variant)", nc(y2), yr)
# 3. norm alone on the oracle's y
yin = nd(yr.detach().float()).to(dev).requires_grad_(True)
z = ops.norm_act(yin, "instance", act="relu")
yr2 = yr.detach().clone().requires_grad_(True)
zr = oracle.activation(oracle.instance_norm(yr2), "relu")
zr.backward(r.double())
z.backward(nd(r).to(dev))
rep("norm fwd", nc(z), zr); rep("norm bwd", nc(yin.grad), yr2.grad)
# 4. dgrad of the norm-backward gradient
gy = nd(yr2.grad.float()).to(dev)
xd3 = nd(x).to(dev).requires_grad_(True)
y3 = ops.conv3d(xd3, wd, None, 1, 1, 1)
y3.backward(gy)
xr3 = x.double().requires_grad_(True)
y3r = oracle.conv3d(xr3, w.double(), None, 1, 1, 1)
y3r.backward(yr2.grad)
rep("dgrad of norm-bwd grad", nc(xd3.grad), xr3.grad)
