"""Bit-wise repeatability of single fp32x3 operators (one-product backward): conv3d / linear / attention run twice on the same inputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import ops  # noqa: E402

dose_prediction_amd.set_compute_dtype("fp32x3")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)


def rnd(*s):
    return torch.randn(s, generator=g)


def twice(name, f, inputs):
    res = []
    for _ in range(3):
        ins = [t.clone().to(dev).requires_grad_(True) for t in inputs]
        y = f(*ins)
        y.backward(torch.ones_like(y) * 0.37 + y.detach() * 0.1)
        torch.cuda.synchronize()
        res.append([y.detach().clone()] + [t.grad.clone() for t in ins])
    bad = [i for i in range(len(res[0])) if not (torch.equal(res[0][i], res[1][i]) and torch.equal(res[0][i], res[2][i]))]
    d = [float((res[0][i] - res[1][i]).norm() / res[0][i].norm()) for i in bad]
    print(f"{name:50s} differing tensors (0 = y, 1.. = input grads): {bad} {['%.1e' % v for v in d]}")


for (N, Cin, Cout, D, H, W, k) in [(1, 4, 4, 8, 8, 16, 3), (1, 8, 4, 8, 8, 16, 7), (1, 16, 16, 8, 8, 16, 3), (1, 4, 8, 32, 16, 16, 3), (1, 32, 16, 4, 8, 32, 7),
                                    (2, 16, 16, 4, 9, 130, 3), (1, 8, 8, 16, 8, 16, 3)]:
    x, w = rnd(N, D, H, W, Cin), rnd(Cout, Cin, k, k, k) * (Cin * k ** 3) ** -0.5
    twice(f"conv3d {Cin}->{Cout} k{k} {D}x{H}x{W}", lambda a, b: ops.conv3d(a, b, None, 1, k // 2, 1), [x, w])
for (rows, K, Nn) in [(256, 48, 144), (256, 48, 96), (256, 96, 48), (1024, 768, 768)]:
    twice(f"linear {K}->{Nn} rows {rows}", lambda a, b: ops.linear(a, b, None), [rnd(2, rows // 2, K), rnd(Nn, K) * K ** -0.5])
twice("attention 6 heads d 8", lambda a: ops.attention(a, 6), [rnd(2, 64, 3 * 48)])
twice("attention 6 heads d 128", lambda a: ops.attention(a, 6), [rnd(2, 512, 3 * 768)])
