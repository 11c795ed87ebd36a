import sys, torch
sys.path.insert(0, '/root/repo')
from dose_prediction_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
S = 128
a = torch.randn((2, S, S, S, 16), device=dev).to(dt)
bfull = torch.zeros((2, S, S, S, 16), device=dev, dtype=dt); bfull[..., :9] = torch.randn((2, S, S, S, 9), device=dev).to(dt)
for cb in (16, 9):
    b = bfull[..., :cb]
    w = (torch.randn((16, 16 + cb, 3, 3, 3), device=dev) * 0.01)
    with torch.no_grad():
        t = timeit(lambda: ops.conv3d((a, b), w, None, 1, 1, 1))
        t2 = timeit(lambda: ops.conv3d((a, b), w, None, 1, 1, 1, stats=True))
    print(f"virtual concat 16+{cb} -> 16 k3: {t:.3f} ms, with stats {t2:.3f} ms   b.stride={b.stride()}")
