import sys, torch
sys.path.insert(0, '.')
from dose_prediction_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for (S, c, co, k) in [(128, 16, 16, 7), (128, 16, 16, 3), (64, 32, 32, 7)]:
    a = torch.randn((2, S, S, S, c), device=dev).to(dt); b = torch.randn((2, S, S, S, c), device=dev).to(dt)
    ab = torch.cat((a, b), -1).contiguous()
    w = (torch.randn((co, 2 * c, k, k, k), device=dev) * 0.01)
    fl = 2.0 * 2 * S ** 3 * 2 * c * co * k ** 3
    with torch.no_grad():
        t1 = timeit(lambda: ops.conv3d(ab, w, None, 1, k // 2, 1))
        t2 = timeit(lambda: ops.conv3d((a, b), w, None, 1, k // 2, 1))
    print(f"S={S} {2*c}->{co} k{k}: physical cat {t1:.3f} ms ({fl/t1/1e9:.0f} TF)   virtual {t2:.3f} ms ({fl/t2/1e9:.0f} TF)")
