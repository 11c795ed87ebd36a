"""Time the attention core (forward + backward) of one ViT layer: fused dp_attention_* against the GEMM + softmax path.
usage: python tools/bench_attn.py [B N heads d]..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dose_prediction_amd import ops  # noqa: E402


def run(fn, qkv, go, iters=50):
    for _ in range(5):
        x = qkv.detach().requires_grad_(True)
        fn(x).backward(go)
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(iters):
        x = qkv.detach().requires_grad_(True)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = fn(x)
        e[1].record()
        y.backward(go)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    return 1e3 * tf / iters, 1e3 * tb / iters


def main():
    cfgs = [(2, 512, 6, 128), (2, 512, 12, 64), (1, 1152, 6, 128), (1, 1152, 12, 64)]
    dev = torch.device("cuda:0")
    for dt in (torch.bfloat16, torch.float16):
        for B, N, heads, d in cfgs:
            H = heads * d
            qkv = torch.randn(B, N, 3 * H, device=dev).to(dt)
            go = torch.randn(B, N, H, device=dev).to(dt)
            f = run(lambda x: ops.FusedAttention.apply(x, heads), qkv, go)
            u = run(lambda x: ops.Attention.apply(x, heads), qkv, go)
            flops = 4.0 * B * heads * N * N * d
            print(f"{str(dt):16s} B={B} N={N} heads={heads} d={d}: fused fwd {f[0]:7.1f} us bwd {f[1]:7.1f} us | "
                  f"gemm path fwd {u[0]:7.1f} us bwd {u[1]:7.1f} us | fused fwd {flops / f[0] / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
