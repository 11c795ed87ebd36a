#!/usr/bin/env python3
"""Would the N = 768 token GEMMs with deep K gain from split-K into fp32 slabs (batched dp_gemm_nt, 128 x 128 tiles) + one reduce pass?
Times the batched partial GEMM alone (the reduce pass would add ~5 us + a launch gap) against the unsplit launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dose_prediction_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for M, N, K in ((1024, 768, 3072), (1024, 768, 2304), (1024, 768, 9216), (1024, 768, 768)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    base = timeit(lambda: ops.gemm_nt(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N))
    line = f"{M}x{N}x{K}: unsplit {base:6.1f} us |"
    for S in (2, 3, 4, 6, 8, 12, 16):
        ks = K // S
        if ks * S != K or ks % 128:
            continue
        parts = torch.empty(S, M, N, device=dev, dtype=torch.float32)
        t = timeit(lambda: ops.gemm_nt(A, B, parts, M=M, N=N, K=ks, batch=(S, 1), sa=(ks, 0), sb=(ks, 0), sc=(M * N, 0), lda=K, ldb=K, ldc=N))
        red = timeit(lambda: parts.sum(0))
        line += f" S={S}: {t:5.1f} (+torch sum {red:4.1f}) |"
    print(line, flush=True)
