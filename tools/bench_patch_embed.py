"""Patch-embedding GEMM of DOSE-PYFER (1024 token rows x 768 x K = 102400): split-K sweep of dp_gemm_nt and the TN weight gradient."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dose_prediction_amd import ops
dev = torch.device("cuda:0")
M, N, K = 1024, 768, 102400
A = torch.randn((M, K), device=dev).to(torch.bfloat16)
B = torch.randn((N, K), device=dev).to(torch.bfloat16)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for sk in (4, 5, 6, 8, 10, 12, 16, 24):
    C = torch.zeros((M, N), device=dev, dtype=torch.float32)
    ms = timeit(lambda: ops.gemm_nt(A, B, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splitk=sk))
    print(f"splitk {sk:3d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)

# weight gradient of the same layer: dW[768][102400] = gy^T x (dp_gemm_tn).  A 128x128-tile variant measured 383 us against 364 us.
from dose_prediction_amd import _lib
gy = torch.randn((M, N), device=dev).to(torch.bfloat16)
dW = torch.empty((N, K), device=dev, dtype=torch.float32)
ms = timeit(lambda: _lib.call("dp_gemm_tn", gy.data_ptr(), N, A.data_ptr(), K, dW.data_ptr(), K, N, K, M, 1, 1, torch.cuda.current_stream().cuda_stream))
print(f"gemm_tn 768 x 102400 x 1024: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
