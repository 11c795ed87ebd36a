// In-kernel timeline of dp_gemm_nt (s_memtime ticks of one block): where does a token GEMM spend its microseconds?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -DDP_GEMM_PROBE -o gemm_probe tools/probes/gemm_probe.hip ; run on the GPU box.
#include "../dose_prediction_amd/csrc/gemm.hip"
#include <stdarg.h>
#include <vector>
void dp_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main() {
  const int shapes[][3] = {{1024, 768, 768}, {1024, 768, 3072}, {256, 768, 768}, {1024, 3072, 768}};
  for (auto& sh : shapes) {
    int M = sh[0], N = sh[1], K = sh[2];
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; it++) dp_gemm_nt(A, K, 0, 0, B, K, 0, 0, C, N, 0, 0, nullptr, M, N, K, 1, 1, 1.f, 0, 1, DP_BF16, nullptr);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; it++) dp_gemm_nt(A, K, 0, 0, B, K, 0, 0, C, N, 0, 0, nullptr, M, N, K, 1, 1, 1.f, 0, 1, DP_BF16, nullptr);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long p[40]; hipMemcpyFromSymbol(p, HIP_SYMBOL(dp_gemm_probe), sizeof(p));
    int iters = (K + 127) / 128; if (iters > 29) iters = 29;
    printf("M=%d N=%d K=%d: %.1f us per launch (HIP events, back to back). ticks (100 MHz s_memtime = 10 ns): start->loop0 %llu", M, N, K, ms * 50.f, p[1] - p[0]);
    printf(" | per K step:"); for (int i = 1; i < iters; i++) printf(" %llu", p[i + 1] - p[i]);
    printf(" | step 2 phases: wait+barrier %llu, LDS stores+barrier %llu, issue next loads %llu, reads+MFMA %llu", p[33] - p[3], p[34] - p[33], p[35] - p[34], p[4] - p[35]);
    printf(" | last step->loop end %llu | epilogue %llu | block total %llu\n", p[31] - p[iters], p[32] - p[31], p[32] - p[0]);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
