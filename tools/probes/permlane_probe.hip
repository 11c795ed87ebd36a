// Prints what v_permlane16_swap_b32 does on gfx950 (used by the tap-pairing epilogue of conv_tiled.hip).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/permlane_probe tools/probes/permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  v2u r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("first operand x = lane, second y = 100 + lane\nr[0]:"); for (int i = 0; i < 64; i++) printf(" %u", h[i]);
  printf("\nr[1]:"); for (int i = 0; i < 64; i++) printf(" %u", h[64 + i]);
  printf("\n"); return 0;
}
