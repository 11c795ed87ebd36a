// Hardware-layout probe for gfx950 (test tooling, not product code).
// Verifies, with exact small-integer data, the lane<->element maps this repo's kernels
// rely on: v_mfma_f32_16x16x32_bf16, v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x16_bf16
// and ds_read_b64_tr_b16.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// A[16][32] row-major bf16, B[32][16] row-major bf16 -> C[16][16]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x, r = l & 15, q = l >> 4;
  v8bf a, b;
  for (int j = 0; j < 8; j++) {
    unsigned short ua = A[r * 32 + 8 * q + j], ub = B[(8 * q + j) * 16 + r];
    a[j] = __builtin_bit_cast(__bf16, ua);
    b[j] = __builtin_bit_cast(__bf16, ub);
  }
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) C[(q * 4 + i) * 16 + r] = c[i];
}

// f32: A[16][32], B[32][16]; 8 x mfma 16x16x4 with element j of the same 8-wide fragment
__global__ void k_mfma16_f32(const float* A, const float* B, float* C) {
  int l = threadIdx.x, r = l & 15, q = l >> 4;
  v4f c = {0, 0, 0, 0};
  for (int j = 0; j < 8; j++) {
    float a = A[r * 32 + 8 * q + j], b = B[(8 * q + j) * 16 + r];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 4; i++) C[(q * 4 + i) * 16 + r] = c[i];
}

// 32x32x16: A[32][16], B[16][32] -> C[32][32]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  v8bf a, b;
  for (int j = 0; j < 8; j++) {
    a[j] = __builtin_bit_cast(__bf16, A[r * 16 + 8 * h + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(8 * h + j) * 32 + r]);
  }
  v16f c;
  for (int i = 0; i < 16; i++) c[i] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; i++) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = c[i];
  }
}

// tr read: LDS image [32 rows][16 cols] of 16-bit (row stride 32 B). value = row*100+col.
// lane 16g+4qq+p supplies address of row (rbase(g)+qq), cols 4p..4p+3.  Dump what each lane gets.
__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[32 * 16];
  for (int i = threadIdx.x; i < 32 * 16; i += 64) lds[i] = (short)((i / 16) * 100 + (i % 16));
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i16 = l & 15, qq = i16 >> 2, p = i16 & 3;
  int row = g * 8 + qq;  // group g covers rows 8g..8g+3 (first read) and 8g+4..8g+7 (second)
  v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) v4s*)(lds + row * 16 + 4 * p));
  v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) v4s*)(lds + (row + 4) * 16 + 4 * p));
  for (int j = 0; j < 4; j++) { out[l * 8 + j] = r0[j]; out[l * 8 + 4 + j] = r1[j]; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  srand(1);
  int bad = 0;
  {  // 16x16x32 bf16
    std::vector<unsigned short> A(16 * 32), B(32 * 16);
    std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 512; i++) { Af[i] = (float)(rand() % 7 - 3); A[i] = f2bf(Af[i]); Bf[i] = (float)(rand() % 5 - 2); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 32; k++) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int e = 0; for (int i = 0; i < 256; i++) e += (C[i] != R[i]);
    printf("mfma_16x16x32_bf16 mismatches: %d\n", e); bad += e;
    float *fA, *fB; CK(hipMalloc(&fA, 2048)); CK(hipMalloc(&fB, 2048));
    CK(hipMemcpy(fA, Af.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(fB, Bf.data(), 2048, hipMemcpyHostToDevice));
    k_mfma16_f32<<<1, 64>>>(fA, fB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    e = 0; for (int i = 0; i < 256; i++) e += (C[i] != R[i]);
    printf("mfma_16x16x4_f32 (8 steps) mismatches: %d\n", e); bad += e;
  }
  {  // 32x32x16
    std::vector<unsigned short> A(32 * 16), B(16 * 32);
    std::vector<float> Af(512), Bf(512), C(1024), R(1024, 0.f);
    for (int i = 0; i < 512; i++) { Af[i] = (float)(rand() % 7 - 3); A[i] = f2bf(Af[i]); Bf[i] = (float)(rand() % 5 - 2); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < 16; k++) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    int e = 0; for (int i = 0; i < 1024; i++) e += (C[i] != R[i]);
    printf("mfma_32x32x16_bf16 mismatches: %d\n", e); bad += e;
  }
  {  // tr read
    short* d; CK(hipMalloc(&d, 64 * 8 * 2));
    k_tr<<<1, 64>>>(d); CK(hipDeviceSynchronize());
    std::vector<short> o(512);
    CK(hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost));
    // expectation: lane 16g+i gets column i of rows 8g..8g+3 (r0) and rows 8g+4..8g+7 (r1)
    int e = 0;
    for (int l = 0; l < 64; l++) {
      int g = l >> 4, i = l & 15;
      for (int j = 0; j < 8; j++) e += (o[l * 8 + j] != (short)((8 * g + j) * 100 + i));
    }
    printf("ds_read_b64_tr_b16 mismatches vs expected map: %d\n", e); bad += e;
    if (e) for (int l = 0; l < 64; l++) { printf("lane %2d:", l); for (int j = 0; j < 8; j++) printf(" %5d", o[l * 8 + j]); printf("\n"); }
  }
  printf("PROBE %s\n", bad ? "FAIL" : "OK");
  return bad ? 1 : 0;
}
