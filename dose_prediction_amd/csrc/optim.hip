// Fused multi-tensor Adam (AMSGrad, L2 weight decay) -- SURVEY.md section 8f "next-4": the optimizer NetworkTrainer builds
// (network_trainer.py:120-125: optim.Adam(lr, weight_decay=3e-5, betas=(0.9,0.999), eps=1e-8, amsgrad=True)) as ONE launch
// over all parameter tensors.  Pure HBM stream: 5 reads + 4 writes of fp32 per parameter (36 B), float4 accesses.
#include "common.h"

#define STREAM ((hipStream_t)stream)
#define ADAM_CHUNK 8192          // elements per block

// cdst / ckind: a packed copy of the parameter that the update writes on the way (the parameter is in registers anyway; a separate
// dp_pack_multi pass re-read 4 bytes per element for it): 0 none; 1 / 2 plain cast to bf16 / fp16 (the kernel layout of a Linear
// weight whose row length is a multiple of 8); 3 the fp32x3 Linear operand bf16 [rows][3 cp] over rows of K elements, block b holding
// hi or lo per the pattern bits (cpp = cp | pattern << 32).
struct AdamTensor { float* p; const float* g; float* m; float* v; float* vmax; int64_t n; void* cdst; int64_t ckind, K, cpp; };

__device__ __forceinline__ void adam_write_copy(const AdamTensor& T, int64_t i, const float* v4) {      // 4 consecutive new parameter values at index i
  if (T.ckind == 1) {
    const unsigned a = (unsigned)f2bf(v4[0]) | ((unsigned)f2bf(v4[1]) << 16), b = (unsigned)f2bf(v4[2]) | ((unsigned)f2bf(v4[3]) << 16);
    *(uint2*)((bf16_t*)T.cdst + i) = make_uint2(a, b);
  } else if (T.ckind == 2) {
    f16_t* d = (f16_t*)T.cdst + i;
    d[0] = (f16_t)v4[0]; d[1] = (f16_t)v4[1]; d[2] = (f16_t)v4[2]; d[3] = (f16_t)v4[3];
  } else if (T.ckind == 3) {
    const unsigned K = (unsigned)T.K, cp = (unsigned)(T.cpp & 0xffffffff), pat = (unsigned)(T.cpp >> 32);
    const unsigned r = (unsigned)i / K, c = (unsigned)i - r * K;        // (n < 2^32; K % 4 == 0: the four elements share a row)
    unsigned hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bf16_t h0 = f2bf(v4[2 * q]), h1 = f2bf(v4[2 * q + 1]);
      const bf16_t l0 = f2bf(v4[2 * q] - bf2f(h0)), l1 = f2bf(v4[2 * q + 1] - bf2f(h1));
      hi[q] = (unsigned)h0 | ((unsigned)h1 << 16); lo[q] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    bf16_t* d = (bf16_t*)T.cdst + (int64_t)r * (3 * cp) + c;
#pragma unroll
    for (int b = 0; b < 3; b++) *(uint2*)(d + b * cp) = ((pat >> b) & 1) ? make_uint2(lo[0], lo[1]) : make_uint2(hi[0], hi[1]);
  }
}

__device__ __forceinline__ void adam_write_copy1(const AdamTensor& T, int64_t i, float v) {             // (unaligned tensors: element by element)
  if (T.ckind == 1) ((bf16_t*)T.cdst)[i] = f2bf(v);
  else if (T.ckind == 2) ((f16_t*)T.cdst)[i] = (f16_t)v;
  else if (T.ckind == 3) {
    const unsigned K = (unsigned)T.K, cp = (unsigned)(T.cpp & 0xffffffff), pat = (unsigned)(T.cpp >> 32);
    const unsigned r = (unsigned)i / K, c = (unsigned)i - r * K;
    const bf16_t h = f2bf(v), l = f2bf(v - bf2f(h));
    bf16_t* d = (bf16_t*)T.cdst + (int64_t)r * (3 * cp) + c;
    for (int b = 0; b < 3; b++) d[b * cp] = ((pat >> b) & 1) ? l : h;
  }
}

__global__ void __launch_bounds__(256) k_adam_multi(const AdamTensor* __restrict__ tab, const int* __restrict__ chunk_t, const int* __restrict__ chunk_i,
                                                    float lr_c1, float beta1, float beta2, float omb1, float omb2, float eps, float wd, float rsqrt_c2, int amsgrad,
                                                    const int* __restrict__ step_dev, double lr_d, double beta1_d, double beta2_d, float gscale,
                                                    int* __restrict__ found_inf) {
  const AdamTensor T = tab[chunk_t[blockIdx.x]];
  if (step_dev) {   // capturable mode: the step count lives on the device (a HIP graph replays the launch with fresh bias corrections)
    const double st = (double)*step_dev;
    lr_c1 = (float)(lr_d / (1.0 - pow(beta1_d, st)));
    rsqrt_c2 = (float)(1.0 / sqrt(1.0 - pow(beta2_d, st)));
  }
  const int64_t base = (int64_t)chunk_i[blockIdx.x] * ADAM_CHUNK;
  const int64_t end = min(T.n, base + ADAM_CHUNK);
  const bool vec = (((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v | (uintptr_t)T.vmax) & 15) == 0;
  bool bad = false;
  auto upd = [&](float& p, float g, float& m, float& v, float& vm) {
    // a non-finite gradient (fp16 overflow under a static loss scale) must not reach the parameter or the moments: one NaN in
    // max_exp_avg_sq would stay there for ever.  The element is left untouched and *found_inf is raised.
    if (!(fabsf(g) <= 3.402823466e38f)) { bad = true; return; }
    g = g * gscale + wd * p;             // gscale = 1 / loss scale (fp16 training), 1 otherwise
    m = beta1 * m + omb1 * g;            // (1 - beta) is formed in double on the host, as torch does
    v = beta2 * v + omb2 * g * g;
    float vv = v;
    if (amsgrad) { vm = fmaxf(vm, v); vv = vm; }
    p -= lr_c1 * m / (sqrtf(vv) * rsqrt_c2 + eps);
  };
  if (vec) {
    // two float4 groups per thread per trip: 10 independent 16-byte loads in flight (one group left HBM at ~3 TB/s)
    for (int64_t i0 = base + threadIdx.x * 4; i0 < end; i0 += 2 * 256 * 4) {
      const int64_t i1 = i0 + 256 * 4;
      const bool f0 = i0 + 4 <= end, f1 = i1 + 4 <= end;
      v4f p0, g0, m0, v0, x0, p1, g1, m1, v1, x1;
      const v4f z4 = (v4f){0, 0, 0, 0};
      if (f0) { p0 = *(v4f*)(T.p + i0); g0 = *(const v4f*)(T.g + i0); m0 = *(v4f*)(T.m + i0); v0 = *(v4f*)(T.v + i0); x0 = amsgrad ? *(v4f*)(T.vmax + i0) : z4; }
      if (f1) { p1 = *(v4f*)(T.p + i1); g1 = *(const v4f*)(T.g + i1); m1 = *(v4f*)(T.m + i1); v1 = *(v4f*)(T.v + i1); x1 = amsgrad ? *(v4f*)(T.vmax + i1) : z4; }
      if (f0) {
#pragma unroll
        for (int k = 0; k < 4; k++) { float pp = p0[k], mm = m0[k], vv = v0[k], vx = x0[k]; upd(pp, g0[k], mm, vv, vx); p0[k] = pp; m0[k] = mm; v0[k] = vv; x0[k] = vx; }
        *(v4f*)(T.p + i0) = p0; *(v4f*)(T.m + i0) = m0; *(v4f*)(T.v + i0) = v0;
        if (amsgrad) *(v4f*)(T.vmax + i0) = x0;
        if (T.ckind) { const float t4[4] = {p0[0], p0[1], p0[2], p0[3]}; adam_write_copy(T, i0, t4); }
      } else {
        for (int64_t j = i0; j < end; j++) { float vx = amsgrad ? T.vmax[j] : 0.f; upd(T.p[j], T.g[j], T.m[j], T.v[j], vx); if (amsgrad) T.vmax[j] = vx; if (T.ckind) adam_write_copy1(T, j, T.p[j]); }
      }
      if (f1) {
#pragma unroll
        for (int k = 0; k < 4; k++) { float pp = p1[k], mm = m1[k], vv = v1[k], vx = x1[k]; upd(pp, g1[k], mm, vv, vx); p1[k] = pp; m1[k] = mm; v1[k] = vv; x1[k] = vx; }
        *(v4f*)(T.p + i1) = p1; *(v4f*)(T.m + i1) = m1; *(v4f*)(T.v + i1) = v1;
        if (amsgrad) *(v4f*)(T.vmax + i1) = x1;
        if (T.ckind) { const float t4[4] = {p1[0], p1[1], p1[2], p1[3]}; adam_write_copy(T, i1, t4); }
      } else {
        for (int64_t j = i1; j < end; j++) { float vx = amsgrad ? T.vmax[j] : 0.f; upd(T.p[j], T.g[j], T.m[j], T.v[j], vx); if (amsgrad) T.vmax[j] = vx; if (T.ckind) adam_write_copy1(T, j, T.p[j]); }
      }
    }
  } else {
    for (int64_t j = base + threadIdx.x; j < end; j += 256) { float vx = amsgrad ? T.vmax[j] : 0.f; upd(T.p[j], T.g[j], T.m[j], T.v[j], vx); if (amsgrad) T.vmax[j] = vx; if (T.ckind) adam_write_copy1(T, j, T.p[j]); }
  }
  if (bad && found_inf) *found_inf = 1;
}

// table: device array of {p, g, m, v, vmax, n, cdst, ckind, K, cpp} (10 x 8 bytes per tensor; cdst .. cpp: see AdamTensor); chunk_t / chunk_i: device int arrays mapping each
// block to (tensor, chunk index of ADAM_CHUNK elements).  step >= 1 is the step count AFTER this update.
extern "C" int dp_adam_chunk(void) { return ADAM_CHUNK; }
extern "C" int dp_adam_multi(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, double lr, double beta1, double beta2,
                             double eps, double weight_decay, double inv_grad_scale, int step, int amsgrad, int32_t* found_inf, void* stream) {
  if (nchunks <= 0) return 0;
  if (step < 1) DP_FAIL("adam: step must be >= 1");
  double c1 = 1.0 - pow(beta1, (double)step), c2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(k_adam_multi, dim3(nchunks), dim3(256), 0, STREAM, (const AdamTensor*)table, (const int*)chunk_t, (const int*)chunk_i,
                     (float)(lr / c1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(1.0 / sqrt(c2)), amsgrad,
                     (const int*)nullptr, lr, beta1, beta2, (float)inv_grad_scale, (int*)found_inf);
  DP_CHECK_LAUNCH("adam_multi"); return 0;
}
// Capturable variant: *step_dev (device int32) is incremented first and the bias corrections are formed from it on the device, so
// the launch pair can be replayed from a captured HIP graph and still be Adam (a host-side step count would be frozen at capture).
__global__ void k_adam_tick(int* step) { *step += 1; }
extern "C" int dp_adam_multi_dev(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, double inv_grad_scale, int32_t* step_dev, int amsgrad, int32_t* found_inf, void* stream) {
  if (nchunks <= 0) return 0;
  if (!step_dev) DP_FAIL("adam_multi_dev: step_dev is NULL");
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, STREAM, (int*)step_dev);
  hipLaunchKernelGGL(k_adam_multi, dim3(nchunks), dim3(256), 0, STREAM, (const AdamTensor*)table, (const int*)chunk_t, (const int*)chunk_i,
                     0.f, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, 0.f, amsgrad,
                     (const int*)step_dev, lr, beta1, beta2, (float)inv_grad_scale, (int*)found_inf);
  DP_CHECK_LAUNCH("adam_multi_dev"); return 0;
}
