// Shared device/host helpers for libdose_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dose_hip.h"

typedef unsigned short bf16_t;   // raw bf16 bits
typedef _Float16 f16_t;          // IEEE half (the third storage type: BASELINE.json configs[4] trains in fp16)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

void dp_set_error(const char* fmt, ...);
#define DP_FAIL(...) do { dp_set_error(__VA_ARGS__); return 1; } while (0)
#define DP_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    dp_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return 2; } } while (0)

// Dispatch on the storage dtype: CALL is a statement using the type alias T.
#define DP_DISPATCH(dtype, ...) do { if ((dtype) == DP_F32) { typedef float T; __VA_ARGS__; } \
    else if ((dtype) == DP_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else if ((dtype) == DP_F16) { typedef f16_t T; __VA_ARGS__; } \
    else { DP_FAIL("bad dtype %d", (int)(dtype)); } } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even, NaN preserved (plain cast -> v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ float ld_f(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void st_f(f16_t* p, float v) { *p = (f16_t)v; }
// v as it reads back after being stored as T (the normalisation statistics a convolution epilogue accumulates are those of the
// STORED tensor, so that mean / variance belong to the values the normalisation kernels read: ADVICE r2)
template <typename T> __device__ __forceinline__ float as_stored(float v);
template <> __device__ __forceinline__ float as_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float as_stored<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float as_stored<f16_t>(float v) { return (float)(f16_t)v; }

// 8 consecutive elements of T as a register fragment.
template <typename T> struct Frag8;
template <> struct Frag8<float> { float v[8]; };
template <> struct Frag8<bf16_t> { v4u u; };   // 8 x bf16 in 4 dwords
template <> struct Frag8<f16_t> { v4u u; };    // 8 x fp16 in 4 dwords (same register image; only the MFMA opcode and the converts differ)

template <typename T> __device__ __forceinline__ Frag8<T> frag_zero();
template <> __device__ __forceinline__ Frag8<float> frag_zero<float>() { Frag8<float> f; for (int i = 0; i < 8; i++) f.v[i] = 0.f; return f; }
template <> __device__ __forceinline__ Frag8<bf16_t> frag_zero<bf16_t>() { Frag8<bf16_t> f; f.u = (v4u){0, 0, 0, 0}; return f; }
template <> __device__ __forceinline__ Frag8<f16_t> frag_zero<f16_t>() { Frag8<f16_t> f; f.u = (v4u){0, 0, 0, 0}; return f; }

// Guarded load of up to 8 consecutive elements (nvalid in [0,8]); vector path when full and 16B/32B aligned.
__device__ __forceinline__ Frag8<float> frag_load(const float* p, int nvalid) {
  Frag8<float> f;
  if (nvalid >= 8 && ((uintptr_t)p & 15) == 0) {
    v4f a = *(const v4f*)p, b = *(const v4f*)(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3]; f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
  } else {
    for (int i = 0; i < 8; i++) f.v[i] = (i < nvalid) ? p[i] : 0.f;
  }
  return f;
}
__device__ __forceinline__ Frag8<bf16_t> frag_load(const bf16_t* p, int nvalid) {
  Frag8<bf16_t> f;
  if (nvalid >= 8 && ((uintptr_t)p & 15) == 0) {
    f.u = *(const v4u*)p;
  } else {
    unsigned w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; i++) if (i < nvalid) w[i >> 1] |= ((unsigned)p[i]) << ((i & 1) * 16);
    f.u = (v4u){w[0], w[1], w[2], w[3]};
  }
  return f;
}
__device__ __forceinline__ Frag8<f16_t> frag_load(const f16_t* p, int nvalid) {   // bit copy: identical to the bf16 path
  Frag8<bf16_t> t = frag_load((const bf16_t*)p, nvalid);
  Frag8<f16_t> f; f.u = t.u; return f;
}
// LDS store/load of a fragment (pointer must be 16B aligned for bf16, 16B for float halves)
__device__ __forceinline__ void frag_st_lds(float* p, const Frag8<float>& f) {
  *(v4f*)p = (v4f){f.v[0], f.v[1], f.v[2], f.v[3]};
  *(v4f*)(p + 4) = (v4f){f.v[4], f.v[5], f.v[6], f.v[7]};
}
__device__ __forceinline__ void frag_st_lds(bf16_t* p, const Frag8<bf16_t>& f) { *(v4u*)p = f.u; }
__device__ __forceinline__ void frag_st_lds(f16_t* p, const Frag8<f16_t>& f) { *(v4u*)p = f.u; }
__device__ __forceinline__ Frag8<float> frag_ld_lds(const float* p) {
  Frag8<float> f; v4f a = *(const v4f*)p, b = *(const v4f*)(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3]; f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
  return f;
}
__device__ __forceinline__ Frag8<bf16_t> frag_ld_lds(const bf16_t* p) { Frag8<bf16_t> f; f.u = *(const v4u*)p; return f; }
__device__ __forceinline__ Frag8<f16_t> frag_ld_lds(const f16_t* p) { Frag8<f16_t> f; f.u = *(const v4u*)p; return f; }

// One K=32 MFMA step on 16x16 tiles: acc += A(16x32) * B(32x16).  Lane l holds, for row/col (l&15), the 8
// k-values 8*(l>>4)+j.  bf16: one v_mfma_f32_16x16x32_bf16.  f32: 8 x v_mfma_f32_16x16x4_f32 (step j uses
// element j of every lane's fragment => k = 8q+j for q=0..3; exact fp32 FMA chain).  Verified by tools/probes/mfma_probe.hip.
__device__ __forceinline__ v4f mma16(const Frag8<bf16_t>& a, const Frag8<bf16_t>& b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a.u), __builtin_bit_cast(v8bf, b.u), c, 0, 0, 0);
}
__device__ __forceinline__ v4f mma16(const Frag8<f16_t>& a, const Frag8<f16_t>& b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a.u), __builtin_bit_cast(v8h, b.u), c, 0, 0, 0);
}
__device__ __forceinline__ v4f mma16(const Frag8<float>& a, const Frag8<float>& b, v4f c) {
#pragma unroll
  for (int j = 0; j < 8; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}

// two ds_read_b64_tr_b16 at a lane address and OFF elements further: 8 consecutive voxels (k) of one column
template <int OFF, typename T16>
__device__ __forceinline__ Frag8<T16> tr_pair(const T16* a) {
  static_assert(sizeof(T16) == 2, "transpose reads move 16-bit elements");
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + OFF));
  Frag8<T16> f;
  f.u[0] = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  f.u[1] = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  f.u[2] = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  f.u[3] = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return f;
}

// streaming weight gradient of the skinny row GEMMs (elementwise.hip), used by dp_conv3d_wgrad_tiled2 for k = 1
bool wgrad_rows_ok(int ldx, int ldgy, int64_t rows, int Cin, int Cout, int dtype);
int64_t wgrad_rows_ws_elems(int Cin, int Cout);
int wgrad_rows_launch(const void* x, int ldx, const void* gy, int ldgy, float* dw, int64_t s_co, int64_t s_ci, float* part, int64_t rows, int Cin, int Cout,
                      int dtype, int rezero, hipStream_t s);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence over ALL address spaces, which on
// gfx9 is s_waitcnt vmcnt(0): every global load still in flight (the software-prefetched next tile) is drained at each
// barrier.  Use this one when the barrier only publishes LDS data and the kernel has no global producer/consumer pair.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wave (64-lane) reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Mish for 16-bit storage: hardware exp2 / rcp approximations (1 ulp-class fp32 error, three orders below the bf16 rounding of
// the result) instead of the correctly rounded expf and IEEE divisions, which made the Mish row streams math-bound
// (3.1 TB/s against 4.6 TB/s for ReLU).  fp32 storage (parity mode) keeps the exact forms below.
__device__ __forceinline__ float mish_fwd_fast(float z) {
  if (z > 20.f) return z;
  const float n = __builtin_amdgcn_exp2f(z * 1.4426950408889634f), t = n * (n + 2.f);
  return z * t * __builtin_amdgcn_rcpf(t + 2.f);
}
__device__ __forceinline__ float mish_bwd_fast(float z) {
  if (z > 20.f) return 1.f;
  const float n = __builtin_amdgcn_exp2f(z * 1.4426950408889634f), t = n * (n + 2.f), th = t * __builtin_amdgcn_rcpf(t + 2.f);
  return th + z * (1.f - th * th) * n * __builtin_amdgcn_rcpf(1.f + n);
}
// activations (forward value and derivative w.r.t. the pre-activation z)
__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case DP_ACT_RELU: return z > 0.f ? z : 0.f;
    case DP_ACT_LRELU: return z >= 0.f ? z : 0.01f * z;
    case DP_ACT_MISH: {   // z*tanh(softplus(z)) with n = e^z: tanh(log(1+n)) = t/(t+2), t = n(n+2)  (one exp, one divide)
      if (z > 20.f) return z;
      float n = expf(z), t = n * (n + 2.f);
      return z * (t / (t + 2.f));
    }
    case DP_ACT_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
    default: return z;
  }
}
__device__ __forceinline__ float act_bwd(float z, int act) {
  switch (act) {
    case DP_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case DP_ACT_LRELU: return z >= 0.f ? 1.f : 0.01f;
    case DP_ACT_MISH: {   // d/dz [z tanh(sp)] = th + z (1 - th^2) sigmoid(z), th = t/(t+2), sigmoid = n/(1+n)
      if (z > 20.f) return 1.f;
      float n = expf(z), t = n * (n + 2.f), th = t / (t + 2.f);
      return th + z * (1.f - th * th) * (n / (1.f + n));
    }
    case DP_ACT_GELU: return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
    default: return 1.f;
  }
}

// conv_wgrad_hk.hip: weight gradient of the <= 16-output-channel 7x7x7 layers with K along H (shares the tap-major scratch and the
// unpack kernel of dp_conv3d_wgrad_tiled2, which dispatches to it)
struct WgHkGeom {
  int N, D, H, W, Cin, Cout, ldx, ldgy;
  int tiles_h, tiles_w, MT, NTn, ydim;
  const void* x2; int ldx2, csplit;   // virtual concat of the input
  int dbg;                            // experiments only (env DP_DBG): 1 = no staging, 2 = no sweep, 4 = no epilogue
  float* dw; int64_t s_co, s_ci, s_tap; int rezero;     // destination of the finish pass (3^3: per-block slabs instead of atomics)
  int64_t slab;                       // 7^3, deterministic mode: voxel share yb accumulates into scratch slab yb (slab elements apart); 0 = one shared scratch
  int max_slabs; int* nslab_out;      // ... at most max_slabs shares; *nslab_out = the number used (what the unpack pass has to add up)
};
bool wgrad_hk_applicable(int Cout, int k, int H, int W, int dtype);
int64_t wgrad_hk_ws_elems(int Cin, int Cout, int k);     // fp32 scratch elements the K-along-H kernels may use (0: no extra need)
// *finished = 1: dW has been written (no unpack of the tap-major scratch needed)
int wgrad_hk_launch(const void* x, const void* gy, float* ws, const WgHkGeom& g, int k, int dtype, hipStream_t s, int* finished);

// conv_cc16.hip: 16x16x32-MFMA convolution for the Cout <= 16 layers at W >= 96 (chosen by shape alone, so that the packed-weight
// layout is known from (Cin, Cout, k, W))
bool cc16_applicable(int Cin, int Cout, int k, int W);
int cc16_weight_elems(int Cin, int Cout, int k);
int cc16_pack(const float* w, void* dst, int Cout, int Cin, int k, int tf, int dtype, hipStream_t s);
int cc16_stat_blocks(int D, int H, int W, int k, int dtype);
bool cc16_wide(const void* y, int ldy, const void* y2, int ldy2, int osplit, int dtype);
int cc16_launch(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias, void* y, int ldy,
                void* y2, int ldy2, int osplit, float* stat_part, int N, int D, int H, int W, int Cin, int Cout, int k, int dtype, hipStream_t s);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// dp_set_deterministic (elementwise.hip): 1 = every reduction that normally meets in fp32 atomics takes a fixed-order path instead, so that
// two runs on the same inputs are bit-identical (split-kd convolutions run unsplit, split-K GEMMs unsplit, the weight-gradient kernels write one
// scratch slab per voxel share and the unpack pass adds the slabs in order, LayerNorm's dgamma / dbeta go through per-block partial rows).
int dp_det(int site = 0x7fffffff);        // site: DET_* bit(s); experiments (tools/probes/determinism_probe.py) switch single sites on through dp_set_deterministic(mask)
enum { DET_SPLITKD = 1, DET_SPLITK = 2, DET_WGRAD = 4, DET_WGRAD_GENERIC = 8, DET_TRILINEAR = 16 };
// slabs of `base` fp32 elements the tap-major weight-gradient scratch holds in deterministic mode (<= 32 Mi elements in total)
static inline int det_slabs(int64_t base) { int64_t n = (32ll << 20) / (base > 0 ? base : 1); return n < 1 ? 1 : (n > 512 ? 512 : (int)n); }
static inline int roundup8(int c) { return (c + 7) & ~7; }
