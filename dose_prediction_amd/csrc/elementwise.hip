// Data-movement and elementwise kernels (HBM-bound; 16-byte vector accesses wherever rows allow).
#include "common.h"
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void dp_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* dp_last_error(void) { return g_err; }
extern "C" int dp_version(void) { return 100; }
// Deterministic mode (process-wide; see common.h dp_det()).
static int g_deterministic = 0;
int dp_det(int site) { return (g_deterministic & site) != 0; }
extern "C" int dp_set_deterministic(int on) { g_deterministic = on == 1 ? 0x7fffffff : on; return 0; }     // (1 = everything; other values: a mask of DET_* sites, experiments only)
extern "C" int dp_get_deterministic(void) { return g_deterministic; }

#define STREAM ((hipStream_t)stream)
static inline int grid_for(int64_t work, int block, int cap = 256 * 16) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

template <typename T> __device__ __forceinline__ void frag_store(T* p, const Frag8<T>& f, int nvalid);
template <> __device__ __forceinline__ void frag_store<float>(float* p, const Frag8<float>& f, int nvalid) {
  if (nvalid >= 8 && ((uintptr_t)p & 15) == 0) {
    *(v4f*)p = (v4f){f.v[0], f.v[1], f.v[2], f.v[3]}; *(v4f*)(p + 4) = (v4f){f.v[4], f.v[5], f.v[6], f.v[7]};
  } else { for (int i = 0; i < 8; i++) if (i < nvalid) p[i] = f.v[i]; }
}
template <> __device__ __forceinline__ void frag_store<bf16_t>(bf16_t* p, const Frag8<bf16_t>& f, int nvalid) {
  if (nvalid >= 8 && ((uintptr_t)p & 15) == 0) { *(v4u*)p = f.u; }
  else { for (int i = 0; i < 8; i++) if (i < nvalid) p[i] = (bf16_t)((f.u[i >> 1] >> ((i & 1) * 16)) & 0xffff); }
}
template <> __device__ __forceinline__ void frag_store<f16_t>(f16_t* p, const Frag8<f16_t>& f, int nvalid) {
  Frag8<bf16_t> t; t.u = f.u; frag_store<bf16_t>((bf16_t*)p, t, nvalid);        // bit copy
}
__device__ __forceinline__ void frag_unpack(const Frag8<float>& f, float* o) { for (int i = 0; i < 8; i++) o[i] = f.v[i]; }
__device__ __forceinline__ void frag_unpack(const Frag8<f16_t>& f, float* o) {
  const v8h h = __builtin_bit_cast(v8h, f.u);
  for (int i = 0; i < 8; i++) o[i] = (float)h[i];
}
__device__ __forceinline__ void frag_pack(Frag8<f16_t>& f, const float* o) {
  v8h h; for (int i = 0; i < 8; i++) h[i] = (f16_t)o[i];
  f.u = __builtin_bit_cast(v4u, h);
}
__device__ __forceinline__ void frag_unpack(const Frag8<bf16_t>& f, float* o) {
  for (int i = 0; i < 4; i++) { o[2 * i] = __uint_as_float(f.u[i] << 16); o[2 * i + 1] = __uint_as_float(f.u[i] & 0xffff0000u); }
}
__device__ __forceinline__ void frag_pack(Frag8<float>& f, const float* o) { for (int i = 0; i < 8; i++) f.v[i] = o[i]; }
__device__ __forceinline__ void frag_pack(Frag8<bf16_t>& f, const float* o) {
  for (int i = 0; i < 4; i++) f.u[i] = (unsigned)f2bf(o[2 * i]) | ((unsigned)f2bf(o[2 * i + 1]) << 16);
}

// ------------------------------------------------------------------------------------------------ layout
// Index arithmetic of the row kernels is done in 32 bits whenever the work fits (it always does for the volumes of this model):
// a 64-bit divide per thread costs more than the 16 bytes the thread moves.
template <typename T, typename I>
__device__ __forceinline__ void ncdhw_to_ndhwc_body(const float* __restrict__ src, T* __restrict__ dst, I total, I V, int C, int ld, int cpad) {
  bool vec = false;
  if constexpr (sizeof(T) == 2) vec = (cpad & 7) == 0 && (ld & 7) == 0 && (((uintptr_t)dst) & 15) == 0;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I n = i / V, v = i - n * V;
    T* d = dst + (int64_t)i * ld;
    const float* s = src + (int64_t)n * C * V + v;
    if constexpr (sizeof(T) == 2) {
      if (vec) {      // 8 channels -> one 16-byte store (2-byte scattered stores ran this conversion at 2.7 TB/s)
        for (int c0 = 0; c0 < cpad; c0 += 8) {
          float t[8];
#pragma unroll
          for (int j = 0; j < 8; j++) t[j] = (c0 + j < C) ? s[(int64_t)(c0 + j) * V] : 0.f;
          Frag8<T> f; frag_pack(f, t);
          *(v4u*)(d + c0) = f.u;
        }
        continue;
      }
    }
    if constexpr (sizeof(T) == 4) {
      // fp32 rows (parity / fp32x3 modes): four channels -> one 16-byte store (round 5: the 4-byte stores ran the 9 -> 16 channel entry
      // conversion of a 2 x 128^3 input at 1 TB/s, 0.45 ms where the bf16 path takes 0.05)
      if ((cpad & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
        for (int c0 = 0; c0 < cpad; c0 += 4) {
          v4f t;
#pragma unroll
          for (int j = 0; j < 4; j++) t[j] = (c0 + j < C) ? s[(int64_t)(c0 + j) * V] : 0.f;
          *(v4f*)((float*)d + c0) = t;
        }
        continue;
      }
    }
    for (int c = 0; c < cpad; c++) st_f(d + c, c < C ? s[(int64_t)c * V] : 0.f);
  }
}
template <typename T>
__global__ void k_ncdhw_to_ndhwc(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int64_t V, int ld, int cpad) {
  const int64_t total = (int64_t)N * V;
  if (total < (1ll << 31)) ncdhw_to_ndhwc_body<T, unsigned>(src, dst, (unsigned)total, (unsigned)V, C, ld, cpad);
  else ncdhw_to_ndhwc_body<T, int64_t>(src, dst, total, V, C, ld, cpad);
}
template <typename T, typename I>
__device__ __forceinline__ void ndhwc_to_ncdhw_body(const T* __restrict__ src, float* __restrict__ dst, I total, I V, int C, int ld, int acc) {
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I n = i / V, v = i - n * V;
    const T* s = src + (int64_t)i * ld;
    float* d = dst + (int64_t)n * C * V + v;
    for (int c = 0; c < C; c++) { float x = ld_f(s + c); if (acc) d[(int64_t)c * V] += x; else d[(int64_t)c * V] = x; }
  }
}
template <typename T>
__global__ void k_ndhwc_to_ncdhw(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int64_t V, int ld, int acc) {
  const int64_t total = (int64_t)N * V;
  if (total < (1ll << 31)) ndhwc_to_ncdhw_body<T, unsigned>(src, dst, (unsigned)total, (unsigned)V, C, ld, acc);
  else ndhwc_to_ncdhw_body<T, int64_t>(src, dst, total, V, C, ld, acc);
}
extern "C" int dp_ncdhw_to_ndhwc(const float* src, void* dst, int N, int C, int64_t V, int ld, int cpad, int dtype, void* stream) {
  if (cpad < C || ld < cpad) DP_FAIL("ncdhw_to_ndhwc: need C <= cpad <= ld");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_ncdhw_to_ndhwc<T>, dim3(grid_for(N * V, 256)), dim3(256), 0, STREAM, src, (T*)dst, N, C, V, ld, cpad));
  DP_CHECK_LAUNCH("ncdhw_to_ndhwc"); return 0;
}
extern "C" int dp_ndhwc_to_ncdhw(const void* src, float* dst, int N, int C, int64_t V, int ld, int accumulate, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_ndhwc_to_ncdhw<T>, dim3(grid_for(N * V, 256)), dim3(256), 0, STREAM, (const T*)src, dst, N, C, V, ld, accumulate));
  DP_CHECK_LAUNCH("ndhwc_to_ncdhw"); return 0;
}

template <typename T, typename I>
__device__ __forceinline__ void copy_rows_body(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, I total, I cg8, int C) {
  const I stride = (I)gridDim.x * blockDim.x;
  for (I i0 = blockIdx.x * (I)blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {   // four 16-byte loads in flight per thread
    Frag8<T> f[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const I i = i0 + u * stride;
      if (i < total) { const I r = i / cg8; const int cg = (int)(i - r * cg8); f[u] = frag_load(src + (int64_t)r * lds + cg * 8, min(8, C - cg * 8)); }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const I i = i0 + u * stride;
      if (i < total) { const I r = i / cg8; const int cg = (int)(i - r * cg8); frag_store<T>(dst + (int64_t)r * ldd + cg * 8, f[u], min(8, C - cg * 8)); }
    }
  }
}
template <typename T>
__global__ void k_copy_rows(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int64_t rows, int C) {
  const int cg8 = (C + 7) >> 3;
  const int64_t total = rows * cg8;
  if (total < (1ll << 31)) copy_rows_body<T, unsigned>(src, lds, dst, ldd, (unsigned)total, (unsigned)cg8, C);
  else copy_rows_body<T, int64_t>(src, lds, dst, ldd, total, (int64_t)cg8, C);
}
// torch.cat of two row tensors in one pass: every destination row is written whole (two dp_copy_rows calls wrote alternating
// 32-byte halves of each row and ran at 2.5 TB/s).
template <typename T>
__global__ void k_cat2_rows(const T* __restrict__ a, int lda, int Ca, const T* __restrict__ b, int ldb, int Cb, T* __restrict__ dst, int ldd,
                            unsigned total, unsigned cg8, unsigned cga) {
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
    Frag8<T> f[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned i = i0 + u * stride;
      if (i < total) {
        const unsigned r = i / cg8, cg = i - r * cg8;
        f[u] = cg < cga ? frag_load(a + (int64_t)r * lda + cg * 8, min(8, Ca - (int)cg * 8))
                        : frag_load(b + (int64_t)r * ldb + (cg - cga) * 8, min(8, Cb - (int)(cg - cga) * 8));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned i = i0 + u * stride;
      if (i < total) {
        const unsigned r = i / cg8, cg = i - r * cg8;
        const int nv = cg < cga ? min(8, Ca - (int)cg * 8) : min(8, Cb - (int)(cg - cga) * 8);
        frag_store<T>(dst + (int64_t)r * ldd + (cg < cga ? cg * 8 : Ca + (cg - cga) * 8), f[u], nv);
      }
    }
  }
}
extern "C" int dp_cat2_rows(const void* a, int lda, int Ca, const void* b, int ldb, int Cb, void* dst, int ldd, int64_t rows, int dtype, void* stream) {
  if ((Ca & 7) || Ca <= 0 || Cb <= 0 || ldd < Ca + Cb) DP_FAIL("cat2_rows: first width must be a positive multiple of 8 (got %d) and ldd >= Ca + Cb", Ca);
  const int cga = Ca / 8, cg8 = cga + (Cb + 7) / 8;
  const int64_t total = rows * cg8;
  if (total >= (1ll << 31)) DP_FAIL("cat2_rows: too many rows");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_cat2_rows<T>, dim3(grid_for(total, 256 * 4, 256 * 32)), dim3(256), 0, STREAM, (const T*)a, lda, Ca, (const T*)b,
                                        ldb, Cb, (T*)dst, ldd, (unsigned)total, (unsigned)cg8, (unsigned)cga));
  DP_CHECK_LAUNCH("cat2_rows"); return 0;
}
extern "C" int dp_copy_rows(const void* src, int lds, void* dst, int ldd, int64_t rows, int C, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_copy_rows<T>, dim3(grid_for(rows * ((C + 7) / 8), 256)), dim3(256), 0, STREAM, (const T*)src, lds, (T*)dst, ldd, rows, C));
  DP_CHECK_LAUNCH("copy_rows"); return 0;
}

template <typename S, typename D>
__global__ void k_cast(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) st_f(d + i, ld_f(s + i));
}
extern "C" int dp_cast(const void* src, int sdt, void* dst, int ddt, int64_t n, void* stream) {
  dim3 g(grid_for(n, 256)), b(256);
  if (sdt == DP_F32 && ddt == DP_BF16) hipLaunchKernelGGL((k_cast<float, bf16_t>), g, b, 0, STREAM, (const float*)src, (bf16_t*)dst, n);
  else if (sdt == DP_BF16 && ddt == DP_F32) hipLaunchKernelGGL((k_cast<bf16_t, float>), g, b, 0, STREAM, (const bf16_t*)src, (float*)dst, n);
  else if (sdt == DP_F32 && ddt == DP_F32) hipLaunchKernelGGL((k_cast<float, float>), g, b, 0, STREAM, (const float*)src, (float*)dst, n);
  else if (sdt == DP_BF16 && ddt == DP_BF16) hipLaunchKernelGGL((k_cast<bf16_t, bf16_t>), g, b, 0, STREAM, (const bf16_t*)src, (bf16_t*)dst, n);
  else if (sdt == DP_F32 && ddt == DP_F16) hipLaunchKernelGGL((k_cast<float, f16_t>), g, b, 0, STREAM, (const float*)src, (f16_t*)dst, n);
  else if (sdt == DP_F16 && ddt == DP_F32) hipLaunchKernelGGL((k_cast<f16_t, float>), g, b, 0, STREAM, (const f16_t*)src, (float*)dst, n);
  else if (sdt == DP_F16 && ddt == DP_F16) hipLaunchKernelGGL((k_cast<f16_t, f16_t>), g, b, 0, STREAM, (const f16_t*)src, (f16_t*)dst, n);
  else DP_FAIL("cast: bad dtypes");
  DP_CHECK_LAUNCH("cast"); return 0;
}
__global__ void k_fill(float* p, float v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int dp_fill_f32(float* p, float v, int64_t n, void* stream) {
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, p, v, n); DP_CHECK_LAUNCH("fill"); return 0;
}

// patchify: one thread per (b, token, p1, p2, p3): C contiguous elements on both sides.
template <typename T, bool INV, typename I>
__device__ __forceinline__ void patchify_body(const T* __restrict__ x, T* __restrict__ out, I total, int S0, int S1, int S2, int C, int ld, int p) {
  const I f1 = S1 / p, f2 = S2 / p, ntok = (I)(S0 / p) * f1 * f2, pu = p;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    I t = i; const I q1 = t / pu; const int p3 = (int)(t - q1 * pu); t = q1;
    const I q2 = t / pu; const int p2 = (int)(t - q2 * pu); t = q2;
    const I bt = t / pu; const int p1 = (int)(t - bt * pu);
    const I b = bt / ntok; I tok = bt - b * ntok;
    const I r1 = tok / f2; const int t2 = (int)(tok - r1 * f2);
    const I t0 = r1 / f1; const int t1 = (int)(r1 - t0 * f1);
    const int64_t vox = (((int64_t)b * S0 + t0 * pu + p1) * S1 + t1 * p + p2) * S2 + t2 * p + p3;
    const T* a = x + vox * ld; T* o = out + (int64_t)i * C;   // i == ((b*ntok+tok)*p^3 + pp)
    if (!INV) { for (int c = 0; c < C; c++) o[c] = a[c]; }
    else { T* aw = const_cast<T*>(a); for (int c = 0; c < C; c++) aw[c] = o[c]; }
  }
}
template <typename T, bool INV>
__global__ void k_patchify(const T* __restrict__ x, T* __restrict__ out, int B, int S0, int S1, int S2, int C, int ld, int p) {
  const int64_t total = (int64_t)B * S0 * S1 * S2;
  if (total < (1ll << 31)) patchify_body<T, INV, unsigned>(x, out, (unsigned)total, S0, S1, S2, C, ld, p);
  else patchify_body<T, INV, int64_t>(x, out, total, S0, S1, S2, C, ld, p);
}
// Fast path (16-bit storage): one wave per run of p voxels along W, i.e. per (b, token, p1, p2).  The run is contiguous on both
// sides (p*ld elements in x, p*C in the token row), so it is moved with 16-byte accesses through a wave-private LDS strip; the
// per-voxel kernel above issued 2*C two-byte accesses per voxel and ran at 1.5 TB/s.
template <typename T, bool INV>
__global__ void __launch_bounds__(256) k_patchify_runs(const T* __restrict__ x, T* __restrict__ out, int nruns, int S0, int S1, int S2, int C, int ld,
                                                       int p, unsigned cmul, int64_t ldo) {
  __shared__ __attribute__((aligned(16))) T strip[4][1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* ls = strip[wv];
  const int f1 = S1 / p, f2 = S2 / p, ntok = (S0 / p) * f1 * f2, nin = p * ld / 8, nout = p * C / 8, cg8 = ld / 8;
  for (int run = blockIdx.x * 4 + wv; run < nruns; run += gridDim.x * 4) {
    int t = run; const int p2 = t % p; t /= p; const int p1 = t % p; t /= p;
    const int tok = t % ntok, b = t / ntok, t2 = tok % f2, r1 = tok / f2, t1 = r1 % f1, t0 = r1 / f1;
    const int64_t vox = (((int64_t)b * S0 + t0 * p + p1) * S1 + t1 * p + p2) * S2 + t2 * p;
    const T* xr = x + vox * ld; T* orow = out + (int64_t)t * ldo + (p1 * p + p2) * p * C;      // token row pitch ldo (p^3 C when the rows are dense)
    if (!INV) {
      for (int j = lane; j < nin; j += 64) *(v4u*)(ls + 8 * j) = *(const v4u*)(xr + 8 * j);
      for (int j = lane; j < nout; j += 64) {
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const unsigned i0 = 8 * j + e, v0 = (i0 * cmul) >> 20, i1 = i0 + 1, v1 = (i1 * cmul) >> 20;   // v = i / C (exact: i*C < 2^20)
          const unsigned a = __builtin_bit_cast(unsigned short, ls[v0 * ld + (i0 - v0 * C)]), c = __builtin_bit_cast(unsigned short, ls[v1 * ld + (i1 - v1 * C)]);
          w[e >> 1] = a | (c << 16);
        }
        *(v4u*)(orow + 8 * j) = (v4u){w[0], w[1], w[2], w[3]};
      }
    } else {   // token-row gradient -> voxel rows (pad channels zero)
      T* xw = const_cast<T*>(xr);
      for (int j = lane; j < nout; j += 64) *(v4u*)(ls + 8 * j) = *(const v4u*)(orow + 8 * j);
      for (int j = lane; j < nin; j += 64) {
        const int v = j / cg8, c0 = (j - v * cg8) * 8;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const unsigned a = (c0 + e < C) ? __builtin_bit_cast(unsigned short, ls[v * C + c0 + e]) : 0u;
          const unsigned c = (c0 + e + 1 < C) ? __builtin_bit_cast(unsigned short, ls[v * C + c0 + e + 1]) : 0u;
          w[e >> 1] = a | (c << 16);
        }
        *(v4u*)(xw + 8 * j) = (v4u){w[0], w[1], w[2], w[3]};
      }
    }
  }
}
static bool patchify_fast(const void* x, const void* out, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, int64_t ldo = 0) {
  const int64_t nruns = (int64_t)B * (S0 / p) * (S1 / p) * (S2 / p) * p * p;
  return dtype != DP_F32 && (ld & 7) == 0 && ((p * C) & 7) == 0 && (ldo & 7) == 0 && p * ld <= 1024 && p * C <= 1024 && (int64_t)p * C * C < (1 << 20) && nruns < (1ll << 31) &&
         ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
}
template <bool INV>
static void patchify_runs_launch(const void* x, void* out, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, hipStream_t st, int64_t ldo = 0) {
  const int nruns = B * (S0 / p) * (S1 / p) * (S2 / p) * p * p;
  const unsigned cmul = ((1u << 20) + C - 1) / C;
  const int grid = grid_for(nruns, 4, 256 * 32);
  if (ldo <= 0) ldo = (int64_t)p * p * p * C;
  if (dtype == DP_BF16) hipLaunchKernelGGL((k_patchify_runs<bf16_t, INV>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, nruns, S0, S1, S2, C, ld, p, cmul, ldo);
  else hipLaunchKernelGGL((k_patchify_runs<f16_t, INV>), dim3(grid), dim3(256), 0, st, (const f16_t*)x, (f16_t*)out, nruns, S0, S1, S2, C, ld, p, cmul, ldo);
}
// dp_patchify into token rows of pitch ldo >= p^3 C (16-bit storage, fast path shapes only): the fp32x3 mode writes the hi / lo
// halves of the patch-embedding input straight into the column blocks of its split operand matrix.
extern "C" int dp_patchify_ld(const void* x, void* out, int64_t ldo, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream) {
  if (S0 % p || S1 % p || S2 % p) DP_FAIL("patchify_ld: size not divisible by patch");
  if (ldo < (int64_t)p * p * p * C || !patchify_fast(x, out, B, S0, S1, S2, C, ld, p, dtype, ldo)) DP_FAIL("patchify_ld: shape outside the 16-bit run kernel");
  patchify_runs_launch<false>(x, out, B, S0, S1, S2, C, ld, p, dtype, STREAM, ldo);
  DP_CHECK_LAUNCH("patchify_ld"); return 0;
}
extern "C" int dp_patchify(const void* x, void* out, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream) {
  if (S0 % p || S1 % p || S2 % p) DP_FAIL("patchify: size not divisible by patch");
  int64_t total = (int64_t)B * S0 * S1 * S2;
  if (patchify_fast(x, out, B, S0, S1, S2, C, ld, p, dtype)) patchify_runs_launch<false>(x, out, B, S0, S1, S2, C, ld, p, dtype, STREAM);
  else DP_DISPATCH(dtype, hipLaunchKernelGGL((k_patchify<T, false>), dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)x, (T*)out, B, S0, S1, S2, C, ld, p));
  DP_CHECK_LAUNCH("patchify"); return 0;
}
extern "C" int dp_unpatchify(const void* gout, void* gx, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream) {
  if (S0 % p || S1 % p || S2 % p) DP_FAIL("unpatchify: size not divisible by patch");
  int64_t total = (int64_t)B * S0 * S1 * S2;
  if (patchify_fast(gx, gout, B, S0, S1, S2, C, ld, p, dtype)) patchify_runs_launch<true>(gx, const_cast<void*>(gout), B, S0, S1, S2, C, ld, p, dtype, STREAM);
  else DP_DISPATCH(dtype, hipLaunchKernelGGL((k_patchify<T, true>), dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)gx, (T*)const_cast<void*>(gout), B, S0, S1, S2, C, ld, p));
  DP_CHECK_LAUNCH("unpatchify"); return 0;
}

// pixel shuffle (2x2x2): src [V][8][C] <-> dst NDHWC(2D,2H,2W) pitch ldd
template <typename T, bool INV, typename I>
__device__ __forceinline__ void pixel_shuffle2_body(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, I total, I cg8, int D, int H, int W,
                                                    int C) {
  const I Wu = W, Hu = H, Du = D;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I r = i / cg8; const int cg = (int)(i - r * cg8), abc = (int)(r & 7); const I v = r >> 3;
    I t = v; const I q1 = t / Wu; const int w = (int)(t - q1 * Wu); t = q1;
    const I q2 = t / Hu; const int h = (int)(t - q2 * Hu); t = q2;
    const I n = t / Du; const int d = (int)(t - n * Du);
    const int a = abc >> 2, b = (abc >> 1) & 1, c = abc & 1;
    const int64_t ov = (((int64_t)n * 2 * D + 2 * d + a) * 2 * H + 2 * h + b) * 2 * W + 2 * w + c;
    const int nv = min(8, C - cg * 8);
    if (!INV) {  // src compact [v][abc][C] -> dst big (pitch ldd)
      Frag8<T> f = frag_load(src + ((int64_t)v * 8 + abc) * C + cg * 8, nv);
      frag_store<T>(dst + ov * ldd + cg * 8, f, nv);
    } else {     // src big (pitch lds) -> dst compact
      Frag8<T> f = frag_load(src + ov * lds + cg * 8, nv);
      frag_store<T>(dst + ((int64_t)v * 8 + abc) * C + cg * 8, f, nv);
    }
  }
}
template <typename T, bool INV>
__global__ void k_pixel_shuffle2(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int N, int D, int H, int W, int C) {
  const int cg8 = (C + 7) >> 3;
  const int64_t total = (int64_t)N * D * H * W * 8 * cg8;
  if (total < (1ll << 31)) pixel_shuffle2_body<T, INV, unsigned>(src, lds, dst, ldd, (unsigned)total, (unsigned)cg8, D, H, W, C);
  else pixel_shuffle2_body<T, INV, int64_t>(src, lds, dst, ldd, total, (int64_t)cg8, D, H, W, C);
}
extern "C" int dp_pixel_shuffle2(const void* src, void* dst, int N, int D, int H, int W, int C, int ldd, int dtype, void* stream) {
  int64_t total = (int64_t)N * D * H * W * 8 * ((C + 7) / 8);
  DP_DISPATCH(dtype, hipLaunchKernelGGL((k_pixel_shuffle2<T, false>), dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)src, 0, (T*)dst, ldd, N, D, H, W, C));
  DP_CHECK_LAUNCH("pixel_shuffle2"); return 0;
}
extern "C" int dp_pixel_unshuffle2(const void* src, int lds, void* dst, int N, int D, int H, int W, int C, int dtype, void* stream) {
  int64_t total = (int64_t)N * D * H * W * 8 * ((C + 7) / 8);
  DP_DISPATCH(dtype, hipLaunchKernelGGL((k_pixel_shuffle2<T, true>), dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)src, lds, (T*)dst, 0, N, D, H, W, C));
  DP_CHECK_LAUNCH("pixel_unshuffle2"); return 0;
}

// trilinear x2, align_corners=True: src = dst * (in-1)/(out-1) (computed in fp32 like ATen's area_pixel_compute_scale)
__device__ __forceinline__ void tri_coord(int o, int n_in, int& i0, int& i1, float& f) {
  if (n_in == 1) { i0 = i1 = 0; f = 0.f; return; }
  float scale = (float)(n_in - 1) / (float)(2 * n_in - 1);
  float s = scale * (float)o;
  i0 = (int)s; if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  f = s - (float)i0;
}
// SPLIT (fp32x3 mode, round 6): fp32 input, the result written as the bf16 [hi | lo] operand of the x3 convolution that consumes it (hi at
// channel c, lo at c + ldy / 2 of a 2 cp-channel row): the fp32 up-sampled tensor and the split pass over it never exist.
template <typename T, typename I, bool SPLIT = false>
__device__ __forceinline__ void trilinear_fwd_body(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, I total, I cg8, int D, int H, int W,
                                                   int C) {
  const I W2 = 2 * W, H2 = 2 * H, D2 = 2 * D;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I ov = i / cg8; const int cg = (int)(i - ov * cg8);
    I t = ov; const I q1 = t / W2; const int ow = (int)(t - q1 * W2); t = q1;
    const I q2 = t / H2; const int oh = (int)(t - q2 * H2); t = q2;
    const I n = t / D2; const int od = (int)(t - n * D2);
    int d0, d1, h0, h1, w0, w1; float fd, fh, fw;
    tri_coord(od, D, d0, d1, fd); tri_coord(oh, H, h0, h1, fh); tri_coord(ow, W, w0, w1, fw);
    const int nv = min(8, C - cg * 8);
    Frag8<T> in[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int dd = (k & 4) ? d1 : d0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
      in[k] = frag_load(x + ((((int64_t)n * D + dd) * H + hh) * W + ww) * ldx + cg * 8, nv);
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float wgt = ((k & 4) ? fd : 1.f - fd) * ((k & 2) ? fh : 1.f - fh) * ((k & 1) ? fw : 1.f - fw);
      float tt[8];
      frag_unpack(in[k], tt);
      for (int j = 0; j < 8; j++) acc[j] += wgt * tt[j];
    }
    if constexpr (SPLIT) {
      static_assert(sizeof(T) == 4, "split output: fp32 input");
      bf16_t* p = (bf16_t*)y + (int64_t)ov * ldy + cg * 8;
      v4u hi, lo;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bf16_t h0 = f2bf(acc[2 * r]), h1 = f2bf(acc[2 * r + 1]);
        const bf16_t l0 = f2bf(acc[2 * r] - bf2f(h0)), l1 = f2bf(acc[2 * r + 1] - bf2f(h1));
        hi[r] = (unsigned)h0 | ((unsigned)h1 << 16); lo[r] = (unsigned)l0 | ((unsigned)l1 << 16);
      }
      *(v4u*)p = hi; *(v4u*)(p + ldy / 2) = lo;
    } else {
      Frag8<T> f; frag_pack(f, acc);
      frag_store<T>(y + (int64_t)ov * ldy + cg * 8, f, nv);
    }
  }
}
__global__ void k_trilinear_fwd_split(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N, int D, int H, int W, int C) {
  const int cg8 = C >> 3;
  const int64_t total = (int64_t)N * 8 * D * H * W * cg8;
  if (total < (1ll << 31)) trilinear_fwd_body<float, unsigned, true>(x, ldx, y, ldy, (unsigned)total, (unsigned)cg8, D, H, W, C);
  else trilinear_fwd_body<float, int64_t, true>(x, ldx, y, ldy, total, (int64_t)cg8, D, H, W, C);
}
// The same through LDS (round 4; 16-bit storage, C a multiple of 8 and <= 64: the 128^3 and 64^3 outputs of the C3D decoder).  One thread
// per (output voxel, 8 channels) gathers its eight corners from global memory: 8 x the output bytes through the CU's load path (2.1 GB
// for the 268 MB of the 32-channel 128^3 tensor: 156 us).  Here a block owns 2 x 8 x 32 output voxels, stages the <= 3 x 6 x 18 input
// voxels they read once (16-byte pieces, coalesced rows) and takes the corners from LDS; the arithmetic (weights, order of the eight
// FMAs) is trilinear_fwd_body's, so the results are bit-identical.
template <typename T>
__global__ void __launch_bounds__(256) k_trilinear_fwd_lds(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int D, int H, int W, int C) {
  constexpr int TD = 2, TH = 8, TW = 32, ED = 3, EH = 6, EW = 18;
  extern __shared__ __attribute__((aligned(16))) unsigned char tri_smem[];
  T* img = (T*)tri_smem;                                    // [ED][EH][EW][C]
  const int tid = threadIdx.x, cg8 = C >> 3;
  const int tiles_w = (2 * W + TW - 1) / TW, tiles_h = (2 * H + TH - 1) / TH;
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w; const int th = b % tiles_h; b /= tiles_h; const int odp = b % D, n = b / D;
  const int od0 = odp * TD, oh0 = th * TH, ow0 = tw * TW;
  const int od1 = min(od0 + TD, 2 * D) - 1, oh1 = min(oh0 + TH, 2 * H) - 1, ow1 = min(ow0 + TW, 2 * W) - 1;
  int d_lo, h_lo, w_lo, t0, t1; float tf;
  tri_coord(od0, D, d_lo, t1, tf); tri_coord(oh0, H, h_lo, t1, tf); tri_coord(ow0, W, w_lo, t1, tf);
  int d_hi, h_hi, w_hi;
  tri_coord(od1, D, t0, d_hi, tf); tri_coord(oh1, H, t0, h_hi, tf); tri_coord(ow1, W, t0, w_hi, tf);
  const int ed = d_hi - d_lo + 1, eh = h_hi - h_lo + 1, ew = w_hi - w_lo + 1;      // <= ED, EH, EW
  // stage: pieces (di, hi, wi, cg), cg fastest: a row of ew voxels is ew * C contiguous elements when ldx == C
  const int pieces = ed * eh * ew * cg8;
  for (int p = tid; p < pieces; p += 256) {
    const int cg = p % cg8; int t = p / cg8;
    const int wi = t % ew; t /= ew; const int hi = t % eh, di = t / eh;
    const v4u v = *(const v4u*)(x + ((((int64_t)n * D + d_lo + di) * H + h_lo + hi) * W + w_lo + wi) * ldx + cg * 8);
    *(v4u*)(img + (((di * EH + hi) * EW + wi) * cg8 + cg) * 8) = v;
  }
  __syncthreads();
  const int items = TD * TH * TW * cg8;
  for (int it = tid; it < items; it += 256) {
    const int cg = it % cg8; int t = it / cg8;
    const int owl = t % TW; t /= TW; const int ohl = t % TH, odl = t / TH;
    const int od = od0 + odl, oh = oh0 + ohl, ow = ow0 + owl;
    if (od > od1 || oh > oh1 || ow > ow1) continue;
    int d0, d1, h0, h1, w0, w1; float fd, fh, fw;
    tri_coord(od, D, d0, d1, fd); tri_coord(oh, H, h0, h1, fh); tri_coord(ow, W, w0, w1, fw);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int dd = ((k & 4) ? d1 : d0) - d_lo, hh = ((k & 2) ? h1 : h0) - h_lo, ww = ((k & 1) ? w1 : w0) - w_lo;
      Frag8<T> in; in.u = *(const v4u*)(img + (((dd * EH + hh) * EW + ww) * cg8 + cg) * 8);
      const float wgt = ((k & 4) ? fd : 1.f - fd) * ((k & 2) ? fh : 1.f - fh) * ((k & 1) ? fw : 1.f - fw);
      float tt[8];
      frag_unpack(in, tt);
      for (int j = 0; j < 8; j++) acc[j] += wgt * tt[j];
    }
    Frag8<T> f; frag_pack(f, acc);
    *(v4u*)(y + ((((int64_t)n * 2 * D + od) * 2 * H + oh) * 2 * W + ow) * ldy + cg * 8) = f.u;
  }
}
template <typename T>
__global__ void k_trilinear_fwd(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int D, int H, int W, int C) {
  const int cg8 = (C + 7) >> 3;
  const int64_t total = (int64_t)N * 8 * D * H * W * cg8;
  if (total < (1ll << 31)) trilinear_fwd_body<T, unsigned>(x, ldx, y, ldy, (unsigned)total, (unsigned)cg8, D, H, W, C);
  else trilinear_fwd_body<T, int64_t>(x, ldx, y, ldy, total, (int64_t)cg8, D, H, W, C);
}
template <typename T>
__global__ void k_trilinear_bwd(const T* __restrict__ gy, int ldgy, float* __restrict__ gx, int N, int D, int H, int W, int C) {
  int64_t OV = (int64_t)N * 8 * D * H * W, total = OV * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t ov = i / C;
    int ow = (int)(ov % (2 * W)), oh = (int)((ov / (2 * W)) % (2 * H)), od = (int)((ov / ((int64_t)4 * W * H)) % (2 * D));
    int64_t n = ov / ((int64_t)8 * W * H * D);
    int d0, d1, h0, h1, w0, w1; float fd, fh, fw;
    tri_coord(od, D, d0, d1, fd); tri_coord(oh, H, h0, h1, fh); tri_coord(ow, W, w0, w1, fw);
    float g = ld_f(gy + ov * ldgy + c);
    for (int k = 0; k < 8; k++) {
      int dd = (k & 4) ? d1 : d0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
      float wgt = ((k & 4) ? fd : 1.f - fd) * ((k & 2) ? fh : 1.f - fh) * ((k & 1) ? fw : 1.f - fw);
      if (wgt != 0.f) atomicAdd(gx + (((n * D + dd) * H + hh) * W + ww) * C + c, wgt * g);
    }
  }
}
extern "C" int dp_trilinear_up2_fwd(const void* x, int ldx, void* y, int ldy, int N, int D, int H, int W, int C, int dtype, void* stream) {
  if (dtype == DP_X3) {        // fp32 rows in, bf16 [hi | lo] rows of ldy = 2 cp channels out (cp >= C, both multiples of 8)
    if (C % 8 || ldy % 16 || ldy / 2 < C || ldx % 4 || (((uintptr_t)x | (uintptr_t)y) & 15)) DP_FAIL("trilinear_up2_fwd (DP_X3): C %% 8, ldy = 2 cp >= 2 C, 16-byte aligned rows");
    const int64_t total = (int64_t)N * 8 * D * H * W * (C / 8);
    hipLaunchKernelGGL(k_trilinear_fwd_split, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const float*)x, ldx, (float*)y, ldy, N, D, H, W, C);
    DP_CHECK_LAUNCH("trilinear_fwd_split"); return 0;
  }
  static const int lds_on = [] { const char* e = getenv("DP_TRILINEAR_LDS"); return e ? atoi(e) : 1; }();
  if (lds_on && (dtype == DP_BF16 || dtype == DP_F16) && C % 8 == 0 && C <= 64 && ldx % 8 == 0 && ldy % 8 == 0 && W >= 16 && H >= 4 &&
      (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const int64_t blocks = (int64_t)N * D * ((2 * H + 7) / 8) * ((2 * W + 31) / 32);
    if (blocks < 2000000000LL) {
      const size_t smem = (size_t)3 * 6 * 18 * C * 2;
      if (dtype == DP_BF16) hipLaunchKernelGGL(k_trilinear_fwd_lds<bf16_t>, dim3((unsigned)blocks), dim3(256), smem, STREAM, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, D, H, W, C);
      else hipLaunchKernelGGL(k_trilinear_fwd_lds<f16_t>, dim3((unsigned)blocks), dim3(256), smem, STREAM, (const f16_t*)x, ldx, (f16_t*)y, ldy, N, D, H, W, C);
      DP_CHECK_LAUNCH("trilinear_fwd_lds"); return 0;
    }
  }
  int64_t total = (int64_t)N * 8 * D * H * W * ((C + 7) / 8);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_trilinear_fwd<T>, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)x, ldx, (T*)y, ldy, N, D, H, W, C));
  DP_CHECK_LAUNCH("trilinear_fwd"); return 0;
}
// The same as a GATHER (deterministic mode): one thread per input element walks the <= 8 output coordinates per axis that can read it
// (src = o (n - 1) / (2n - 1) ~ o / 2: outputs 2i - 3 .. 2i + 4 cover every case), with the very weights the forward pass computes, and
// adds them in a fixed order; gx is OVERWRITTEN.
template <typename T>
__global__ void k_trilinear_bwd_gather(const T* __restrict__ gy, int ldgy, float* __restrict__ gx, int N, int D, int H, int W, int C) {
  const int64_t IV = (int64_t)N * D * H * W, total = IV * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); int64_t v = i / C;
    const int w = (int)(v % W); v /= W; const int h = (int)(v % H); v /= H; const int d = (int)(v % D); const int64_t n = v / D;
    float wd[8], wh[8], ww[8];
    auto weights = [](int idx, int n_in, float* out) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int o = 2 * idx - 3 + j;
        float wsum = 0.f;
        if (o >= 0 && o < 2 * n_in) {
          int i0, i1; float f; tri_coord(o, n_in, i0, i1, f);
          if (i0 == idx) wsum += 1.f - f;
          if (i1 == idx) wsum += f;
        }
        out[j] = wsum;
      }
    };
    weights(d, D, wd); weights(h, H, wh); weights(w, W, ww);
    float acc = 0.f;
    for (int a = 0; a < 8; a++) {
      if (wd[a] == 0.f) continue;
      const int od = 2 * d - 3 + a;
      for (int b = 0; b < 8; b++) {
        if (wh[b] == 0.f) continue;
        const int oh = 2 * h - 3 + b;
        for (int e = 0; e < 8; e++) {
          if (ww[e] == 0.f) continue;
          const int ow = 2 * w - 3 + e;
          acc += wd[a] * wh[b] * ww[e] * ld_f(gy + ((((int64_t)n * 2 * D + od) * 2 * H + oh) * 2 * W + ow) * ldgy + c);
        }
      }
    }
    gx[i] = acc;
  }
}
extern "C" int dp_trilinear_up2_bwd(const void* gy, int ldgy, float* gx, int N, int D, int H, int W, int C, int dtype, void* stream) {
  if (dp_det(DET_TRILINEAR)) {
    const int64_t tot = (int64_t)N * D * H * W * C;
    DP_DISPATCH(dtype, hipLaunchKernelGGL(k_trilinear_bwd_gather<T>, dim3(grid_for(tot, 256)), dim3(256), 0, STREAM, (const T*)gy, ldgy, gx, N, D, H, W, C));
    DP_CHECK_LAUNCH("trilinear_bwd_gather"); return 0;
  }
  int64_t total = (int64_t)N * 8 * D * H * W * C;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_trilinear_bwd<T>, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const T*)gy, ldgy, gx, N, D, H, W, C));
  DP_CHECK_LAUNCH("trilinear_bwd"); return 0;
}

// batched transpose through a 32x33 LDS tile
template <typename T>
__global__ void k_transpose(const T* __restrict__ src, int64_t lds, int64_t sb0, int64_t sb1, T* __restrict__ dst, int64_t ldd,
                            int64_t db0, int64_t db1, int rows, int cols, int nb1) {
  __shared__ float tile[32][33];
  int b = blockIdx.z, b0 = b / nb1, b1 = b % nb1;
  const T* s = src + b0 * sb0 + b1 * sb1; T* d = dst + b0 * db0 + b1 * db1;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) { int r = r0 + j, c = c0 + tx; tile[j][tx] = (r < rows && c < cols) ? ld_f(s + (int64_t)r * lds + c) : 0.f; }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) { int c = c0 + j, r = r0 + tx; if (r < rows && c < cols) st_f(d + (int64_t)c * ldd + r, tile[tx][j]); }
}
extern "C" int dp_transpose(const void* src, int64_t lds, int64_t sb0, int64_t sb1, void* dst, int64_t ldd, int64_t db0, int64_t db1,
                            int rows, int cols, int nb0, int nb1, int dtype, void* stream) {
  dim3 g(cdiv(cols, 32), cdiv(rows, 32), nb0 * nb1);
  if (g.y > 65535 || g.z > 65535) DP_FAIL("transpose: grid too large");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_transpose<T>, g, dim3(256), 0, STREAM, (const T*)src, lds, sb0, sb1, (T*)dst, ldd, db0, db1, rows, cols, nb1));
  DP_CHECK_LAUNCH("transpose"); return 0;
}

// ------------------------------------------------------------------------------------------------ elementwise
template <typename T>
__global__ void k_add(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n, int64_t period) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st_f(y + i, ld_f(a + i) + ld_f(b + (i % period)));
}
extern "C" int dp_add(const void* a, const void* b, void* y, int64_t n, int64_t period, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_add<T>, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, (const T*)a, (const T*)b, (T*)y, n, period));
  DP_CHECK_LAUNCH("add"); return 0;
}
// out[i] (fp32) = sum_b g[b * per + i]: gradient of a parameter broadcast over the batch (position embeddings)
template <typename T>
__global__ void k_sum_batch(const T* __restrict__ g, float* __restrict__ out, int B, int64_t per) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; b++) s += ld_f(g + b * per + i);
    out[i] = s;
  }
}
extern "C" int dp_sum_batch(const void* g, float* out, int B, int64_t per, int dtype, void* stream) {
  if (B < 1 || per < 1) DP_FAIL("sum_batch: empty problem");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_sum_batch<T>, dim3(grid_for(per, 256)), dim3(256), 0, STREAM, (const T*)g, out, B, per));
  DP_CHECK_LAUNCH("sum_batch"); return 0;
}
template <typename T, bool BWD>
__global__ void k_gelu(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float z = ld_f(x + i);
    st_f(out + i, BWD ? ld_f(gy + i) * act_bwd(z, DP_ACT_GELU) : act_fwd(z, DP_ACT_GELU));
  }
}
extern "C" int dp_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL((k_gelu<T, false>), dim3(grid_for(n, 256)), dim3(256), 0, STREAM, (const T*)x, (const T*)nullptr, (T*)y, n));
  DP_CHECK_LAUNCH("gelu_fwd"); return 0;
}
extern "C" int dp_gelu_bwd(const void* x, const void* gy, void* gx, int64_t n, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL((k_gelu<T, true>), dim3(grid_for(n, 256)), dim3(256), 0, STREAM, (const T*)x, (const T*)gy, (T*)gx, n));
  DP_CHECK_LAUNCH("gelu_bwd"); return 0;
}

// softmax over rows: one wave per row (4 rows per 256-thread block)
template <typename T>
__global__ void k_softmax_fwd(const T* __restrict__ s, T* __restrict__ p, int64_t rows, int cols, float scale) {
  int lane = threadIdx.x & 63; int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* a = s + row * cols; T* o = p + row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, ld_f(a + c) * scale);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += expf(ld_f(a + c) * scale - m);
  sum = wave_sum(sum);
  float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) st_f(o + c, expf(ld_f(a + c) * scale - m) * inv);
}
template <typename T>
__global__ void k_softmax_bwd(const T* __restrict__ p, const T* __restrict__ gp, T* __restrict__ gs, int64_t rows, int cols, float scale) {
  int lane = threadIdx.x & 63; int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* a = p + row * cols; const T* g = gp + row * cols; T* o = gs + row * cols;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += ld_f(a + c) * ld_f(g + c);
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) { float pv = ld_f(a + c); st_f(o + c, scale * pv * (ld_f(g + c) - dot)); }
}
extern "C" int dp_softmax_fwd(const void* s, void* p, int64_t rows, int cols, float scale, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_softmax_fwd<T>, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, (const T*)s, (T*)p, rows, cols, scale));
  DP_CHECK_LAUNCH("softmax_fwd"); return 0;
}
extern "C" int dp_softmax_bwd(const void* p, const void* gp, void* gs, int64_t rows, int cols, float scale, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_softmax_bwd<T>, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, (const T*)p, (const T*)gp, (T*)gs, rows, cols, scale));
  DP_CHECK_LAUNCH("softmax_bwd"); return 0;
}

// ------------------------------------------------------------------------------------------------ cascade glue
template <typename T, typename TO = T>
__global__ void k_argmax_onehot(const T* __restrict__ lg, int ld, TO* __restrict__ out, int ldo, int choff, int32_t* labels, int64_t rows, int C) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const T* a = lg + i * ld; float best = ld_f(a); int bi = 0;
    for (int c = 1; c < C; c++) { float v = ld_f(a + c); if (v > best) { best = v; bi = c; } }
    if (labels) labels[i] = bi;
    if (out) for (int c = 1; c < C; c++) st_f(out + i * ldo + choff + c - 1, c == bi ? 1.f : 0.f);
  }
}
extern "C" int dp_argmax_onehot(const void* logits, int ld, void* out, int ldo, int choff, int32_t* labels, int64_t rows, int C, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_argmax_onehot<T>, dim3(grid_for(rows, 256)), dim3(256), 0, STREAM, (const T*)logits, ld, (T*)out, ldo, choff, labels, rows, C));
  DP_CHECK_LAUNCH("argmax_onehot"); return 0;
}
// logits and one-hot destination of different storage types: the cascade runs its segmentation network in fp32x3 (fp32 logits, the
// reference's masks) while the dose network's staging buffer is in the training storage type.
extern "C" int dp_argmax_onehot2(const void* logits, int ld, int logits_dtype, void* out, int ldo, int out_dtype, int choff, int32_t* labels,
                                 int64_t rows, int C, void* stream) {
  if (logits_dtype == out_dtype || !out) return dp_argmax_onehot(logits, ld, out, ldo, choff, labels, rows, C, logits_dtype, stream);
#define GO2(TL, TOUT) hipLaunchKernelGGL((k_argmax_onehot<TL, TOUT>), dim3(grid_for(rows, 256)), dim3(256), 0, STREAM, (const TL*)logits, ld, (TOUT*)out, ldo, choff, labels, rows, C)
  if (logits_dtype == DP_F32 && out_dtype == DP_BF16) GO2(float, bf16_t);
  else if (logits_dtype == DP_F32 && out_dtype == DP_F16) GO2(float, f16_t);
  else if (logits_dtype == DP_BF16 && out_dtype == DP_F32) GO2(bf16_t, float);
  else if (logits_dtype == DP_F16 && out_dtype == DP_F32) GO2(f16_t, float);
  else DP_FAIL("argmax_onehot2: unsupported type pair (%d, %d)", logits_dtype, out_dtype);
#undef GO2
  DP_CHECK_LAUNCH("argmax_onehot2"); return 0;
}

// ------------------------------------------------------------------------------------------------ skinny pointwise conv
// y[v][co] = sum_ci x[v][ci] * w[co][ci] (+ bias) for few channels (Cin <= 64, Cout <= 32) over millions of voxels:
// an HBM-bound row stream (AI ~ Cout FLOP/B), so one thread owns one voxel row, reads it as 16-byte chunks, keeps the
// Cout accumulators in registers and takes the weights from LDS (every lane reads the same address: a broadcast).
template <typename T, int COUT>
__global__ void __launch_bounds__(256) k_pointwise_rows(const T* __restrict__ x, int ldx, const T* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                        T* __restrict__ y, int ldy, int64_t rows, int Cin, int Cout) {
  __shared__ float ws[64 * COUT];
  for (int i = threadIdx.x; i < Cin * COUT; i += 256) { int ci = i / COUT, co = i - ci * COUT; ws[i] = co < Cout ? ld_f(w + (int64_t)co * ldw + ci) : 0.f; }
  __syncthreads();
  const int nch = (Cin + 7) >> 3;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < rows; v += (int64_t)gridDim.x * blockDim.x) {
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; c++) acc[c] = (bias && c < Cout) ? bias[c] : 0.f;
    for (int ch = 0; ch < nch; ch++) {
      float t[8];
      frag_unpack(frag_load(x + v * ldx + ch * 8, min(8, Cin - ch * 8)), t);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float* wr = ws + (ch * 8 + j) * COUT;
        if (ch * 8 + j < Cin) {
#pragma unroll
          for (int c = 0; c < COUT; c++) acc[c] += t[j] * wr[c];
        }
      }
    }
    T* o = y + v * ldy;
#pragma unroll
    for (int c8 = 0; c8 < COUT; c8 += 8) {
      if (c8 < Cout) { Frag8<T> f; frag_pack(f, acc + c8); frag_store<T>(o + c8, f, min(8, Cout - c8)); }
    }
  }
}
// The same row stream on the matrix cores (round 4; north_star: "MFMA ... for the 1x1x1 pointwise convs where they really are dense
// GEMMs"): with 32-64 input and 16-32 output channels the VALU version spends 512-2048 FMAs per voxel (the 64 -> 32 mixer of the 64^3
// level: 88 us for 100 MB) -- one v_mfma_f32_16x16x32 does 16 output channels x 16 voxels x 32 inputs.  No LDS at all: the weights are
// the A operand (rows = output channels, held in registers for the whole launch), a tile of 16 voxels is the B operand and the fragment
// of a lane IS one 16-byte global load (voxel lane % 16, channels 8 (lane / 16) .. + 7 of a 32-channel chunk).  The C layout then gives
// a lane four consecutive output rows of ONE voxel; with 32 outputs the weight rows are dealt to the two MFMA tiles so that those rows
// are channels 8q .. 8q+3 (tile 0) and 8q+4 .. 8q+7 (tile 1) -- the lane's eight results are one 16-byte store.  U tiles per trip are
// requested before the first MFMA.
template <typename T> __device__ __forceinline__ unsigned pw_pack2(float a, float b) {
  T lo, hi; st_f(&lo, a); st_f(&hi, b);
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}
// Generalised in the same round to every skinny row GEMM of the step (grid.y walks the output columns in tiles of 16 NT, up to 256
// input channels): ConvTranspose 2 x 2 x 2 forward (K = Cin, 8 Cout columns) and data gradient (K = 8 Cout), the data gradients of
// the mixers -- the generic 128 x 128-tile GEMM moved 0.9-1.7 TB/s on these shapes.  ShuffleGeom: ConvTranspose forward with the
// 2 x 2 x 2 pixel shuffle in the store: column group g = 4a + 2b + c (cg channels each) of input voxel (n, d, h, w) IS output voxel
// (n, 2d + a, 2h + b, 2w + c) -- the [voxels][8 Cout] intermediate and its shuffle pass (a read and a write of the whole upsampled
// tensor) do not exist.
struct ShuffleGeom { int cg, D, H, W; float rW, rH, rD; };
__device__ __forceinline__ unsigned div_small(unsigned v, unsigned d, float rd) {      // v / d for v < 2^24 (one multiply + fix-ups)
  unsigned qv = (unsigned)((float)v * rd);
  if (qv * d > v) qv--;
  if ((qv + 1) * d <= v) qv++;
  return qv;
}
template <typename T, int KC, int NT>
__global__ void __launch_bounds__(256) k_pointwise_mfma(const T* __restrict__ x, int ldx, const T* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                        T* __restrict__ y, int ldy, int64_t rows, int Cin, int Cout, ShuffleGeom sg) {
  constexpr int U = KC >= 5 ? 1 : KC >= 3 ? 2 : 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 16 * NT;
  Frag8<T> aw[NT][KC];
  unsigned keep[KC][4];                                       // dword masks of the channels < Cin (input padding may hold anything)
#pragma unroll
  for (int kc = 0; kc < KC; kc++) {
    const int k0 = kc * 32 + q * 8;
#pragma unroll
    for (int d = 0; d < 4; d++) keep[kc][d] = (k0 + 2 * d + 1 < Cin) ? 0xffffffffu : (k0 + 2 * d < Cin) ? 0x0000ffffu : 0u;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int co = n0 + (NT == 2 ? 8 * (r >> 2) + 4 * nt + (r & 3) : r);
      aw[nt][kc] = (co < Cout && k0 < Cin) ? frag_load(w + (int64_t)co * ldw + k0, min(8, Cin - k0)) : frag_zero<T>();
    }
  }
  const int c0 = n0 + (NT == 2 ? 8 * q : 4 * q);              // the lane's first output channel
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int e = 0; e < 4; e++) bv[nt][e] = (bias && c0 + 4 * nt + e < Cout) ? bias[c0 + 4 * nt + e] : 0.f;
  // pixel-shuffle store: the lane's channels all belong to one (a, b, c) group
  const int grp = sg.cg ? c0 / sg.cg : 0, cs = sg.cg ? c0 - grp * sg.cg : c0;
  const int ga = grp >> 2, gb = (grp >> 1) & 1, gc = grp & 1;
  const bool ragged = (Cin & 31) != 0;
  const int64_t ntiles = (rows + 15) / 16, wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t0 = wave * U; t0 < ntiles; t0 += nwaves * U) {
    Frag8<T> b[U][KC];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int64_t v = (t0 + u) * 16 + r; v = v < rows ? v : rows - 1;      // (tail rows re-read the last voxel: never stored)
#pragma unroll
      for (int kc = 0; kc < KC; kc++) b[u][kc].u = *(const v4u*)(x + v * ldx + (keep[kc][0] ? kc * 32 + q * 8 : 0));   // (chunks beyond Cin: any valid address, masked below)
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = (t0 + u) * 16 + r;
      // (the masks are all-ones / all-zeros except in the chunk that holds channel Cin - 1, which for the eight-chunk instance serving
      // 5-7 real chunks is not chunk KC - 1: every chunk from the last real one on is masked -- ADVICE r4)
      if (ragged)
#pragma unroll
        for (int kc = (KC == 8 ? 4 : KC - 1); kc < KC; kc++)
#pragma unroll
          for (int d = 0; d < 4; d++) b[u][kc].u[d] &= keep[kc][d];
      v4f acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        acc[nt] = (v4f){bv[nt][0], bv[nt][1], bv[nt][2], bv[nt][3]};
#pragma unroll
        for (int kc = 0; kc < KC; kc++) acc[nt] = mma16(aw[nt][kc], b[u][kc], acc[nt]);
      }
      if (v >= rows) continue;
      int64_t orow = v;
      if (sg.cg) {
        const unsigned vv = (unsigned)v, q1 = div_small(vv, sg.W, sg.rW), ww = vv - q1 * sg.W, q2 = div_small(q1, sg.H, sg.rH), hh = q1 - q2 * sg.H,
                       nn = div_small(q2, sg.D, sg.rD), dd = q2 - nn * sg.D;
        orow = (((int64_t)nn * 2 * sg.D + 2 * dd + ga) * 2 * sg.H + 2 * hh + gb) * 2 * sg.W + 2 * ww + gc;
      }
      T* yo = y + orow * ldy + cs;
      if (c0 + 4 * NT <= Cout) {
        if constexpr (NT == 2) {
          v4u o; o.x = pw_pack2<T>(acc[0][0], acc[0][1]); o.y = pw_pack2<T>(acc[0][2], acc[0][3]); o.z = pw_pack2<T>(acc[1][0], acc[1][1]); o.w = pw_pack2<T>(acc[1][2], acc[1][3]);
          *(v4u*)yo = o;
        } else {
          uint2 o; o.x = pw_pack2<T>(acc[0][0], acc[0][1]); o.y = pw_pack2<T>(acc[0][2], acc[0][3]);
          *(uint2*)yo = o;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int e = 0; e < 4; e++) if (c0 + 4 * nt + e < Cout) st_f(yo + 4 * nt + e, acc[nt][e]);
      }
    }
  }
}

// The fp32x3 mode's twin (dtype DP_X3 / DP_X1): fp32 rows and fp32 weights in HBM, split into bf16 halves IN REGISTERS (a lane's B
// fragment is two 16-byte loads of eight floats -> x_hi, x_lo), three products x_hi w_hi + x_lo w_hi + x_hi w_lo (DP_X3: forward
// passes) or x_hi w_hi alone (DP_X1: the one-product data gradients), fp32 results as 16-byte stores.  Replaces the exact-fp32 tiled
// GEMM on the skinny shapes (ConvTranspose forward / data gradient, mixer data gradients: 0.16-0.26 ms launches at 1.3 TB/s).
__device__ __forceinline__ void split8(const float* v, int nv, Frag8<bf16_t>& hi, Frag8<bf16_t>& lo) {
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const float a = 2 * d < nv ? v[2 * d] : 0.f, b = 2 * d + 1 < nv ? v[2 * d + 1] : 0.f;
    const bf16_t ah = f2bf(a), bh = f2bf(b);
    hi.u[d] = (unsigned)ah | ((unsigned)bh << 16);
    lo.u[d] = (unsigned)f2bf(a - bf2f(ah)) | ((unsigned)f2bf(b - bf2f(bh)) << 16);
  }
}
template <int KC, int NT>
__global__ void __launch_bounds__(256) k_rows_mfma_f32(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                       float* __restrict__ y, int ldy, int64_t rows, int Cin, int Cout, int terms, ShuffleGeom sg) {
  constexpr int U = KC >= 3 ? 1 : 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 16 * NT;
  Frag8<bf16_t> awh[NT][KC], awl[NT][KC];
  int nvk[KC];                                                // valid channels of the lane's eight in chunk kc
#pragma unroll
  for (int kc = 0; kc < KC; kc++) {
    const int k0 = kc * 32 + q * 8;
    nvk[kc] = Cin - k0 < 0 ? 0 : (Cin - k0 > 8 ? 8 : Cin - k0);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int co = n0 + (NT == 2 ? 8 * (r >> 2) + 4 * nt + (r & 3) : r);
      float wv8[8];
#pragma unroll
      for (int e = 0; e < 8; e++) wv8[e] = (co < Cout && e < nvk[kc]) ? w[(int64_t)co * ldw + k0 + e] : 0.f;
      split8(wv8, 8, awh[nt][kc], awl[nt][kc]);
    }
  }
  const int c0 = n0 + (NT == 2 ? 8 * q : 4 * q);
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int e = 0; e < 4; e++) bv[nt][e] = (bias && c0 + 4 * nt + e < Cout) ? bias[c0 + 4 * nt + e] : 0.f;
  const int grp = sg.cg ? c0 / sg.cg : 0, cs = sg.cg ? c0 - grp * sg.cg : c0;
  const int ga = grp >> 2, gb = (grp >> 1) & 1, gc = grp & 1;
  const bool three = terms == 3;
  const int64_t ntiles = (rows + 15) / 16, wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t0 = wave * U; t0 < ntiles; t0 += nwaves * U) {
    v4f raw[U][KC][2];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int64_t v = (t0 + u) * 16 + r; v = v < rows ? v : rows - 1;
#pragma unroll
      for (int kc = 0; kc < KC; kc++) {
        const float* p = x + v * ldx + (nvk[kc] ? kc * 32 + q * 8 : 0);      // (chunks beyond Cin: any valid address, zeroed by the split)
        raw[u][kc][0] = *(const v4f*)p; raw[u][kc][1] = *(const v4f*)(p + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = (t0 + u) * 16 + r;
      v4f acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[nt] = (v4f){bv[nt][0], bv[nt][1], bv[nt][2], bv[nt][3]};
#pragma unroll
      for (int kc = 0; kc < KC; kc++) {
        const float f8[8] = {raw[u][kc][0][0], raw[u][kc][0][1], raw[u][kc][0][2], raw[u][kc][0][3], raw[u][kc][1][0], raw[u][kc][1][1], raw[u][kc][1][2], raw[u][kc][1][3]};
        Frag8<bf16_t> bh, bl;
        split8(f8, nvk[kc], bh, bl);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          acc[nt] = mma16(awh[nt][kc], bh, acc[nt]);
          if (three) { acc[nt] = mma16(awh[nt][kc], bl, acc[nt]); acc[nt] = mma16(awl[nt][kc], bh, acc[nt]); }
        }
      }
      if (v >= rows) continue;
      int64_t orow = v;
      if (sg.cg) {
        const unsigned vv = (unsigned)v, q1 = div_small(vv, sg.W, sg.rW), ww = vv - q1 * sg.W, q2 = div_small(q1, sg.H, sg.rH), hh = q1 - q2 * sg.H,
                       nn = div_small(q2, sg.D, sg.rD), dd = q2 - nn * sg.D;
        orow = (((int64_t)nn * 2 * sg.D + 2 * dd + ga) * 2 * sg.H + 2 * hh + gb) * 2 * sg.W + 2 * ww + gc;
      }
      float* yo = y + orow * ldy + cs;
      if (c0 + 4 * NT <= Cout) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) *(v4f*)(yo + 4 * nt) = acc[nt];
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int e = 0; e < 4; e++) if (c0 + 4 * nt + e < Cout) yo[4 * nt + e] = acc[nt][e];
      }
    }
  }
}

// Shapes the matrix-core row kernel takes: 16-bit storage, 16 <= Cin <= 256, more than 8 output columns, rows readable in 16-byte
// pieces up to Cin rounded up to 8, 16-byte aligned output rows (pointer alignment is checked at the launch).
extern "C" int dp_rows_mfma_ok(int ldx, int ldw, int ldy, int Cin, int Cout, int dtype) {
  static const int mfma_on = [] { const char* e = getenv("DP_POINTWISE_MFMA"); return e ? atoi(e) : 1; }();
  if (dtype == DP_X3 || dtype == DP_X1)      // fp32 rows, split in registers: up to 128 input channels, rows readable / writable in 16-byte pieces
    return mfma_on && Cin >= 16 && Cin <= 128 && Cout > 8 && (Cin + 7) / 8 * 8 <= ldx && ldx % 4 == 0 && ldy % 4 == 0 && ldw >= Cin;
  return mfma_on && (dtype == DP_BF16 || dtype == DP_F16) && Cin >= 16 && Cin <= 256 && Cout > 8 && (Cin + 7) / 8 * 8 <= ldx && ldx % 8 == 0 && ldy % 8 == 0 &&
         ldw >= Cin;
}
static int rows_mfma_launch(const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int64_t rows, int Cin, int Cout, int dtype,
                            ShuffleGeom sg, hipStream_t s) {
  const int kc = (Cin + 31) / 32, nt = Cout > 16 ? 2 : 1, ny = (Cout + 16 * nt - 1) / (16 * nt);
  const int64_t tiles = (rows + 15) / 16;
  int64_t g = (tiles + 15) / 16; if (g * ny > 4096) g = (4096 + ny - 1) / ny; if (g < 1) g = 1;
#define GOM(TT, KC_, NT_) hipLaunchKernelGGL((k_pointwise_mfma<TT, KC_, NT_>), dim3((unsigned)g, ny), dim3(256), 0, s, (const TT*)x, ldx, (const TT*)w, ldw, bias, (TT*)y, ldy, rows, Cin, Cout, sg)
#define GOMK(TT, NT_) do { if (kc == 1) GOM(TT, 1, NT_); else if (kc == 2) GOM(TT, 2, NT_); else if (kc == 3) GOM(TT, 3, NT_); else if (kc == 4) GOM(TT, 4, NT_); \
                            else GOM(TT, 8, NT_); } while (0)      /* (5-8 chunks: the eight-chunk instance, absent chunks masked) */
#define GOMT(TT) do { if (nt == 1) GOMK(TT, 1); else GOMK(TT, 2); } while (0)
#define GOF(KC_, NT_) hipLaunchKernelGGL((k_rows_mfma_f32<KC_, NT_>), dim3((unsigned)g, ny), dim3(256), 0, s, (const float*)x, ldx, (const float*)w, ldw, bias, (float*)y, ldy, rows, Cin, Cout, dtype == DP_X3 ? 3 : 1, sg)
#define GOFK(NT_) do { if (kc == 1) GOF(1, NT_); else if (kc == 2) GOF(2, NT_); else if (kc == 3) GOF(3, NT_); else GOF(4, NT_); } while (0)
  if (dtype == DP_X3 || dtype == DP_X1) { if (nt == 1) GOFK(1); else GOFK(2); }
  else if (dtype == DP_BF16) GOMT(bf16_t); else GOMT(f16_t);
#undef GOFK
#undef GOF
#undef GOMT
#undef GOMK
#undef GOM
  return 0;
}
// nn.ConvTranspose3d(kernel 2, stride 2, bias = False) forward in ONE launch (base_blocks.py:118-127): w = the [(abc, co)][ldw >= Cin]
// pack of dp_pack_multi / _pack_tconv, y the NDHWC tensor of the (2D, 2H, 2W) volume.  Needs dp_rows_mfma_ok(ldx, ldw, ldy, Cin,
// 8 Cout, dtype), Cout % 8 == 0 and N D H W < 2^24; returns 3 ("not this kernel's shape") otherwise without launching.
extern "C" int dp_tconv2x_fwd(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int dtype,
                              void* stream) {
  const int64_t rows = (int64_t)N * D * H * W;
  if (rows <= 0) return 0;
  if (!dp_rows_mfma_ok(ldx, ldw, ldy, Cin, 8 * Cout, dtype) || Cout % 8 || rows >= (1 << 24) || (((uintptr_t)x | (uintptr_t)y) & 15) != 0) return 3;
  ShuffleGeom sg; sg.cg = Cout; sg.D = D; sg.H = H; sg.W = W; sg.rW = 1.f / (float)W; sg.rH = 1.f / (float)H; sg.rD = 1.f / (float)D;
  rows_mfma_launch(x, ldx, w, ldw, nullptr, y, ldy, rows, Cin, 8 * Cout, dtype, sg, STREAM);
  DP_CHECK_LAUNCH("tconv2x_fwd"); return 0;
}

extern "C" int dp_pointwise_rows(const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int64_t rows, int Cin, int Cout,
                                 int dtype, void* stream) {
  if (Cin < 1 || Cout < 1) DP_FAIL("pointwise_rows: empty channel count");
  if (dp_rows_mfma_ok(ldx, ldw, ldy, Cin, Cout, dtype) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && rows >= 16) {
    ShuffleGeom sg; sg.cg = 0; sg.D = sg.H = sg.W = 1; sg.rW = sg.rH = sg.rD = 1.f;
    rows_mfma_launch(x, ldx, w, ldw, bias, y, ldy, rows, Cin, Cout, dtype, sg, STREAM);
    DP_CHECK_LAUNCH("pointwise_mfma"); return 0;
  }
  if (Cin > 64 || Cout > 32) DP_FAIL("pointwise_rows: outside the matrix-core kernel's shapes (dp_rows_mfma_ok) the row stream needs Cin <= 64 and Cout <= 32");
  int g = grid_for(rows, 256, 256 * 32);
#define GO(CO) DP_DISPATCH(dtype, hipLaunchKernelGGL((k_pointwise_rows<T, CO>), dim3(g), dim3(256), 0, STREAM, (const T*)x, ldx, (const T*)w, ldw, bias, (T*)y, ldy, rows, Cin, Cout))
  if (Cout <= 8) GO(8); else if (Cout <= 16) GO(16); else GO(32);
#undef GO
  DP_CHECK_LAUNCH("pointwise_rows"); return 0;
}

// Weight (and bias) gradient of the same skinny pointwise conv: dw[co][ci] = sum_v gy[v][co] x[v][ci], db[co] = sum_v gy[v][co].
// One 16-byte chunk of one voxel row per thread (CPR = Cin/8 chunks per row, a power of two), COUT x 8 accumulators in
// registers over a grid-stride sweep, a wave shuffle tree, an LDS sum over the 4 waves and one partial row per block in
// the workspace; a second tiny kernel adds the partials in a fixed order (deterministic, no same-address atomics).
template <typename T, int COUT>
__global__ void __launch_bounds__(256) k_pointwise_wgrad_rows(const T* __restrict__ x, int ldx, const T* __restrict__ gy, int ldgy, float* __restrict__ ws,
                                                              int64_t rows, int Cin, int Cout, int cpr_log2) {
  __shared__ float sm[4][COUT][72];     // [wave][co][ci | 64: bias]
  const int CPR = 1 << cpr_log2, RPI = 256 >> cpr_log2, chunk = threadIdx.x & (CPR - 1), roff = threadIdx.x >> cpr_log2;
  float acc[COUT][8], accb[COUT];
#pragma unroll
  for (int c = 0; c < COUT; c++) { accb[c] = 0.f; for (int j = 0; j < 8; j++) acc[c][j] = 0.f; }
  const int nv = min(8, Cin - chunk * 8);
  // U rows per trip with every load issued before the first FMA, the gy row as 16-byte pieces (round 4: one row per trip with COUT
  // scalar 2-byte loads was a dependent load -> FMA chain: 1.2 TB/s on the 7-OAR head of OAR-TRANSEG, 165 us for 201 MB)
  constexpr int U = COUT >= 16 ? 2 : 4, GP = (COUT + 7) / 8;
  const int64_t stride = (int64_t)gridDim.x * RPI;
  for (int64_t r0 = (int64_t)blockIdx.x * RPI + roff; r0 < rows; r0 += stride * U) {
    Frag8<T> xf[U], gf[U][GP];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t r = r0 + u * stride;
      const bool ok = r < rows;
      xf[u] = ok ? frag_load(x + r * ldx + chunk * 8, nv) : frag_zero<T>();
#pragma unroll
      for (int p8 = 0; p8 < GP; p8++) {
        const int ng = min(8, Cout - p8 * 8);
        gf[u][p8] = (ok && ng > 0) ? frag_load(gy + r * ldgy + p8 * 8, ng) : frag_zero<T>();
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float t[8];
      frag_unpack(xf[u], t);
#pragma unroll
      for (int p8 = 0; p8 < GP; p8++) {
        float gvv[8];
        frag_unpack(gf[u][p8], gvv);
#pragma unroll
        for (int cc = 0; cc < 8; cc++) {
          const int c = p8 * 8 + cc;
          if (c < COUT) {
            accb[c] += gvv[cc];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[c][j] += gvv[cc] * t[j];
          }
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < COUT; c++) {
    for (int o = 32; o >= CPR; o >>= 1) {
      accb[c] += __shfl_xor(accb[c], o, 64);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[c][j] += __shfl_xor(acc[c][j], o, 64);
    }
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < 8; j++) sm[wv][c][chunk * 8 + j] = acc[c][j];
      if (lane == 0) sm[wv][c][64] = accb[c];
    }
  }
  __syncthreads();
  const int E = Cout * (Cin + 1);          // [co][ci] then [co] bias sums
  for (int e = threadIdx.x; e < E; e += 256) {
    int c = e < Cout * Cin ? e / Cin : e - Cout * Cin, i = e < Cout * Cin ? e - c * Cin : 64;
    ws[(int64_t)blockIdx.x * E + e] = sm[0][c][i] + sm[1][c][i] + sm[2][c][i] + sm[3][c][i];
  }
}
__global__ void __launch_bounds__(64) k_pointwise_wgrad_finish(const float* __restrict__ ws, int nblk, int E, int Cin, int Cout, float* __restrict__ dw, int s_co, float* __restrict__ db) {
  // one wave per output element: lanes stride over the per-block partials (fixed order => deterministic), then a shuffle tree
  const int e = blockIdx.x;
  float a = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) a += ws[(int64_t)b * E + e];
  a = wave_sum(a);
  if (threadIdx.x != 0) return;
  if (e < Cout * Cin) { int c = e / Cin; dw[(int64_t)c * s_co + (e - c * Cin)] = a; }
  else if (db) db[e - Cout * Cin] = a;
}
static int pw_wgrad_grid(int64_t rows, int Cin) { int cpr = (Cin + 7) / 8, rpi = 256 / cpr; int64_t g = (rows + rpi - 1) / rpi; return (int)(g > 1024 ? 1024 : g); }
extern "C" int64_t dp_pointwise_wgrad_ws_elems(int64_t rows, int Cin, int Cout) {
  int cpr = (Cin + 7) / 8;
  if (Cin < 1 || Cin > 64 || (cpr & (cpr - 1)) || Cout < 1 || Cout > 16) return 0;
  return (int64_t)pw_wgrad_grid(rows, Cin) * Cout * (Cin + 1);
}
extern "C" int dp_pointwise_wgrad_rows(const void* x, int ldx, const void* gy, int ldgy, float* dw, int s_co, float* db, float* ws, int64_t rows,
                                       int Cin, int Cout, int dtype, void* stream) {
  int cpr = (Cin + 7) / 8, lg = 0;
  while ((1 << lg) < cpr) lg++;
  if (Cin < 1 || Cin > 64 || (1 << lg) != cpr || Cout < 1 || Cout > 16) DP_FAIL("pointwise_wgrad_rows: needs Cin in {<=8,16,32,64} and Cout <= 16");
  if (!ws) DP_FAIL("pointwise_wgrad_rows: workspace missing");
  int g = pw_wgrad_grid(rows, Cin), E = Cout * (Cin + 1);
#define GO(CO) DP_DISPATCH(dtype, hipLaunchKernelGGL((k_pointwise_wgrad_rows<T, CO>), dim3(g), dim3(256), 0, STREAM, (const T*)x, ldx, (const T*)gy, ldgy, ws, rows, Cin, Cout, lg))
  if (Cout <= 1) GO(1); else if (Cout <= 4) GO(4); else if (Cout <= 8) GO(8); else GO(16);
#undef GO
  hipLaunchKernelGGL(k_pointwise_wgrad_finish, dim3(E), dim3(64), 0, STREAM, (const float*)ws, g, E, Cin, Cout, dw, s_co, db);
  DP_CHECK_LAUNCH("pointwise_wgrad_rows"); return 0;
}

// Weight gradient of the skinny row GEMMs on the matrix cores (round 4): dW[co][ci] = sum_v gy[v][co] x[v][ci] over millions of voxel rows
// -- the 1x1x1 mixers at the 128^3 / 64^3 levels (up to 32 output and 64 input channels).  The tiled weight-gradient kernel
// (k_wgrad_tiled<T,1,1,1>: 64-voxel tiles, block-wide barriers) read these operands at 2.1-2.4 TB/s.  Here every WAVE runs its own
// pipeline over 32-row slabs: 16-byte global loads of the slab after next are in flight while the current one goes through a
// wave-private LDS image (rows as they lie in memory) and comes back as k-major MFMA fragments (ds_read_b64_tr_b16, the map of
// gemm_tn_tile); no block barrier until the end, where the four waves' accumulators meet in LDS and the block writes ONE partial
// [Cout][Cin] to the workspace.  A second small kernel adds the partials in a fixed order (deterministic), writes dW with the caller's
// strides and hands the workspace back zeroed.
// S = the type the rows have in memory: T, or float (the fp32x3 mode's one-product weight gradient, DP_X1: fp32 rows rounded to bf16
// between the global load and the LDS image -- dW = x_hi gy_hi without a cast pass over either tensor).
template <typename S> struct RowPiece { v4u v; };
template <> struct RowPiece<float> { v4f a, b; };
template <typename T, typename S> __device__ __forceinline__ RowPiece<S> row_piece_load(const S* p, bool ok) {
  RowPiece<S> r;
  if constexpr (sizeof(S) == 4) { const v4f a = *(const v4f*)p, b = *(const v4f*)(p + 4); r.a = ok ? a : (v4f){0.f, 0.f, 0.f, 0.f}; r.b = ok ? b : (v4f){0.f, 0.f, 0.f, 0.f}; }
  else { const v4u v = *(const v4u*)p; r.v = ok ? v : (v4u){0, 0, 0, 0}; }
  return r;
}
template <typename T, typename S> __device__ __forceinline__ v4u row_piece_bits(const RowPiece<S>& r) {
  if constexpr (sizeof(S) == 4) {
    return (v4u){(unsigned)f2bf(r.a[0]) | ((unsigned)f2bf(r.a[1]) << 16), (unsigned)f2bf(r.a[2]) | ((unsigned)f2bf(r.a[3]) << 16),
                 (unsigned)f2bf(r.b[0]) | ((unsigned)f2bf(r.b[1]) << 16), (unsigned)f2bf(r.b[2]) | ((unsigned)f2bf(r.b[3]) << 16)};
  } else return r.v;
}
template <typename T, int MT, int NT, typename S = T>
__global__ void __launch_bounds__(256) k_wgrad_rows(const S* __restrict__ x, int ldx, const S* __restrict__ gy, int ldgy, float* __restrict__ part,
                                                    int64_t rows, int Cin, int Cout) {
  constexpr int GP = MT * 16 + 8, XP = NT * 16 + 8;            // LDS row pitches in elements (16-byte rows)
  constexpr int WSZ = 32 * GP + 32 * XP;
  __shared__ __attribute__((aligned(16))) T lds[4 * WSZ];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, i16 = lane & 15;
  T* gimg = lds + wv * WSZ; T* ximg = gimg + 32 * GP;
  // 16-byte pieces of a slab: gy 32 rows x 2 MT pieces, x 32 rows x 2 NT pieces; piece p = lane + 64 i
  int grow[MT], gcc[MT], xrow[NT], xcc[NT];
#pragma unroll
  for (int i = 0; i < MT; i++) { const int p = lane + 64 * i; grow[i] = p / (2 * MT); gcc[i] = (p % (2 * MT)) * 8; }
#pragma unroll
  for (int i = 0; i < NT; i++) { const int p = lane + 64 * i; xrow[i] = p / (2 * NT); xcc[i] = (p % (2 * NT)) * 8; }
  const int64_t nslab = (rows + 31) / 32, wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  v4f acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < NT; n++) acc[m][n] = (v4f){0.f, 0.f, 0.f, 0.f};
  RowPiece<S> rg0[MT], rx0[NT], rg1[MT], rx1[NT];              // two slabs in flight per wave
  auto gload = [&](int64_t sl, RowPiece<S>* rg, RowPiece<S>* rx) __attribute__((always_inline)) {
    const int64_t r0 = sl * 32;
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int64_t r = r0 + grow[i];
      const bool ok = r < rows && gcc[i] < Cout;                // (columns beyond Cout: zero rows of dW that are never stored)
      rg[i] = row_piece_load<T, S>(gy + (ok ? r * ldgy + gcc[i] : 0), ok);
    }
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int64_t r = r0 + xrow[i];
      const bool ok = r < rows && xcc[i] < Cin;
      rx[i] = row_piece_load<T, S>(x + (ok ? r * ldx + xcc[i] : 0), ok);
    }
  };
  const int trg = (8 * q + (i16 >> 2)) * GP + 4 * (i16 & 3), trx = (8 * q + (i16 >> 2)) * XP + 4 * (i16 & 3);
  auto step = [&](int64_t nxt, RowPiece<S>* rg, RowPiece<S>* rx) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++) *(v4u*)(gimg + grow[i] * GP + gcc[i]) = row_piece_bits<T, S>(rg[i]);
#pragma unroll
    for (int i = 0; i < NT; i++) *(v4u*)(ximg + xrow[i] * XP + xcc[i]) = row_piece_bits<T, S>(rx[i]);
    if (nxt < nslab) gload(nxt, rg, rx);                        // this register set's next slab flies during the LDS round trip and the MFMAs
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
    Frag8<T> fa[MT], fb[NT];
#pragma unroll
    for (int m = 0; m < MT; m++) fa[m] = tr_pair<4 * GP, T>(gimg + m * 16 + trg);
#pragma unroll
    for (int n = 0; n < NT; n++) fb[n] = tr_pair<4 * XP, T>(ximg + n * 16 + trx);
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int n = 0; n < NT; n++) acc[m][n] = mma16(fa[m], fb[n], acc[m][n]);
    __builtin_amdgcn_wave_barrier();
  };
  if (wave < nslab) gload(wave, rg0, rx0);
  if (wave + nwaves < nslab) gload(wave + nwaves, rg1, rx1);
  for (int64_t sl = wave; sl < nslab; sl += 2 * nwaves) {
    step(sl + 2 * nwaves, rg0, rx0);
    if (sl + nwaves < nslab) step(sl + 3 * nwaves, rg1, rx1);
  }
  // the four waves' tiles meet in LDS (the images are dead): C layout col (ci) = lane & 15, row (co) = 4 (lane >> 4) + e
  __syncthreads();
  float* red = (float*)lds;                                    // [wave][MT*16][NT*16] floats: 4 * MT * NT * 256 * 4 B <= the image space?
  constexpr int TILE = MT * 16 * NT * 16;
  static_assert(4 * TILE * 4 <= 4 * WSZ * (int)sizeof(T) || TILE * 4 <= 4 * WSZ * (int)sizeof(T), "reduction space");
  constexpr bool ALL4 = 4 * TILE * 4 <= 4 * WSZ * (int)sizeof(T);
  if constexpr (ALL4) {
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) red[wv * TILE + (m * 16 + 4 * q + e) * (NT * 16) + n * 16 + i16] = acc[m][n][e];
    __syncthreads();
    float* dst = part + (int64_t)blockIdx.x * Cout * Cin;
    for (int o = tid; o < Cout * Cin; o += 256) {
      const int co = o / Cin, ci = o - co * Cin, a = co * (NT * 16) + ci;
      dst[o] = red[a] + red[TILE + a] + red[2 * TILE + a] + red[3 * TILE + a];
    }
  } else {
    // large tiles: the waves take turns adding into one tile
    for (int w = 0; w < 4; w++) {
      if (wv == w) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int n = 0; n < NT; n++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              float* p_ = red + (m * 16 + 4 * q + e) * (NT * 16) + n * 16 + i16;
              *p_ = (w ? *p_ : 0.f) + acc[m][n][e];
            }
      }
      __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * Cout * Cin;
    for (int o = tid; o < Cout * Cin; o += 256) { const int co = o / Cin, ci = o - co * Cin; dst[o] = red[co * (NT * 16) + ci]; }
  }
}
__global__ void __launch_bounds__(64) k_wgrad_rows_finish(float* __restrict__ part, int nblk, int E, int Cin, float* __restrict__ dw, int64_t s_co, int64_t s_ci,
                                                          int rezero) {
  const int e = blockIdx.x;
  float a = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) { a += part[(int64_t)b * E + e]; if (rezero) part[(int64_t)b * E + e] = 0.f; }
  a = wave_sum(a);
  if (threadIdx.x == 0) { const int co = e / Cin, ci = e - co * Cin; dw[co * s_co + ci * s_ci] = a; }
}
static int wgrad_rows_blocks(int64_t rows) {
  static const int cap = [] { const char* e = getenv("DP_WGRAD_ROWS_BLOCKS"); return e ? (atoi(e) < 1024 ? atoi(e) : 1024) : 256; }();      // (one block per CU: 1024 blocks 0.120 ms, 512 / 256 blocks 0.105 for 32 -> 16 at 2 x 128^3 -- the finish pass reads fewer partials)
  const int64_t slabs = (rows + 31) / 32; int64_t g = slabs / 16; return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}
// shapes the streaming weight-gradient kernel takes (16-bit storage; alignment of the pointers is checked by the caller)
bool wgrad_rows_ok(int ldx, int ldgy, int64_t rows, int Cin, int Cout, int dtype) {
  static const int on = [] { const char* e = getenv("DP_WGRAD_ROWS"); return e ? atoi(e) : 1; }();
  if (dtype == DP_X1)        // fp32 rows, rounded to bf16 in registers (the fp32x3 mode's one-product weight gradients): 16-byte pieces of four floats
    return on && rows >= 32768 && Cin >= 8 && Cin <= 64 && Cout >= 8 && Cout <= 32 && Cout % 8 == 0 && ldx % 4 == 0 && ldgy % 4 == 0 &&
           (Cin + 7) / 8 * 8 <= ldx && Cout <= ldgy;
  return on && (dtype == DP_BF16 || dtype == DP_F16) && rows >= 32768 && Cin >= 8 && Cin <= 64 && Cout >= 8 && Cout <= 32 && Cout % 8 == 0 &&
         ldx % 8 == 0 && ldgy % 8 == 0 && (Cin + 7) / 8 * 8 <= ldx && Cout <= ldgy;      // (wider outputs -- ConvTranspose's 8 Cout -- measured slower than the tiled kernel: 16 MB of partials)
}
int64_t wgrad_rows_ws_elems(int Cin, int Cout) { return (int64_t)1024 * Cin * Cout; }
// dw[co * s_co + ci * s_ci] = sum_v gy[v][co] x[v][ci]; part: wgrad_rows_ws_elems() floats (zeroed on return when `rezero`)
int wgrad_rows_launch(const void* x, int ldx, const void* gy, int ldgy, float* dw, int64_t s_co, int64_t s_ci, float* part, int64_t rows, int Cin, int Cout,
                      int dtype, int rezero, hipStream_t s) {
  const int mt = (Cout + 15) / 16, nt = (Cin + 15) / 16, g = wgrad_rows_blocks(rows);
#define GOW(TT, MT_, NT_) hipLaunchKernelGGL((k_wgrad_rows<TT, MT_, NT_>), dim3(g), dim3(256), 0, s, (const TT*)x, ldx, (const TT*)gy, ldgy, part, rows, Cin, Cout)
#define GOWN(TT, MT_) do { if (nt == 1) GOW(TT, MT_, 1); else if (nt == 2) GOW(TT, MT_, 2); else GOW(TT, MT_, 4); } while (0)
#define GOWM(TT) do { if (mt == 1) GOWN(TT, 1); else GOWN(TT, 2); } while (0)
#define GOW32(MT_, NT_) hipLaunchKernelGGL((k_wgrad_rows<bf16_t, MT_, NT_, float>), dim3(g), dim3(256), 0, s, (const float*)x, ldx, (const float*)gy, ldgy, part, rows, Cin, Cout)
#define GOWN32(MT_) do { if (nt == 1) GOW32(MT_, 1); else if (nt == 2) GOW32(MT_, 2); else GOW32(MT_, 4); } while (0)
  if (dtype == DP_X1) { if (mt == 1) GOWN32(1); else GOWN32(2); }
  else if (dtype == DP_BF16) GOWM(bf16_t); else GOWM(f16_t);
#undef GOWN32
#undef GOW32
#undef GOWM
#undef GOWN
#undef GOW
  hipLaunchKernelGGL(k_wgrad_rows_finish, dim3(Cout * Cin), dim3(64), 0, s, part, g, Cout * Cin, Cin, dw, s_co, s_ci, rezero);
  return 0;
}

// ------------------------------------------------------------------------------------------------ sliding-window stitching
// MONAI sliding_window_inference (constant blend mode), call site train_light_linked_model.py:152-153: window predictions are
// summed into an fp32 volume with a per-voxel visit count, then divided.  One thread per (window voxel, 8-channel chunk).
template <typename T>
__global__ void k_window_accumulate(const T* __restrict__ win, int ldw, float* __restrict__ acc, float* __restrict__ cnt, int n, int D, int H, int W,
                                    int rz, int ry, int rx, int z0, int y0, int x0, int C) {
  const int cg8 = (C + 7) >> 3;
  const int64_t total = (int64_t)rz * ry * rx * cg8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg8); int64_t v = i / cg8;
    const int x = (int)(v % rx); v /= rx; const int y = (int)(v % ry); const int z = (int)(v / ry);
    const int64_t src = ((int64_t)(z * ry + y) * rx + x), dst = (((int64_t)n * D + z0 + z) * H + y0 + y) * W + x0 + x;
    const int nv = min(8, C - cg * 8);
    for (int k = 0; k < nv; k++) acc[dst * C + cg * 8 + k] += ld_f(win + src * ldw + cg * 8 + k);
    if (cg == 0) cnt[dst] += 1.f;
  }
}
template <typename T>
__global__ void k_window_normalize(const float* __restrict__ acc, const float* __restrict__ cnt, T* __restrict__ out, int ldo, int64_t rows, int C) {
  const int64_t total = rows * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C; const int c = (int)(i - r * C);
    st_f(out + r * ldo + c, acc[i] / cnt[r]);
  }
}
extern "C" int dp_window_accumulate(const void* win, int ldw, float* acc, float* cnt, int n, int D, int H, int W, int rz, int ry, int rx,
                                    int z0, int y0, int x0, int C, int dtype, void* stream) {
  if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + rz > D || y0 + ry > H || x0 + rx > W) DP_FAIL("window_accumulate: window outside the volume");
  int64_t total = (int64_t)rz * ry * rx * ((C + 7) / 8);
  int g = grid_for(total, 256, 256 * 64);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_window_accumulate<T>, dim3(g), dim3(256), 0, STREAM, (const T*)win, ldw, acc, cnt, n, D, H, W, rz, ry, rx, z0, y0, x0, C));
  DP_CHECK_LAUNCH("window_accumulate"); return 0;
}
extern "C" int dp_window_normalize(const float* acc, const float* cnt, void* out, int ldo, int64_t rows, int C, int dtype, void* stream) {
  int g = grid_for(rows * C, 256, 256 * 64);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_window_normalize<T>, dim3(g), dim3(256), 0, STREAM, acc, cnt, (T*)out, ldo, rows, C));
  DP_CHECK_LAUNCH("window_normalize"); return 0;
}

// ------------------------------------------------------------------------------------------------ im2col for small volumes
// col[row][tap * CinP + c] = x[n, od*s + kd*dil - pad, ...][c] (zero outside / for c >= Cin).  A few thousand output voxels
// with hundreds of channels (the deepest C3D stages: 256 -> 256 at 8^3, stride-2 128 -> 256 at 16^3 -> 8^3) are a plain
// GEMM with K = taps * Cin once gathered: the gathered matrix is a few MB, the generic per-tap gather kernel spent 0.2-0.4 ms
// there.  One 16-byte chunk per thread.
template <typename T>
__global__ void k_im2col3d(const T* __restrict__ x, int ldx, T* __restrict__ col, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cin, int CinP,
                           int k, int stride, int pad, int dil) {
  const int cg8 = CinP >> 3, taps = k * k * k;
  const int64_t total = (int64_t)N * Do * Ho * Wo * taps * cg8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg8); int64_t t = i / cg8;
    const int tap = (int)(t % taps); int64_t row = t / taps;
    const int kw = tap % k, kh = (tap / k) % k, kd = tap / (k * k);
    int64_t v = row; const int ow = (int)(v % Wo); v /= Wo; const int oh = (int)(v % Ho); v /= Ho; const int od = (int)(v % Do); const int n = (int)(v / Do);
    const int id = od * stride + kd * dil - pad, ih = oh * stride + kh * dil - pad, iw = ow * stride + kw * dil - pad;
    const int nv = min(8, Cin - cg * 8);
    const bool ok = id >= 0 && id < Di && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi && nv > 0;
    Frag8<T> f = ok ? frag_load(x + ((((int64_t)n * Di + id) * Hi + ih) * Wi + iw) * ldx + cg * 8, nv) : frag_zero<T>();
    frag_store<T>(col + (row * taps + tap) * CinP + cg * 8, f, 8);
  }
}
extern "C" int dp_im2col3d(const void* x, int ldx, void* col, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cin, int k, int stride, int pad,
                           int dil, int dtype, void* stream) {
  const int CinP = (Cin + 7) & ~7;
  int64_t total = (int64_t)N * Do * Ho * Wo * k * k * k * (CinP / 8);
  int g = grid_for(total, 256, 256 * 64);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_im2col3d<T>, dim3(g), dim3(256), 0, STREAM, (const T*)x, ldx, (T*)col, N, Di, Hi, Wi, Do, Ho, Wo, Cin, CinP, k,
                                        stride, pad, dil));
  DP_CHECK_LAUNCH("im2col3d"); return 0;
}

