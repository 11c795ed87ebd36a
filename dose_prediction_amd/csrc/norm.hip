// Normalisation kernels: InstanceNorm3d / BatchNorm3d (two-phase: per-block partial sums -> fp64 finalize ->
// fused normalise + affine + residual + activation) with their backward, and LayerNorm rows.
// All are HBM-bound row streams over NDHWC voxel rows: each thread owns one 8-channel (16 B bf16) chunk of a row,
// keeps its per-channel constants in registers and walks the block's voxel range.
#include "common.h"
#include <stdlib.h>
#include <mutex>

#define STREAM ((hipStream_t)stream)
#define NT 256
// Voxel rows per block of the row-stream kernels: 2048 for big volumes (4096 is 2-4 % faster alone on the chip, nothing in the step), fewer for small ones so that even an 8^3 / 16^3
// feature map spreads over a few hundred blocks (a 4-block launch over 768 channels took ~190 us).
static inline int rows_per_block(int64_t V) {
  static const int max_rpb = getenv("DP_NORM_RPB") ? atoi(getenv("DP_NORM_RPB")) : 2048;      // (experiment switch)
  int64_t r = V / 128;
  if (r < 32) r = 32;
  if (r > max_rpb) r = max_rpb;
  return (int)r;
}

template <typename T> __device__ __forceinline__ void unpack8(const T* p, int nv, float* o);
template <> __device__ __forceinline__ void unpack8<float>(const float* p, int nv, float* o) {
  Frag8<float> f = frag_load(p, nv); for (int i = 0; i < 8; i++) o[i] = f.v[i];
}
template <> __device__ __forceinline__ void unpack8<bf16_t>(const bf16_t* p, int nv, float* o) {
  Frag8<bf16_t> f = frag_load(p, nv);
  for (int i = 0; i < 4; i++) { o[2 * i] = __uint_as_float(f.u[i] << 16); o[2 * i + 1] = __uint_as_float(f.u[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack8<f16_t>(const f16_t* p, int nv, float* o) {
  Frag8<f16_t> f = frag_load(p, nv);
  const v8h h = __builtin_bit_cast(v8h, f.u);
  for (int i = 0; i < 8; i++) o[i] = (float)h[i];
}
__device__ __forceinline__ void pack8(f16_t* p, int nv, const float* o) {
  if (nv >= 8 && ((uintptr_t)p & 15) == 0) { v8h h; for (int i = 0; i < 8; i++) h[i] = (f16_t)o[i]; *(v4u*)p = __builtin_bit_cast(v4u, h); }
  else for (int i = 0; i < 8; i++) if (i < nv) p[i] = (f16_t)o[i];
}
__device__ __forceinline__ void pack8(float* p, int nv, const float* o) {
  if (nv >= 8 && ((uintptr_t)p & 15) == 0) { *(v4f*)p = (v4f){o[0], o[1], o[2], o[3]}; *(v4f*)(p + 4) = (v4f){o[4], o[5], o[6], o[7]}; }
  else for (int i = 0; i < 8; i++) if (i < nv) p[i] = o[i];
}
__device__ __forceinline__ void pack8(bf16_t* p, int nv, const float* o) {
  if (nv >= 8 && ((uintptr_t)p & 15) == 0) {
    v4u u; for (int i = 0; i < 4; i++) u[i] = (unsigned)f2bf(o[2 * i]) | ((unsigned)f2bf(o[2 * i + 1]) << 16);
    *(v4u*)p = u;
  } else for (int i = 0; i < 8; i++) if (i < nv) p[i] = f2bf(o[i]);
}

// unguarded 8-element row-chunk access for the fast paths (C % 8 == 0, pitches % 8 == 0, 16-byte aligned bases)
__device__ __forceinline__ void ld8(const float* p, float* o) { v4f a = *(const v4f*)p, b = *(const v4f*)(p + 4); o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3]; }
__device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
  v4u u = *(const v4u*)p;
#pragma unroll
  for (int i = 0; i < 4; i++) { o[2 * i] = __uint_as_float(u[i] << 16); o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
__device__ __forceinline__ void ld8(const f16_t* p, float* o) {
  const v8h h = __builtin_bit_cast(v8h, *(const v4u*)p);
#pragma unroll
  for (int i = 0; i < 8; i++) o[i] = (float)h[i];
}
__device__ __forceinline__ void st8(f16_t* p, const float* o) {
  v8h h;
#pragma unroll
  for (int i = 0; i < 8; i++) h[i] = (f16_t)o[i];
  *(v4u*)p = __builtin_bit_cast(v4u, h);
}
__device__ __forceinline__ void st8(float* p, const float* o) { *(v4f*)p = (v4f){o[0], o[1], o[2], o[3]}; *(v4f*)(p + 4) = (v4f){o[4], o[5], o[6], o[7]}; }
__device__ __forceinline__ void st8(bf16_t* p, const float* o) {
  v4u u;
#pragma unroll
  for (int i = 0; i < 4; i++) u[i] = (unsigned)f2bf(o[2 * i]) | ((unsigned)f2bf(o[2 * i + 1]) << 16);
  *(v4u*)p = u;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
#define RU 4     // rows in flight per thread in the fast paths

extern "C" int dp_stats_nblk(int64_t V) { int r = rows_per_block(V); return (int)((V + r - 1) / r); }

// Common geometry: block b of sample n covers voxels [b*RPB, min(V,(b+1)*RPB)); thread t owns chunk cg = t % cg8 and
// rows r0 + k*rpi.  (cg8 = chunks per row, rpi = rows per iteration = NT / cg8.)
struct RowGeom { int cg8, rpi, cg, r0, nv; bool active; };
__device__ __forceinline__ RowGeom row_geom(int C) {
  RowGeom g; g.cg8 = (C + 7) >> 3; g.rpi = NT / g.cg8; if (g.rpi < 1) g.rpi = 1;
  g.cg = threadIdx.x % g.cg8; g.r0 = threadIdx.x / g.cg8; g.active = g.r0 < g.rpi; g.nv = min(8, C - g.cg * 8);
  return g;
}

// Deterministic block reduction of per-thread 16 partials (acc[0..7], acc2[0..7]) over the threads that share a chunk.
// red: LDS float [NT][16].  Result written by threads c < 2*C... to part[(2)][C].
// coh: the row goes out with device-scope (sc1, written through the XCD's L2) stores, for a reader in ANOTHER block of the same launch
// (ticket_is_last below)
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void block_reduce_store(float* red, const float* a1, const float* a2, const RowGeom& g, int C, float* part, bool coh = false) {
  for (int i = 0; i < 8; i++) { red[threadIdx.x * 16 + i] = g.active ? a1[i] : 0.f; red[threadIdx.x * 16 + 8 + i] = g.active ? a2[i] : 0.f; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += NT) {
    int which = o / C, c = o - which * C, cg = c >> 3, j = c & 7;
    float s = 0.f;
    for (int r = 0; r < g.rpi; r++) s += red[(r * g.cg8 + cg) * 16 + which * 8 + j];
    if (coh) st_coh(part + which * C + c, s); else part[which * C + c] = s;
  }
}

// Parallel fp64 combine of the per-block partials: one 256-thread block per (statistics group, 16/32-channel slice);
// threads are arranged [rows][channels] so every partial row is read coalesced, then the rows are tree-reduced in LDS.
// COH = 1: the rows were written by other blocks of the SAME launch (st_coh): device-scope loads that do not trust this XCD's L2.
template <int COH>
__device__ __forceinline__ void combine_partials(const float* __restrict__ part, int n0, int n1, int nblk, int C, int c0, int cpb,
                                                 double& s1, double& s2, double* red) {
  const int t = threadIdx.x, cc = t % cpb, r = t / cpb, rows = 256 / cpb, c = c0 + cc;
  double a1 = 0, a2 = 0;
  if (c < C) {
    const int64_t total = (int64_t)(n1 - n0) * nblk;
    const float* p0 = part + (int64_t)n0 * nblk * 2 * C + c;
    int64_t b = r;
    // 8 partial rows per trip with all 16 loads issued before the first add: one thread walks up to 64 rows, and a load -> add
    // chain paid one L2 round trip per row (these 10-us launches were almost pure latency)
    for (; b + 7 * (int64_t)rows < total; b += 8 * (int64_t)rows) {
      float u[8], v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float* p = p0 + (b + k * (int64_t)rows) * 2 * C;
        if constexpr (COH) { u[k] = ld_coh(p); v[k] = ld_coh(p + C); } else { u[k] = p[0]; v[k] = p[C]; }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) { a1 += u[k]; a2 += v[k]; }
    }
    for (; b < total; b += rows) {
      const float* p = p0 + b * 2 * C;
      if constexpr (COH) { a1 += ld_coh(p); a2 += ld_coh(p + C); } else { a1 += p[0]; a2 += p[C]; }
    }
  }
  red[t * 2] = a1; red[t * 2 + 1] = a2;
  __syncthreads();
  for (int st = rows / 2; st > 0; st >>= 1) {
    if (r < st) { red[t * 2] += red[(t + st * cpb) * 2]; red[t * 2 + 1] += red[(t + st * cpb) * 2 + 1]; }
    __syncthreads();
  }
  s1 = red[cc * 2]; s2 = red[cc * 2 + 1];
}

// One unit of the statistics finalize: (sum, sum of squares) partial rows of statistics group gidx, channels [bx * cpb, +cpb) -> mean / rstd
// (+ running statistics, + the folded scale / shift).  Called by a whole 256-thread block (block-uniform arguments).
struct StatsFin {
  const float* part; int N, nblk, C; int64_t V; int batch_mode; float eps; float* mean; float* rstd; float* rmean; float* rvar; float momentum;
  int cpb; const float* gamma; const float* beta; float* sc; float* sh; int cpad;
};
template <int COH>
__device__ __forceinline__ void stats_finalize_unit(const StatsFin& f, int bx, int gidx, double* red) {
  const int c0 = bx * f.cpb;
  const int n0 = f.batch_mode ? 0 : gidx, n1 = f.batch_mode ? f.N : gidx + 1;
  double s1, s2;
  combine_partials<COH>(f.part, n0, n1, f.nblk, f.C, c0, f.cpb, s1, s2, red);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x >= f.cpb || c >= f.C) return;
  double cnt = (double)f.V * (n1 - n0);
  double m = s1 / cnt, var = s2 / cnt - m * m;
  if (var < 0) var = 0;
  f.mean[gidx * f.C + c] = (float)m; f.rstd[gidx * f.C + c] = (float)(1.0 / sqrt(var + (double)f.eps));
  if (f.sc) {   // z = x * sc + sh, the form the fused normalise kernels and the convolution prologue evaluate (float arithmetic as in k_norm_act_fwd)
    const float r_ = f.rstd[gidx * f.C + c], g_ = f.gamma ? f.gamma[c] : 1.f, b_ = f.beta ? f.beta[c] : 0.f, s_ = r_ * g_;
    f.sc[gidx * f.cpad + c] = s_; f.sh[gidx * f.cpad + c] = b_ - f.mean[gidx * f.C + c] * s_;
  }
  if (f.rmean && f.batch_mode) {
    double unb = cnt > 1 ? var * cnt / (cnt - 1) : var;
    f.rmean[c] = (float)((1.0 - f.momentum) * f.rmean[c] + f.momentum * m);
    f.rvar[c] = (float)((1.0 - f.momentum) * f.rvar[c] + f.momentum * unb);
  }
}
__global__ void __launch_bounds__(256) k_stats_finalize(StatsFin f) {
  __shared__ double red[512];
  stats_finalize_unit<0>(f, blockIdx.x, blockIdx.y, red);
}

// ---------------------------------------------------------------------------- finalize folded into the partial pass ("last block" ticket)
// The partial-sum kernels below are pure read streams that end in one small store per block.  Instead of a second launch that combines
// the rows (5-7 us alone on the chip, 10-20 us of the dependent chain inside the step, where the launch queues behind the blocks of the
// other streams), every block of a statistics group takes a ticket after its row is written (device-scope stores, acknowledged, then a
// device-scope atomicAdd) and the block that draws the last one runs the finalize units of that group itself, reading the rows with
// device-scope loads: same code, same order of additions, bit-identical results.  One counter per statistics group (instance mode: the samples finish independently) from
// a ring of zeroed counters; the last block puts the zero back.  Off with DP_NO_TICKET=1 (the callers then launch the finalize).
#define TICKET_RING 8192
#define TICKET_CAPTURE 4096     // counters handed to launches that are being CAPTURED into a HIP graph: never recycled (see ticket_counters)
static unsigned* g_ticket_ring[32] = {};
static unsigned g_ticket_next[32] = {};
static unsigned g_ticket_cap_next[32] = {};
static std::mutex g_ticket_mutex;          // (forward and backward may be driven from different host threads)
// Counters for one launch.  Eager launches rotate through a ring of TICKET_RING zeroed counters (a slot comes back after ~8 000 later
// hand-outs, i.e. dozens of steps: its launch has long retired and re-armed it).  A launch that is being CAPTURED keeps its slot for as
// long as the graph lives and runs it again on every replay, so it must never share a slot with a later eager launch (two launches
// drawing tickets from one counter at the same time would both miscount: ADVICE r5): captured launches get counters from a separate
// region that is only ever bumped, and the two-launch form (nullptr) when that region is used up.
static unsigned* ticket_counters(int n, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32 || n > TICKET_RING) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  const bool capturing = cs != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(g_ticket_mutex);
  if (!g_ticket_ring[dev]) {
    // (first use inside a stream capture: no allocation / memset / device synchronisation there -- the caller takes the two-launch form)
    if (capturing) return nullptr;
    unsigned* p = nullptr;
    if (hipMalloc(&p, (TICKET_RING + TICKET_CAPTURE) * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(p, 0, (TICKET_RING + TICKET_CAPTURE) * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError(); (void)hipFree(p); return nullptr;
    }
    g_ticket_ring[dev] = p;
  }
  if (capturing) {
    unsigned at = g_ticket_cap_next[dev];
    if (at + n > TICKET_CAPTURE) return nullptr;
    g_ticket_cap_next[dev] = at + n;
    return g_ticket_ring[dev] + TICKET_RING + at;
  }
  unsigned at = g_ticket_next[dev];
  if (at + n > TICKET_RING) at = 0;
  g_ticket_next[dev] = at + n;
  return g_ticket_ring[dev] + at;
}
static inline bool ticket_enabled() { static const bool off = getenv("DP_NO_TICKET") != nullptr; return !off; }
extern "C" int dp_ticket_enabled(void) { return ticket_enabled() ? 1 : 0; }
// true in exactly one block per counter: the one whose ticket is the last of `total`.  Every thread of the block must call it, after the
// block's own row stores (st_coh).  NO agent-scope release / acquire fence: on this part that is `buffer_wbl2 sc1` + `buffer_inv sc1`
// -- a write-back and an invalidate of the XCD's whole L2, i.e. of the convolutions running beside this kernel on the other streams
// (measured with fences: the 23.4-ms step took 28.8 ms).  Instead the rows travel as agent-scope atomic stores / loads (sc1: written
// through and read past the L2), and the ORDER row stores -> ticket -> row loads is made explicit at both levels:
//   * compiler: a workgroup-scope release fence after the row stores and a workgroup-scope acquire fence after the ticket is known
//     (atomic accesses may not be moved across them; on gfx950 they lower to s_waitcnt only -- no L2 maintenance -- which is checked
//     in the ISA by tests/test_cabi_cpu.py::test_ticket_fences_do_not_touch_the_l2);
//   * hardware: s_waitcnt vmcnt(0) -- every row store of this thread has been ACKNOWLEDGED by the memory system (sc1 stores are
//     acknowledged at device coherence) -- then the block barrier, then thread 0's agent-scope fetch_add; the reader's loads are issued
//     after its own fetch_add returned the last ticket, which every writer's fetch_add (and hence its acknowledged stores) precedes.
__device__ __forceinline__ bool ticket_is_last(unsigned* counter, unsigned total, int* flag) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);           // this thread's row stores have been acknowledged ...
  asm volatile("" ::: "memory");           // (... and nothing that touches memory is scheduled across this point)
  __syncthreads();                         // ... for every thread of the block, before the ticket is drawn
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == total - 1) ? 1 : 0;
    if (t == total - 1) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm: the next user is a later launch
  }
  __syncthreads();
  const bool last = *flag != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  asm volatile("" ::: "memory");
  return last;
}

static inline int pick_cpb(int C) { return C >= 32 ? 32 : (C >= 16 ? 16 : 8); }
static inline StatsFin stats_fin(const float* part, int N, int nblk, int C, int64_t V, int batch_mode, float eps, float* mean, float* rstd,
                                 float* running_mean, float* running_var, float momentum) {
  StatsFin f = {}; f.part = part; f.N = N; f.nblk = nblk; f.C = C; f.V = V; f.batch_mode = batch_mode; f.eps = eps; f.mean = mean; f.rstd = rstd;
  f.rmean = running_mean; f.rvar = running_var; f.momentum = momentum; f.cpb = pick_cpb(C);
  return f;
}
extern "C" int dp_stats_finalize(const float* part, int N, int nblk, int C, int64_t V, int batch_mode, float eps, float* mean, float* rstd,
                                 float* running_mean, float* running_var, float momentum, void* stream) {
  StatsFin f = stats_fin(part, N, nblk, C, V, batch_mode, eps, mean, rstd, running_mean, running_var, momentum);
  hipLaunchKernelGGL(k_stats_finalize, dim3(cdiv(C, f.cpb), batch_mode ? 1 : N), dim3(256), 0, STREAM, f);
  DP_CHECK_LAUNCH("stats_finalize"); return 0;
}

template <typename T>
__global__ void __launch_bounds__(NT) k_stats_partial(const T* __restrict__ x, int ld, int64_t V, int C, float* __restrict__ part, int nblk, int ROWS_PER_BLOCK, int fast,
                                                      unsigned* ticket, StatsFin f) {
  __shared__ __align__(16) float red[NT * 16];
  __shared__ int last_flag;
  if (C > 8 * NT) return;
  int b = blockIdx.x, n = blockIdx.y;
  RowGeom g = row_geom(C);
  int64_t v0 = (int64_t)b * ROWS_PER_BLOCK, v1 = min(V, v0 + ROWS_PER_BLOCK);
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t vs = v0 + g.r0;
  if (g.active && fast) {
    const T* xb = x + (int64_t)n * V * ld + g.cg * 8;
    constexpr int RUS = 8;       // a pure read stream: 8 x 16 bytes in flight per thread
    for (; vs + (RUS - 1) * (int64_t)g.rpi < v1; vs += RUS * g.rpi) {
      float t[RUS][8];
#pragma unroll
      for (int u = 0; u < RUS; u++) ld8(xb + (vs + u * g.rpi) * ld, t[u]);
#pragma unroll
      for (int u = 0; u < RUS; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) { s1[i] += t[u][i]; s2[i] += t[u][i] * t[u][i]; }
    }
  }
  if (g.active) for (int64_t v = vs; v < v1; v += g.rpi) {
    float t[8]; unpack8<T>(x + ((int64_t)n * V + v) * ld + g.cg * 8, g.nv, t);
    for (int i = 0; i < 8; i++) { s1[i] += t[i]; s2[i] += t[i] * t[i]; }
  }
  block_reduce_store(red, s1, s2, g, C, part + ((int64_t)n * nblk + b) * 2 * C, ticket != nullptr);
  if (!ticket) return;
  // folded finalize: batch mode = one group over the whole grid, instance mode = one group per sample (blockIdx.y)
  if (!ticket_is_last(ticket + (f.batch_mode ? 0 : n), f.batch_mode ? gridDim.x * gridDim.y : gridDim.x, &last_flag)) return;
  for (int bx = 0; bx * f.cpb < C; bx++) { stats_finalize_unit<1>(f, bx, f.batch_mode ? 0 : n, (double*)red); __syncthreads(); }
}
extern "C" int dp_stats_partial(const void* x, int ld, int N, int64_t V, int C, float* part, int dtype, void* stream) {
  if (C > 8 * NT) DP_FAIL("stats: C too large");
  int nblk = dp_stats_nblk(V);
  int fast = (C % 8 == 0) && (ld % 8 == 0) && aligned16(x);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_stats_partial<T>, dim3(nblk, N), dim3(NT), 0, STREAM, (const T*)x, ld, V, C, part, nblk, rows_per_block(V), fast,
                                        (unsigned*)nullptr, StatsFin{}));
  DP_CHECK_LAUNCH("stats_partial"); return 0;
}
// dp_stats_partial + dp_stats_finalize in ONE launch (the last block of every statistics group finalizes it, see ticket_is_last).
// Returns 3 (nothing launched) when the folded form is switched off or unavailable: the caller then makes the two calls.
extern "C" int dp_stats_partial_finalize(const void* x, int ld, int N, int64_t V, int C, float* part, int batch_mode, float eps, float* mean,
                                         float* rstd, float* running_mean, float* running_var, float momentum, int dtype, void* stream) {
  if (C > 8 * NT) DP_FAIL("stats: C too large");
  if (!ticket_enabled()) return 3;
  unsigned* ticket = ticket_counters(batch_mode ? 1 : N, STREAM);
  if (!ticket) return 3;
  int nblk = dp_stats_nblk(V);
  int fast = (C % 8 == 0) && (ld % 8 == 0) && aligned16(x);
  StatsFin f = stats_fin(part, N, nblk, C, V, batch_mode, eps, mean, rstd, running_mean, running_var, momentum);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_stats_partial<T>, dim3(nblk, N), dim3(NT), 0, STREAM, (const T*)x, ld, V, C, part, nblk, rows_per_block(V), fast,
                                        ticket, f));
  DP_CHECK_LAUNCH("stats_partial_finalize"); return 0;
}


// ---------------------------------------------------------------------------- fused normalise + affine + residual + act
struct NormConst { float m[8], r[8], ga[8], be[8]; };
__device__ __forceinline__ NormConst load_consts(const float* mean, const float* rstd, int sidx, const float* gamma, const float* beta, int c0, int nv) {
  NormConst k;
  for (int i = 0; i < 8; i++) {
    bool ok = i < nv;
    k.m[i] = (ok && mean) ? mean[sidx + c0 + i] : 0.f; k.r[i] = (ok && rstd) ? rstd[sidx + c0 + i] : 1.f;
    k.ga[i] = (ok && gamma) ? gamma[c0 + i] : 1.f; k.be[i] = (ok && beta) ? beta[c0 + i] : 0.f;
  }
  return k;
}

// fp32x3 mode: 8 fp32 values written as their bf16 halves, hi at p, lo at p + cp (the [x_hi | x_lo] operand layout of csrc/x3.hip)
__device__ __forceinline__ void st8_split(bf16_t* p, int cp, const float* o) {
  v4u hi, lo;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const bf16_t h0 = f2bf(o[2 * r]), h1 = f2bf(o[2 * r + 1]);
    const bf16_t l0 = f2bf(o[2 * r] - bf2f(h0)), l1 = f2bf(o[2 * r + 1] - bf2f(h1));
    hi[r] = (unsigned)h0 | ((unsigned)h1 << 16); lo[r] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
  *(v4u*)p = hi; *(v4u*)(p + cp) = lo;
}

struct NormBwdFin { const float* part; int N, nblk, C, batch_mode; float* s1o; float* s2o; float* dgamma; float* dbeta; int cpb; };
// One unit of the backward finalize (statistics group gidx, channels [bx * cpb, +cpb)); called by a whole 256-thread block.
template <int COH>
__device__ __forceinline__ void norm_bwd_finalize_unit(const NormBwdFin& f, int bx, int gidx, double* red) {
  const int c0 = bx * f.cpb;
  const int n0 = f.batch_mode ? 0 : gidx, n1 = f.batch_mode ? f.N : gidx + 1;
  double s1, s2;
  combine_partials<COH>(f.part, n0, n1, f.nblk, f.C, c0, f.cpb, s1, s2, red);
  // dgamma / dbeta are sums over ALL samples.  Batch mode has ONE statistics group: its sums ARE dgamma / dbeta.  Instance mode: the
  // block of sample 0 combines the partial rows of every sample once more, in the same fixed order (round 5: one fp32 atomic per sample
  // used to meet here -- order-dependent for N > 2).  Either way a plain store: the caller need not zero them.
  double t1 = s1, t2 = s2;
  const bool owner = f.batch_mode || gidx == 0;
  if (!f.batch_mode && gidx == 0 && f.N > 1 && (f.dgamma || f.dbeta)) {          // (block-uniform)
    __syncthreads();
    combine_partials<COH>(f.part, 0, f.N, f.nblk, f.C, c0, f.cpb, t1, t2, red);
  }
  const int c = c0 + threadIdx.x;
  if (threadIdx.x >= f.cpb || c >= f.C) return;
  f.s1o[gidx * f.C + c] = (float)s1; f.s2o[gidx * f.C + c] = (float)s2;
  if (f.dgamma && owner) f.dgamma[c] = (float)t2;
  if (f.dbeta && owner) f.dbeta[c] = (float)t1;
}

static inline NormBwdFin norm_bwd_fin(const float* part, int N, int nblk, int C, int batch_mode, float* s1, float* s2, float* dgamma, float* dbeta) {
  NormBwdFin f = {}; f.part = part; f.N = N; f.nblk = nblk; f.C = C; f.batch_mode = batch_mode; f.s1o = s1; f.s2o = s2; f.dgamma = dgamma; f.dbeta = dbeta;
  f.cpb = pick_cpb(C);
  return f;
}

// The three row-stream kernels are templated on the activation (no per-element switch) and have a fast path (C % 8 == 0,
// pitches % 8 == 0, aligned bases: unguarded 16-byte accesses, RU rows in flight per thread) next to the generic guarded loop.
struct NormArgs {
  const void* x; int ldx; const void* gy; int ldgy; const float* mean; const float* rstd; int ssn; const float* gamma; const float* beta;
  const void* res; int ldr; void* y; int ldy; void* gres; int ldgres; const float* s1; const float* s2; float inv_count; int use_stats;
  int64_t V; int C; int rpb; int nblk; float* part; int fast;
  // second row source for channels >= csplit (normalise-into-concat and its backward in ONE pass over whole y / gy rows);
  // backward: y2 / ldy2 = the second source's gx
  const void* x2; int ldx2; const float* mean2; const float* rstd2; int ssn2; int csplit; void* y2; int ldy2;
  // fp32x3 (T = float, fast path only): split_cp > 0: y (forward) / gx (backward) is a bf16 [rows][2 * split_cp] tensor of hi | lo
  // halves instead of fp32 rows (split_cp2: the same for the second source's gx); the consumer is an x3 convolution
  int split_cp, split_cp2;
  unsigned* ticket; NormBwdFin fin;      // backward partial pass with the finalize folded in (ticket_is_last); ticket == nullptr: rows only
  int rev;      // backward apply walks the blocks in reverse order (the rows the partial pass read last are the likeliest still cached)
};

template <typename T, int ACT>
__global__ void __launch_bounds__(NT) k_norm_act_fwd(NormArgs a) {
  const T* res = (const T*)a.res; T* y = (T*)a.y;
  const int b = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x, n = a.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  RowGeom g = row_geom(a.C);
  if (!g.active) return;
  const bool second = a.x2 != nullptr && g.cg * 8 >= a.csplit;      // this thread's 8-channel chunk comes from the second source
  const T* x = second ? (const T*)a.x2 : (const T*)a.x;
  const int ldx = second ? a.ldx2 : a.ldx, xoff = (second ? g.cg - (a.csplit >> 3) : g.cg) * 8;
  NormConst k = second ? load_consts(a.mean2, a.rstd2, n * a.ssn2, nullptr, nullptr, xoff, g.nv)
                       : load_consts(a.mean, a.rstd, n * a.ssn, a.gamma, a.beta, g.cg * 8, g.nv);
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { sc[i] = k.r[i] * k.ga[i]; sh[i] = k.be[i] - k.m[i] * sc[i]; }     // z = x*sc + sh
  const int64_t v0 = (int64_t)b * a.rpb, v1 = min(a.V, v0 + a.rpb), nb = (int64_t)n * a.V;
  int64_t vs = v0 + g.r0;
  if (a.fast) {
    for (; vs + (RU - 1) * (int64_t)g.rpi < v1; vs += RU * g.rpi) {
      float t[RU][8], rr[RU][8];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int64_t row = nb + vs + u * g.rpi;
        ld8(x + row * ldx + xoff, t[u]);
        if (res) ld8(res + row * a.ldr + g.cg * 8, rr[u]);
      }
#pragma unroll
      for (int u = 0; u < RU; u++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { float z = t[u][i] * sc[i] + sh[i]; if (res) z += rr[u][i]; t[u][i] = ((ACT == DP_ACT_MISH && sizeof(T) == 2) || ACT == DP_ACT_MISH_FAST) ? mish_fwd_fast(z) : act_fwd(z, ACT); }
        if constexpr (sizeof(T) == 4) {
          if (a.split_cp) { st8_split((bf16_t*)a.y + (nb + vs + u * g.rpi) * (2 * a.split_cp) + g.cg * 8, a.split_cp, t[u]); continue; }
        }
        st8(y + (nb + vs + u * g.rpi) * a.ldy + g.cg * 8, t[u]);
      }
    }
  }
  for (int64_t v = vs; v < v1; v += g.rpi) {
    const int64_t row = nb + v;
    float t[8], rr[8];
    unpack8<T>(x + row * ldx + xoff, g.nv, t);
    if (res) unpack8<T>(res + row * a.ldr + g.cg * 8, g.nv, rr);
#pragma unroll
    for (int i = 0; i < 8; i++) { float z = t[i] * sc[i] + sh[i]; if (res) z += rr[i]; t[i] = ((ACT == DP_ACT_MISH && sizeof(T) == 2) || ACT == DP_ACT_MISH_FAST) ? mish_fwd_fast(z) : act_fwd(z, ACT); }
    if constexpr (sizeof(T) == 4) {
      if (a.split_cp) { st8_split((bf16_t*)a.y + row * (2 * a.split_cp) + g.cg * 8, a.split_cp, t); continue; }      // (C % 8 == 0 is required)
    }
    pack8(y + row * a.ldy + g.cg * 8, g.nv, t);
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(NT) k_norm_act_bwd_partial(NormArgs a) {
  __shared__ __align__(16) float red[NT * 16];
  __shared__ int last_flag;
  const T* gy = (const T*)a.gy; const T* res = (const T*)a.res;
  const int b = blockIdx.x, n = blockIdx.y;
  RowGeom g = row_geom(a.C);
  const bool second = a.x2 != nullptr && g.cg * 8 >= a.csplit;
  const T* x = second ? (const T*)a.x2 : (const T*)a.x;
  const int ldx = second ? a.ldx2 : a.ldx, xoff = (second ? g.cg - (a.csplit >> 3) : g.cg) * 8;
  NormConst k = second ? load_consts(a.mean2, a.rstd2, n * a.ssn2, nullptr, nullptr, xoff, g.active ? g.nv : 0)
                       : load_consts(a.mean, a.rstd, n * a.ssn, a.gamma, a.beta, g.cg * 8, g.active ? g.nv : 0);
  const int64_t v0 = (int64_t)b * a.rpb, v1 = min(a.V, v0 + a.rpb), nb = (int64_t)n * a.V;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t vs = v0 + g.r0;
  auto body = [&](const float* t, const float* d, const float* rr) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float xh = (t[i] - k.m[i]) * k.r[i];
      float z = xh * k.ga[i] + k.be[i]; if (res) z += rr[i];
      float gg = d[i] * (((ACT == DP_ACT_MISH && sizeof(T) == 2) || ACT == DP_ACT_MISH_FAST) ? mish_bwd_fast(z) : act_bwd(z, ACT));
      s1[i] += gg; s2[i] += gg * xh;
    }
  };
  if (g.active && a.fast) {
    for (; vs + (RU - 1) * (int64_t)g.rpi < v1; vs += RU * g.rpi) {
      float t[RU][8], d[RU][8], rr[RU][8];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int64_t row = nb + vs + u * g.rpi;
        ld8(x + row * ldx + xoff, t[u]); ld8(gy + row * a.ldgy + g.cg * 8, d[u]);
        if (res) ld8(res + row * a.ldr + g.cg * 8, rr[u]);
      }
#pragma unroll
      for (int u = 0; u < RU; u++) body(t[u], d[u], rr[u]);
    }
  }
  if (g.active) for (int64_t v = vs; v < v1; v += g.rpi) {
    const int64_t row = nb + v;
    float t[8], d[8], rr[8];
    unpack8<T>(x + row * ldx + xoff, g.nv, t);
    unpack8<T>(gy + row * a.ldgy + g.cg * 8, g.nv, d);
    if (res) unpack8<T>(res + row * a.ldr + g.cg * 8, g.nv, rr);
    body(t, d, rr);
  }
  block_reduce_store(red, s1, s2, g, a.C, a.part + ((int64_t)n * a.nblk + b) * 2 * a.C, a.ticket != nullptr);
  if (!a.ticket) return;
  const bool bm = a.fin.batch_mode != 0;
  if (!ticket_is_last(a.ticket + (bm ? 0 : n), bm ? gridDim.x * gridDim.y : gridDim.x, &last_flag)) return;
  for (int bx = 0; bx * a.fin.cpb < a.C; bx++) { norm_bwd_finalize_unit<1>(a.fin, bx, bm ? 0 : n, (double*)red); __syncthreads(); }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(NT) k_norm_act_bwd_apply(NormArgs a) {
  const T* gy = (const T*)a.gy; const T* res = (const T*)a.res; T* gres = (T*)a.gres;
  const int b = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x, n = a.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  RowGeom g = row_geom(a.C);
  if (!g.active) return;
  const bool second = a.x2 != nullptr && g.cg * 8 >= a.csplit;
  const T* x = second ? (const T*)a.x2 : (const T*)a.x;
  T* gx = second ? (T*)a.y2 : (T*)a.y;
  const int ldx = second ? a.ldx2 : a.ldx, ldgx = second ? a.ldy2 : a.ldy, xoff = (second ? g.cg - (a.csplit >> 3) : g.cg) * 8;
  NormConst k = second ? load_consts(a.mean2, a.rstd2, n * a.ssn2, nullptr, nullptr, xoff, g.nv)
                       : load_consts(a.mean, a.rstd, n * a.ssn, a.gamma, a.beta, g.cg * 8, g.nv);
  const int sidx = a.x2 ? n * a.C : n * a.ssn;       // two-source mode: s1 / s2 are [N][Ca + Cb]
  float a1[8], a2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    bool ok = a.use_stats && i < g.nv;
    a1[i] = ok ? a.s1[sidx + g.cg * 8 + i] * a.inv_count : 0.f;
    a2[i] = ok ? a.s2[sidx + g.cg * 8 + i] * a.inv_count : 0.f;
  }
  const int64_t v0 = (int64_t)b * a.rpb, v1 = min(a.V, v0 + a.rpb), nb = (int64_t)n * a.V;
  int64_t vs = v0 + g.r0;
  auto body = [&](float* t, const float* d, const float* rr, float* gg) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float xh = (t[i] - k.m[i]) * k.r[i];
      float z = xh * k.ga[i] + k.be[i]; if (res) z += rr[i];
      gg[i] = d[i] * (((ACT == DP_ACT_MISH && sizeof(T) == 2) || ACT == DP_ACT_MISH_FAST) ? mish_bwd_fast(z) : act_bwd(z, ACT));
      t[i] = k.ga[i] * k.r[i] * (gg[i] - a1[i] - xh * a2[i]);
    }
  };
  if (a.fast) {
    for (; vs + (RU - 1) * (int64_t)g.rpi < v1; vs += RU * g.rpi) {
      float t[RU][8], d[RU][8], rr[RU][8], gg[RU][8];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int64_t row = nb + vs + u * g.rpi;
        ld8(x + row * ldx + xoff, t[u]); ld8(gy + row * a.ldgy + g.cg * 8, d[u]);
        if (res) ld8(res + row * a.ldr + g.cg * 8, rr[u]);
      }
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int64_t row = nb + vs + u * g.rpi;
        body(t[u], d[u], rr[u], gg[u]);
        bool done = false;
        if constexpr (sizeof(T) == 4) {
          const int scp = second ? a.split_cp2 : a.split_cp;
          if (scp && gx) { st8_split((bf16_t*)gx + row * (2 * scp) + xoff, scp, t[u]); done = true; }
        }
        if (gx && !done) st8(gx + row * ldgx + xoff, t[u]);
        if (gres) st8(gres + row * a.ldgres + g.cg * 8, gg[u]);
      }
    }
  }
  for (int64_t v = vs; v < v1; v += g.rpi) {
    const int64_t row = nb + v;
    float t[8], d[8], rr[8], gg[8];
    unpack8<T>(x + row * ldx + xoff, g.nv, t);
    unpack8<T>(gy + row * a.ldgy + g.cg * 8, g.nv, d);
    if (res) unpack8<T>(res + row * a.ldr + g.cg * 8, g.nv, rr);
    body(t, d, rr, gg);
    bool done = false;
    if constexpr (sizeof(T) == 4) {
      const int scp = second ? a.split_cp2 : a.split_cp;
      if (scp && gx) { st8_split((bf16_t*)gx + row * (2 * scp) + xoff, scp, t); done = true; }
    }
    if (gx && !done) pack8(gx + row * ldgx + xoff, g.nv, t);
    if (gres) pack8(gres + row * a.ldgres + g.cg * 8, g.nv, gg);
  }
}

// bit 0: the backward apply pass, bit 1: the forward pass walk their blocks from the last row to the first -- the rows the pass before
// (partial sums / the producing convolution) touched last are the likeliest still in L2 / MALL: 128^3 backward -7..-11 % alone on the
// chip (tools/bench_norm.py), but 23.25 -> 23.3 ms in the training step (profiles/r05_k_norm_block_order.txt): off by default
static inline int norm_rev() { static const int r = getenv("DP_NORM_REV") ? atoi(getenv("DP_NORM_REV")) : 0; return r; }
#define NORM_LAUNCH(KERN, args, grid) do { \
    switch (act) { \
      case DP_ACT_NONE: DP_DISPATCH(dtype, hipLaunchKernelGGL((KERN<T, DP_ACT_NONE>), grid, dim3(NT), 0, STREAM, args)); break; \
      case DP_ACT_RELU: DP_DISPATCH(dtype, hipLaunchKernelGGL((KERN<T, DP_ACT_RELU>), grid, dim3(NT), 0, STREAM, args)); break; \
      case DP_ACT_LRELU: DP_DISPATCH(dtype, hipLaunchKernelGGL((KERN<T, DP_ACT_LRELU>), grid, dim3(NT), 0, STREAM, args)); break; \
      case DP_ACT_MISH: DP_DISPATCH(dtype, hipLaunchKernelGGL((KERN<T, DP_ACT_MISH>), grid, dim3(NT), 0, STREAM, args)); break; \
      case DP_ACT_GELU: DP_DISPATCH(dtype, hipLaunchKernelGGL((KERN<T, DP_ACT_GELU>), grid, dim3(NT), 0, STREAM, args)); break; \
      case DP_ACT_MISH_FAST: if (dtype != DP_F32) DP_FAIL("DP_ACT_MISH_FAST is an fp32-storage activation code"); \
        hipLaunchKernelGGL((KERN<float, DP_ACT_MISH_FAST>), grid, dim3(NT), 0, STREAM, args); break; \
      default: DP_FAIL("bad activation %d", act); } } while (0)

static inline int norm_fast(int C, int ldx, const void* x, int ldg, const void* gy, int ldr, const void* res, int ldy, const void* y, int ldgr, const void* gr) {
  auto ok = [](int ld, const void* p) { return p == nullptr || (ld % 8 == 0 && aligned16(p)); };
  const int f = (C % 8 == 0) && ok(ldx, x) && ok(ldg, gy) && ok(ldr, res) && ok(ldy, y) && ok(ldgr, gr);
  if (!f && getenv("DP_DEBUG_SLOW")) fprintf(stderr, "[dp slow] norm row stream C=%d ldx=%d ldg=%d ldr=%d ldy=%d: guarded 8-element accesses\n", C, ldx, ldg, ldr, ldy);
  return f;
}

extern "C" int dp_norm_act_fwd(const void* x, int ldx, const float* mean, const float* rstd, int ssn, const float* gamma, const float* beta,
                               const void* res, int ldr, int act, void* y, int ldy, int N, int64_t V, int C, int dtype, void* stream) {
  if (C > 8 * NT) DP_FAIL("norm_act_fwd: C too large");
  NormArgs a = {}; a.x = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd; a.ssn = ssn; a.gamma = gamma; a.beta = beta; a.res = res; a.ldr = ldr;
  a.y = y; a.ldy = ldy; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, ldx, x, 0, nullptr, ldr, res, ldy, y, 0, nullptr);
  a.rev = (norm_rev() >> 1) & 1;
  NORM_LAUNCH(k_norm_act_fwd, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_fwd"); return 0;
}
extern "C" int dp_norm_act_cat_fwd(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                   const float* mean_b, const float* rstd_b, int Cb, int act, void* y, int ldy, int N, int64_t V, int dtype,
                                   void* stream) {
  const int C = Ca + Cb;
  if (C > 8 * NT || (Ca & 7) || (Cb & 7) || Ca <= 0 || Cb <= 0) DP_FAIL("norm_act_cat_fwd: channel counts must be positive multiples of 8 (%d, %d)", Ca, Cb);
  NormArgs a = {}; a.x = xa; a.ldx = lda; a.mean = mean_a; a.rstd = rstd_a; a.ssn = Ca; a.x2 = xb; a.ldx2 = ldb; a.mean2 = mean_b; a.rstd2 = rstd_b;
  a.ssn2 = Cb; a.csplit = Ca; a.y = y; a.ldy = ldy; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, lda, xa, 0, nullptr, 0, nullptr, ldy, y, 0, nullptr) && norm_fast(C, ldb, xb, 0, nullptr, 0, nullptr, ldy, y, 0, nullptr);
  a.rev = (norm_rev() >> 1) & 1;
  NORM_LAUNCH(k_norm_act_fwd, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_cat_fwd"); return 0;
}
// Whether the folded finalize can serve this backward: every statistics group finalizes on its own, so the all-sample dgamma / dbeta
// of an affine INSTANCE normalisation over several samples (which needs every group's rows) keeps the separate launch.
static unsigned* bwd_ticket(int N, int batch_mode, const float* dgamma, const float* dbeta, void* stream) {
  if (!ticket_enabled() || (!batch_mode && N > 1 && (dgamma || dbeta))) return nullptr;
  return ticket_counters(batch_mode ? 1 : N, STREAM);
}
static int norm_act_bwd_partial_impl(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                                     const float* gamma, const float* beta, const void* res, int ldr, int act, int N, int64_t V, int C,
                                     float* part, unsigned* ticket, NormBwdFin fin, int dtype, void* stream) {
  if (C > 8 * NT) DP_FAIL("norm_act_bwd_partial: C too large");
  NormArgs a = {}; a.x = x; a.ldx = ldx; a.gy = gy; a.ldgy = ldgy; a.mean = mean; a.rstd = rstd; a.ssn = ssn; a.gamma = gamma; a.beta = beta;
  a.res = res; a.ldr = ldr; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V); a.part = part;
  a.fast = norm_fast(C, ldx, x, ldgy, gy, ldr, res, 0, nullptr, 0, nullptr);
  a.ticket = ticket; a.fin = fin;
  NORM_LAUNCH(k_norm_act_bwd_partial, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_bwd_partial"); return 0;
}
extern "C" int dp_norm_act_bwd_partial(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                                       const float* gamma, const float* beta, const void* res, int ldr, int act, int N, int64_t V, int C,
                                       float* part, int dtype, void* stream) {
  return norm_act_bwd_partial_impl(x, ldx, gy, ldgy, mean, rstd, ssn, gamma, beta, res, ldr, act, N, V, C, part, nullptr, NormBwdFin{}, dtype, stream);
}
// dp_norm_act_bwd_partial + dp_norm_bwd_finalize in ONE launch (the last block of every statistics group finalizes it).  Returns 3
// (nothing launched) when the folded form is off or cannot serve the case (bwd_ticket): the caller then makes the two calls.
extern "C" int dp_norm_act_bwd_partial_finalize(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                                                const float* gamma, const float* beta, const void* res, int ldr, int act, int N, int64_t V, int C,
                                                float* part, int batch_mode, float* s1, float* s2, float* dgamma, float* dbeta, int dtype, void* stream) {
  unsigned* ticket = bwd_ticket(N, batch_mode, dgamma, dbeta, stream);
  if (!ticket) return 3;
  return norm_act_bwd_partial_impl(x, ldx, gy, ldgy, mean, rstd, ssn, gamma, beta, res, ldr, act, N, V, C, part, ticket,
                                   norm_bwd_fin(part, N, dp_stats_nblk(V), C, batch_mode, s1, s2, dgamma, dbeta), dtype, stream);
}

static int norm_act_cat_bwd_partial_impl(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                         const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, int N, int64_t V,
                                         float* part, unsigned* ticket, float* s1, float* s2, int dtype, void* stream) {
  const int C = Ca + Cb;
  if (C > 8 * NT || (Ca & 7) || (Cb & 7) || Ca <= 0 || Cb <= 0) DP_FAIL("norm_act_cat_bwd_partial: channel counts must be positive multiples of 8");
  NormArgs a = {}; a.x = xa; a.ldx = lda; a.mean = mean_a; a.rstd = rstd_a; a.ssn = Ca; a.x2 = xb; a.ldx2 = ldb; a.mean2 = mean_b; a.rstd2 = rstd_b;
  a.ssn2 = Cb; a.csplit = Ca; a.gy = gy; a.ldgy = ldgy; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V); a.part = part;
  a.fast = norm_fast(C, lda, xa, ldgy, gy, 0, nullptr, 0, nullptr, 0, nullptr) && norm_fast(C, ldb, xb, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr);
  a.ticket = ticket;
  if (ticket) a.fin = norm_bwd_fin(part, N, a.nblk, C, 0, s1, s2, nullptr, nullptr);
  NORM_LAUNCH(k_norm_act_bwd_partial, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_cat_bwd_partial"); return 0;
}
extern "C" int dp_norm_act_cat_bwd_partial(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                           const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, int N, int64_t V,
                                           float* part, int dtype, void* stream) {
  return norm_act_cat_bwd_partial_impl(xa, lda, mean_a, rstd_a, Ca, xb, ldb, mean_b, rstd_b, Cb, gy, ldgy, act, N, V, part, nullptr, nullptr, nullptr, dtype, stream);
}
// ... with the instance-mode finalize (s1 / s2 [N][Ca + Cb]) folded in; 3 = not launched (folded form off), make the two calls
extern "C" int dp_norm_act_cat_bwd_partial_finalize(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                                    const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, int N,
                                                    int64_t V, float* part, float* s1, float* s2, int dtype, void* stream) {
  unsigned* ticket = bwd_ticket(N, 0, nullptr, nullptr, stream);
  if (!ticket) return 3;
  return norm_act_cat_bwd_partial_impl(xa, lda, mean_a, rstd_a, Ca, xb, ldb, mean_b, rstd_b, Cb, gy, ldgy, act, N, V, part, ticket, s1, s2, dtype, stream);
}
extern "C" int dp_norm_act_cat_bwd_apply(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                         const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, const float* s1,
                                         const float* s2, float inv_count, void* gxa, int ldgxa, void* gxb, int ldgxb, int N, int64_t V, int dtype,
                                         void* stream) {
  const int C = Ca + Cb;
  if (C > 8 * NT || (Ca & 7) || (Cb & 7) || Ca <= 0 || Cb <= 0) DP_FAIL("norm_act_cat_bwd_apply: channel counts must be positive multiples of 8");
  if (!gxa || !gxb) DP_FAIL("norm_act_cat_bwd_apply: both gradients are produced");
  NormArgs a = {}; a.x = xa; a.ldx = lda; a.mean = mean_a; a.rstd = rstd_a; a.ssn = Ca; a.x2 = xb; a.ldx2 = ldb; a.mean2 = mean_b; a.rstd2 = rstd_b;
  a.ssn2 = Cb; a.csplit = Ca; a.gy = gy; a.ldgy = ldgy; a.y = gxa; a.ldy = ldgxa; a.y2 = gxb; a.ldy2 = ldgxb; a.s1 = s1; a.s2 = s2;
  a.inv_count = inv_count; a.use_stats = 1; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, lda, xa, ldgy, gy, 0, nullptr, ldgxa, gxa, 0, nullptr) && norm_fast(C, ldb, xb, 0, nullptr, 0, nullptr, ldgxb, gxb, 0, nullptr);
  a.rev = norm_rev() & 1;
  NORM_LAUNCH(k_norm_act_bwd_apply, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_cat_bwd_apply"); return 0;
}

__global__ void __launch_bounds__(256) k_norm_bwd_finalize(NormBwdFin f) {
  __shared__ double red[512];
  norm_bwd_finalize_unit<0>(f, blockIdx.x, blockIdx.y, red);
}
extern "C" int dp_norm_bwd_finalize(const float* part, int N, int nblk, int C, int batch_mode, float* s1, float* s2, float* dgamma, float* dbeta, void* stream) {
  NormBwdFin f = norm_bwd_fin(part, N, nblk, C, batch_mode, s1, s2, dgamma, dbeta);
  hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(cdiv(C, f.cpb), batch_mode ? 1 : N), dim3(256), 0, STREAM, f);
  DP_CHECK_LAUNCH("norm_bwd_finalize"); return 0;
}

extern "C" int dp_norm_act_bwd_apply(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                                     const float* gamma, const float* beta, const void* res, int ldr, int act, const float* s1, const float* s2,
                                     float inv_count, int use_stats, void* gx, int ldgx, void* gres, int ldgres, int N, int64_t V, int C,
                                     int dtype, void* stream) {
  if (C > 8 * NT) DP_FAIL("norm_act_bwd_apply: C too large");
  NormArgs a = {}; a.x = x; a.ldx = ldx; a.gy = gy; a.ldgy = ldgy; a.mean = mean; a.rstd = rstd; a.ssn = ssn; a.gamma = gamma; a.beta = beta;
  a.res = res; a.ldr = ldr; a.y = gx; a.ldy = ldgx; a.gres = gres; a.ldgres = ldgres; a.s1 = s1; a.s2 = s2; a.inv_count = inv_count;
  a.use_stats = use_stats; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, ldx, x, ldgy, gy, ldr, res, ldgx, gx, ldgres, gres);
  a.rev = norm_rev() & 1;
  NORM_LAUNCH(k_norm_act_bwd_apply, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_bwd_apply"); return 0;
}

// fp32x3 variants (fp32 tensors in, the result written as the bf16 [hi | lo] operand of the x3 convolution that consumes it; cp =
// channels per half, C <= cp, C % 8 == 0; channels [C, cp) of both halves are left untouched: the caller zero-fills them once)
extern "C" int dp_norm_act_fwd_x3(const void* x, int ldx, const float* mean, const float* rstd, int ssn, const float* gamma, const float* beta,
                                  const void* res, int ldr, int act, void* ys, int cp, int N, int64_t V, int C, void* stream) {
  if (C > 8 * NT || (C & 7) || cp < C || (cp & 7)) DP_FAIL("norm_act_fwd_x3: need C %% 8 == 0 and cp >= C (%d, %d)", C, cp);
  const int dtype = DP_F32;
  NormArgs a = {}; a.x = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd; a.ssn = ssn; a.gamma = gamma; a.beta = beta; a.res = res; a.ldr = ldr;
  a.y = ys; a.ldy = 2 * cp; a.split_cp = cp; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, ldx, x, 0, nullptr, ldr, res, 8, ys, 0, nullptr);
  a.rev = (norm_rev() >> 1) & 1;
  NORM_LAUNCH(k_norm_act_fwd, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_fwd_x3"); return 0;
}
extern "C" int dp_norm_act_bwd_apply_x3(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                                        const float* gamma, const float* beta, const void* res, int ldr, int act, const float* s1, const float* s2,
                                        float inv_count, int use_stats, void* gxs, int cp, void* gres, int ldgres, int N, int64_t V, int C, void* stream) {
  if (C > 8 * NT || (C & 7) || cp < C || (cp & 7) || !gxs) DP_FAIL("norm_act_bwd_apply_x3: need C %% 8 == 0, cp >= C and a destination");
  const int dtype = DP_F32;
  NormArgs a = {}; a.x = x; a.ldx = ldx; a.gy = gy; a.ldgy = ldgy; a.mean = mean; a.rstd = rstd; a.ssn = ssn; a.gamma = gamma; a.beta = beta;
  a.res = res; a.ldr = ldr; a.y = gxs; a.ldy = 2 * cp; a.split_cp = cp; a.gres = gres; a.ldgres = ldgres; a.s1 = s1; a.s2 = s2; a.inv_count = inv_count;
  a.use_stats = use_stats; a.V = V; a.C = C; a.rpb = rows_per_block(V); a.nblk = dp_stats_nblk(V);
  a.fast = norm_fast(C, ldx, x, ldgy, gy, ldr, res, 8, gxs, ldgres, gres);
  a.rev = norm_rev() & 1;
  NORM_LAUNCH(k_norm_act_bwd_apply, a, dim3(a.nblk, N));
  DP_CHECK_LAUNCH("norm_act_bwd_apply_x3"); return 0;
}

// ---------------------------------------------------------------------------- LayerNorm: one wave per row
// xb != nullptr: the row normalised is the STORED sum x + xb (rounded to T exactly as a separate add kernel would), which is also
// written to `sum` -- the residual add of a pre-norm transformer block fused with the LayerNorm that follows it.
template <typename T>
__global__ void k_layernorm_fwd(const T* __restrict__ x, const T* __restrict__ xb, T* __restrict__ sum, const float* __restrict__ gamma,
                                const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                int C, float eps) {
  int lane = threadIdx.x & 63; int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (xb) {
    for (int c = lane; c < C; c += 64) st_f(sum + row * C + c, ld_f(x + row * C + c) + ld_f(xb + row * C + c));
    x = sum;          // (each lane re-reads only the columns it wrote itself)
  }
  const T* a = x + row * C; T* o = y + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ld_f(a + c);
  float m = wave_sum(s) / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { float d = ld_f(a + c) - m; v += d * d; }
  float r = rsqrtf(wave_sum(v) / C + eps);
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
  for (int c = lane; c < C; c += 64) st_f(o + c, (ld_f(a + c) - m) * r * gamma[c] + beta[c]);
}
template <typename T>
__global__ void k_layernorm_bwd(const T* __restrict__ x, const T* __restrict__ gy, const float* __restrict__ gamma, const float* __restrict__ mean,
                                const float* __restrict__ rstd, T* __restrict__ gx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                int64_t rows, int C, int rows_per_block) {
  // block = 4 waves; each wave walks rows_per_block/4 rows; per-column partial sums of dgamma/dbeta kept per lane,
  // then one atomicAdd per (block, column).
  extern __shared__ float sm[];            // [2][C]
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int64_t rbeg = (int64_t)blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  for (int64_t row = rbeg + wv; row < rend; row += 4) {
    const T* a = x + row * C; const T* g = gy + row * C; T* o = gx + row * C;
    float m = mean[row], r = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) { float xh = (ld_f(a + c) - m) * r, dg = ld_f(g + c) * gamma[c]; s1 += dg; s2 += dg * xh; }
    s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
    for (int c = lane; c < C; c += 64) {
      float xh = (ld_f(a + c) - m) * r, d = ld_f(g + c);
      st_f(o + c, r * (d * gamma[c] - s1 - xh * s2));
      atomicAdd(&sm[c], d * xh); atomicAdd(&sm[C + c], d);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(dgamma + c, sm[c]); atomicAdd(dbeta + c, sm[C + c]); }
}
// C <= 1024: every lane keeps its (up to 16) columns of the row in registers -> one read of x / gy per row, and the
// dgamma / dbeta partials stay in registers over the wave's rows (no LDS atomics in the row loop).
template <typename T>
__global__ void __launch_bounds__(256) k_layernorm_bwd_reg(const T* __restrict__ x, const T* __restrict__ gy, const T* __restrict__ gres,
                                const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ gx,
                                float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C, int rows_per_block,
                                float* __restrict__ part = nullptr) {
  __shared__ float sm[4][2][1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float gam[16], ag[16], ab[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { int c = lane + 64 * i; gam[i] = c < C ? gamma[c] : 0.f; ag[i] = 0.f; ab[i] = 0.f; }
  int64_t rbeg = (int64_t)blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  for (int64_t row = rbeg + wv; row < rend; row += 4) {
    const T* a = x + row * C; const T* g = gy + row * C; T* o = gx + row * C;
    float m = mean[row], r = rstd[row];
    float xh[16], d[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int c = lane + 64 * i;
      if (c < C) { xh[i] = (ld_f(a + c) - m) * r; d[i] = ld_f(g + c); } else { xh[i] = 0.f; d[i] = 0.f; }
      float dg = d[i] * gam[i]; s1 += dg; s2 += dg * xh[i];
    }
    s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int c = lane + 64 * i;
      // gres: the gradient that reaches the same tensor through the residual path (fused add + LayerNorm): one rounding, one store
      if (c < C) st_f(o + c, r * (d[i] * gam[i] - s1 - xh[i] * s2) + (gres ? ld_f(gres + row * C + c) : 0.f));
      ag[i] += d[i] * xh[i]; ab[i] += d[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) { int c = lane + 64 * i; sm[wv][0][c] = ag[i]; sm[wv][1][c] = ab[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float dg = sm[0][0][c] + sm[1][0][c] + sm[2][0][c] + sm[3][0][c], db = sm[0][1][c] + sm[1][1][c] + sm[2][1][c] + sm[3][1][c];
    if (part) {      // deterministic mode: one partial row pair per block ([blk][0] = dbeta, [blk][1] = dgamma), combined by k_norm_bwd_finalize
      part[((int64_t)blockIdx.x * 2) * C + c] = db; part[((int64_t)blockIdx.x * 2 + 1) * C + c] = dg;
    } else { atomicAdd(dgamma + c, dg); atomicAdd(dbeta + c, db); }
  }
}
extern "C" int dp_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int C,
                                float eps, int dtype, void* stream) {
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_fwd<T>, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, (const T*)x, (const T*)nullptr, (T*)nullptr, gamma, beta,
                                        (T*)y, mean, rstd, rows, C, eps));
  DP_CHECK_LAUNCH("layernorm_fwd"); return 0;
}
extern "C" int dp_add_layernorm_fwd(const void* a, const void* b, void* sum, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    int64_t rows, int C, float eps, int dtype, void* stream) {
  if (!b || !sum) DP_FAIL("add_layernorm_fwd: operands missing");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_fwd<T>, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, (const T*)a, (const T*)b, (T*)sum, gamma, beta,
                                        (T*)y, mean, rstd, rows, C, eps));
  DP_CHECK_LAUNCH("add_layernorm_fwd"); return 0;
}
extern "C" int dp_layernorm_bwd(const void* x, const void* gy, const float* gamma, const float* mean, const float* rstd, void* gx, float* dgamma,
                                float* dbeta, int64_t rows, int C, int dtype, void* stream) {
  if (C <= 1024) {
    int rpb = rows >= 8192 ? 16 : (rows >= 2048 ? 8 : 4);          // token matrices (1024 rows): one row per wave, 256 blocks
    DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_bwd_reg<T>, dim3(cdiv(rows, rpb)), dim3(256), 0, STREAM, (const T*)x, (const T*)gy, (const T*)nullptr,
                                          gamma, mean, rstd, (T*)gx, dgamma, dbeta, rows, C, rpb));
    DP_CHECK_LAUNCH("layernorm_bwd"); return 0;
  }
  int rpb = 16;
  if ((size_t)2 * C * sizeof(float) > 64 * 1024) DP_FAIL("layernorm_bwd: C too large");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_bwd<T>, dim3(cdiv(rows, rpb)), dim3(256), 2 * C * sizeof(float), STREAM, (const T*)x, (const T*)gy, gamma,
                                        mean, rstd, (T*)gx, dgamma, dbeta, rows, C, rpb));
  DP_CHECK_LAUNCH("layernorm_bwd"); return 0;
}
// Deterministic LayerNorm backward (config.set_deterministic): the blocks' dgamma / dbeta partial sums go to `part`
// (dp_layernorm_bwd_parts(rows, C) x 2 x C floats) and are combined in a fixed order (fp64) instead of meeting in fp32 atomics; dgamma / dbeta
// are OVERWRITTEN.  gsum may be NULL (plain LayerNorm) or the residual-path gradient (fused add + LayerNorm).  C <= 1024.
static inline int ln_bwd_rpb(int64_t rows) { return rows >= 8192 ? 16 : (rows >= 2048 ? 8 : 4); }
extern "C" int dp_layernorm_bwd_parts(int64_t rows, int C) { (void)C; return cdiv(rows, ln_bwd_rpb(rows)); }
extern "C" int dp_add_layernorm_bwd_det(const void* x, const void* gy, const void* gsum, const float* gamma, const float* mean, const float* rstd, void* gx,
                                        float* part, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream) {
  if (C > 1024) DP_FAIL("add_layernorm_bwd_det: C > 1024 not supported");
  if (!part || !dgamma || !dbeta) DP_FAIL("add_layernorm_bwd_det: part / dgamma / dbeta missing");
  const int rpb = ln_bwd_rpb(rows), nblk = cdiv(rows, rpb);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_bwd_reg<T>, dim3(nblk), dim3(256), 0, STREAM, (const T*)x, (const T*)gy, (const T*)gsum,
                                        gamma, mean, rstd, (T*)gx, dgamma, dbeta, rows, C, rpb, part));
  DP_CHECK_LAUNCH("add_layernorm_bwd_det");
  return dp_norm_bwd_finalize(part, 1, nblk, C, 1, dbeta, dgamma, nullptr, nullptr, stream);      // s1 = sum of rows [.][0] = dbeta, s2 = dgamma
}
// backward of the fused residual add + LayerNorm: gx = gsum + LayerNorm'(gy)  (C <= 1024)
extern "C" int dp_add_layernorm_bwd(const void* x, const void* gy, const void* gsum, const float* gamma, const float* mean, const float* rstd, void* gx,
                                    float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream) {
  if (C > 1024) DP_FAIL("add_layernorm_bwd: C > 1024 not supported");
  int rpb = rows >= 8192 ? 16 : (rows >= 2048 ? 8 : 4);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_layernorm_bwd_reg<T>, dim3(cdiv(rows, rpb)), dim3(256), 0, STREAM, (const T*)x, (const T*)gy, (const T*)gsum,
                                        gamma, mean, rstd, (T*)gx, dgamma, dbeta, rows, C, rpb));
  DP_CHECK_LAUNCH("add_layernorm_bwd"); return 0;
}
