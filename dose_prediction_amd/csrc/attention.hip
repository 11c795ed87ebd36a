// Fused multi-head self-attention for the ViT token path (MONAI SABlock; call sites dose_pyfer.py:55-67,129 and
// oar_transeg.py:79-91,172): O = softmax(Q K^T * scale) V and its backward, without materialising the N x N scores.
// 16-bit storage (bf16 / fp16), head dim 64 or 128, any token count; fp32 storage keeps the GEMM + row-softmax path.
//
// The problem is tiny (N = 512 .. 1152 tokens, 12-24 (batch, head) pairs, < 0.3 % of the step's FLOPs) and was 14 launches
// of latency per layer, so the design minimises the dependent chain instead of chasing MFMA rate:
//  * a block owns 16*NF "fixed" rows (queries in the forward and dQ passes, keys in the dK/dV pass) whose fragments stay in
//    registers as MFMA B operands; its four waves split the "streamed" rows (32-row chunks, chunk c -> wave c & 3) and are
//    combined once at the end through LDS, so every wave runs a 4..9 iteration loop;
//  * the score tile is computed TRANSPOSED (streamed x fixed): the 16x16 MFMA accumulator layout (col = lane & 15, rows
//    4*(lane >> 4) + r) then is, register for register, the B-operand layout of the second MFMA (k slots 8*(lane >> 4) + j)
//    once two 16-row tiles are paired, so P / dS never leave registers;
//  * the second MFMA needs the streamed operand k-major (V^T, K^T, Q^T, dO^T): the chunk is staged row-major in a
//    wave-private LDS tile and read with ds_read_b64_tr_b16; the k-slot -> row mapping of that read ({4g..4g+3} of tile a,
//    {4g..4g+3} of tile b) is exactly the pairing above.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

template <int D> struct AttCfg {
  static constexpr int KK = D / 32;        // K=32 MFMA steps over the head dim
  static constexpr int DT = D / 16;        // 16-row output tiles over the head dim
  static constexpr int LDP = D + 8;        // LDS row pitch of a staged chunk (elements)
  static constexpr int CP = D + 4;         // pitch of the fp32 combine tile
  static constexpr int CH = 32;            // streamed rows per chunk
  static constexpr int NW = 4;             // waves per block
};

template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (f16_t)a) | ((unsigned)__builtin_bit_cast(unsigned short, (f16_t)b) << 16);
}
template <typename T> __device__ __forceinline__ Frag8<T> pack_frag(const float* p) {
  Frag8<T> f;
  f.u = (v4u){pack2<T>(p[0], p[1]), pack2<T>(p[2], p[3]), pack2<T>(p[4], p[5]), pack2<T>(p[6], p[7])};
  return f;
}
template <typename T> __device__ __forceinline__ Frag8<T> ldg_frag(const T* p) { Frag8<T> f; f.u = *(const v4u*)p; return f; }
template <typename T> __device__ __forceinline__ float unpack_lo(unsigned w);
template <> __device__ __forceinline__ float unpack_lo<bf16_t>(unsigned w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float unpack_lo<f16_t>(unsigned w) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(w & 0xffff)); }
template <typename T> __device__ __forceinline__ float unpack_hi(unsigned w) { return unpack_lo<T>(w >> 16); }

__device__ __forceinline__ float group_max(float v) {   // over the four lanes l, l^16, l^32, l^48 (same column)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Blocks are dealt to the 8 XCDs round-robin and every XCD has a private L2: give each XCD a contiguous range of the
// (batch*head, tile) space so that one head's K / V (Q / dO) is pulled through the fabric by one or two XCDs, not by all eight
// (measured: the forward was fabric-bound at 15 us for 0.8 GFLOP before this).
__device__ __forceinline__ int xcd_order(int id, int total) {
  return (total & 7) ? id : (id & 7) * (total >> 3) + (id >> 3);
}

// Rows [r0, r0+32) (clamped to N-1) of a [N][ld] matrix: global -> registers, registers -> wave-private LDS tile [32][LDP].
// Split so that the loads of the next chunk are in flight while the current one is being consumed.
template <int D> struct ChunkRegs { v4u t[D / 16]; };
template <typename T, int D>
__device__ __forceinline__ ChunkRegs<D> load_chunk(const T* __restrict__ src, int64_t ld, int r0, int N, int lane) {
  constexpr int PER_ROW = D / 8;
  ChunkRegs<D> c;
#pragma unroll
  for (int i = 0; i < D / 16; i++) {
    const int idx = lane + 64 * i, row = idx / PER_ROW, c8 = idx % PER_ROW;
    c.t[i] = *(const v4u*)(src + (int64_t)min(r0 + row, N - 1) * ld + 8 * c8);
  }
  return c;
}
template <typename T, int D>
__device__ __forceinline__ void store_chunk(const ChunkRegs<D>& c, T* dst, int lane) {
  constexpr int PER_ROW = D / 8;
#pragma unroll
  for (int i = 0; i < D / 16; i++) {
    const int idx = lane + 64 * i, row = idx / PER_ROW, c8 = idx % PER_ROW;
    *(v4u*)(dst + row * AttCfg<D>::LDP + 8 * c8) = c.t[i];
  }
}

// Sum (or softmax-merge) of the four waves' fp32 partial tiles [NW][ROWS][CP] -> 16-bit rows of `out`.
// wscale[w][row] (LDS) is the per-wave, per-row factor (1 for plain sums).
template <typename T, int D, int ROWS, bool SCALED>
__device__ __forceinline__ void combine_store(const float* part, const float* wscale, T* __restrict__ out, int64_t ldo, int row0, int N,
                                              int tid) {
  using C = AttCfg<D>;
  constexpr int PER_ROW = D / 8;
  for (int idx = tid; idx < ROWS * PER_ROW; idx += 256) {
    const int row = idx / PER_ROW, c8 = idx % PER_ROW;
    if (row0 + row >= N) continue;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < C::NW; w++) {
      const float* p = part + ((w * ROWS + row) * C::CP + 8 * c8);
      const v4f a = *(const v4f*)p, b = *(const v4f*)(p + 4);
      const float f = SCALED ? wscale[w * ROWS + row] : 1.f;
      o[0] += a[0] * f; o[1] += a[1] * f; o[2] += a[2] * f; o[3] += a[3] * f;
      o[4] += b[0] * f; o[5] += b[1] * f; o[6] += b[2] * f; o[7] += b[3] * f;
    }
    *(v4u*)(out + (int64_t)(row0 + row) * ldo + 8 * c8) = pack_frag<T>(o).u;
  }
}

template <int D, int NF> struct AttSmem {
  using C = AttCfg<D>;
  static constexpr size_t stage1 = (size_t)C::NW * C::CH * C::LDP * 2;          // one staged array per wave
  static constexpr size_t comb1 = (size_t)C::NW * 16 * NF * C::CP * 4;          // one fp32 combine tile set
  static constexpr size_t stats = (size_t)2 * C::NW * 16 * NF * 4;              // m / l (or scale) per wave and row
  static constexpr size_t fwd = (stage1 > comb1 ? stage1 : comb1) + stats;
  static constexpr size_t bwd_q = (2 * stage1 > comb1 ? 2 * stage1 : comb1);
  static constexpr size_t bwd_kv = (2 * stage1 > 2 * comb1 ? 2 * stage1 : 2 * comb1);
};

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int D, int NF>
__global__ __launch_bounds__(256) void k_attn_fwd(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
                                                  int64_t ld, T* __restrict__ O, int64_t ldo, float* __restrict__ LSE, int heads,
                                                  int N, float c) {
  using C = AttCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const int tiles = (N + 16 * NF - 1) / (16 * NF), flat = xcd_order(blockIdx.x, gridDim.x);
  const int bh = flat / tiles, b = bh / heads, h = bh % heads, q0 = (flat % tiles) * (16 * NF);
  const int64_t boff = (int64_t)b * N * ld + (int64_t)h * D;
  const T *Qb = Q + boff, *Kb = K + boff, *Vb = V + boff;
  Frag8<T> xq[NF][C::KK];
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int kk = 0; kk < C::KK; kk++) xq[f][kk] = ldg_frag(Qb + (int64_t)min(q0 + 16 * f + r, N - 1) * ld + 32 * kk + 8 * g);
  v4f acc[NF][C::DT];
  float m[NF], ls[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    m[f] = -INFINITY; ls[f] = 0.f;
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) acc[f][dt] = (v4f){0, 0, 0, 0};
  }
  T* vs = (T*)smem + wv * (C::CH * C::LDP);
  const T* vtr = vs + (4 * g + (r >> 2)) * C::LDP + 4 * (r & 3);
  const int nch = (N + 31) >> 5;
  auto load_k = [&](int k0, Frag8<T> (&kf)[2][C::KK]) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int kk = 0; kk < C::KK; kk++) kf[t][kk] = ldg_frag(Kb + (int64_t)min(k0 + 16 * t + r, N - 1) * ld + 32 * kk + 8 * g);
  };
  ChunkRegs<D> vn;
  Frag8<T> kn[2][C::KK];
  if (wv < nch) { vn = load_chunk<T, D>(Vb, ld, wv * 32, N, lane); load_k(wv * 32, kn); }
  for (int ch = wv; ch < nch; ch += C::NW) {
    const int k0 = ch * 32;
    store_chunk<T, D>(vn, vs, lane);
    Frag8<T> ka[2][C::KK];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int kk = 0; kk < C::KK; kk++) ka[t][kk] = kn[t][kk];
    if (ch + C::NW < nch) { vn = load_chunk<T, D>(Vb, ld, k0 + 32 * C::NW, N, lane); load_k(k0 + 32 * C::NW, kn); }
    __builtin_amdgcn_sched_barrier(0);
    Frag8<T> pf[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
      v4f s[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int kk = 0; kk < C::KK; kk++) s[t] = mma16(ka[t][kk], xq[f][kk], s[t]);
      float tv[8];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float x = s[t][j] * c;
          tv[4 * t + j] = (k0 + 16 * t + 4 * g + j < N) ? x : -INFINITY;
        }
      float ml = tv[0];
#pragma unroll
      for (int i = 1; i < 8; i++) ml = fmaxf(ml, tv[i]);
      const float mn = fmaxf(m[f], group_max(ml)), alpha = __builtin_amdgcn_exp2f(m[f] - mn);
      m[f] = mn;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) { tv[i] = __builtin_amdgcn_exp2f(tv[i] - mn); sum += tv[i]; }
      ls[f] = ls[f] * alpha + sum;
#pragma unroll
      for (int dt = 0; dt < C::DT; dt++) acc[f][dt] *= alpha;
      pf[f] = pack_frag<T>(tv);
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
      const Frag8<T> vf = tr_pair<16 * C::LDP, T>(vtr + 16 * dt);
#pragma unroll
      for (int f = 0; f < NF; f++) acc[f][dt] = mma16(vf, pf[f], acc[f][dt]);
    }
  }
  // merge the four key ranges: O = sum_w O_w 2^(m_w - M) / sum_w l_w 2^(m_w - M)
  constexpr int ROWS = 16 * NF;
  constexpr size_t big = AttSmem<D, NF>::stage1 > AttSmem<D, NF>::comb1 ? AttSmem<D, NF>::stage1 : AttSmem<D, NF>::comb1;
  float* part = (float*)smem;
  float* mw = (float*)(smem + big);
  float* lw = mw + C::NW * ROWS;
  __syncthreads();   // every wave is done with its staging tile
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const float l = group_sum(ls[f]);
    if (g == 0) { mw[wv * ROWS + 16 * f + r] = m[f]; lw[wv * ROWS + 16 * f + r] = l; }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) *(v4f*)(part + ((wv * ROWS + 16 * f + r) * C::CP + 16 * dt + 4 * g)) = acc[f][dt];
  }
  __syncthreads();
  if (tid < ROWS) {
    float M = mw[tid];
#pragma unroll
    for (int w = 1; w < C::NW; w++) M = fmaxf(M, mw[w * ROWS + tid]);
    float e[C::NW], L = 0.f;
#pragma unroll
    for (int w = 0; w < C::NW; w++) { e[w] = __builtin_amdgcn_exp2f(mw[w * ROWS + tid] - M); L += lw[w * ROWS + tid] * e[w]; }
    const float inv = 1.f / L;
#pragma unroll
    for (int w = 0; w < C::NW; w++) lw[w * ROWS + tid] = e[w] * inv;
    if (q0 + tid < nch * 32) LSE[(int64_t)bh * (nch * 32) + q0 + tid] = M + __builtin_amdgcn_logf(L);   // log2; row pitch = N rounded up to 32
  }
  __syncthreads();
  combine_store<T, D, ROWS, true>(part, lw, O + (int64_t)b * N * ldo + (int64_t)h * D, ldo, q0, N, tid);
}


// ------------------------------------------------------------------------------------------------ forward, fp32x3 mode (DP_X3)
// The same contraction on fp32 q / k / v with the split-bf16 arithmetic of the fp32x3 mode (csrc/x3.hip): every operand is split into
// hi = bf16(v), lo = bf16(v - hi) IN REGISTERS and every product takes three MFMAs (hi hi + lo hi + hi lo), fp32 accumulation, fp32
// output.  Replaces the exact-fp32 GEMM + row softmax + transpose + GEMM chain the mode used for its forward pass (4 launches and
// ~0.11 ms per layer at B = 2, N = 512: `k_gemm_nt<float>` 39 launches, 0.85 ms of the fp32x3 DOSE-PYFER step).  One 16-row query tile
// per block (the doubled fragments leave no room for a second), keys / values streamed in 32-row chunks per wave as above; the V chunk
// is staged as two bf16 tiles (hi, lo) so that the k-major read stays ds_read_b64_tr_b16.
__device__ __forceinline__ void split_frag(const v4f& a, const v4f& b, Frag8<bf16_t>& hi, Frag8<bf16_t>& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const bf16_t ah = f2bf(v[2 * d]), bh = f2bf(v[2 * d + 1]);
    hi.u[d] = (unsigned)ah | ((unsigned)bh << 16);
    lo.u[d] = (unsigned)f2bf(v[2 * d] - bf2f(ah)) | ((unsigned)f2bf(v[2 * d + 1] - bf2f(bh)) << 16);
  }
}
template <int D>
__global__ __launch_bounds__(256) void k_attn_fwd_x3(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, int64_t ld,
                                                     float* __restrict__ O, int64_t ldo, int heads, int N, float c) {
  using C = AttCfg<D>;
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const int tiles = (N + 15) / 16, flat = xcd_order(blockIdx.x, gridDim.x);
  const int bh = flat / tiles, b = bh / heads, h = bh % heads, q0 = (flat % tiles) * 16;
  const int64_t boff = (int64_t)b * N * ld + (int64_t)h * D;
  const float *Qb = Q + boff, *Kb = K + boff, *Vb = V + boff;
  Frag8<T> qh[C::KK], ql[C::KK];
#pragma unroll
  for (int kk = 0; kk < C::KK; kk++) {
    const float* p = Qb + (int64_t)min(q0 + r, N - 1) * ld + 32 * kk + 8 * g;
    split_frag(*(const v4f*)p, *(const v4f*)(p + 4), qh[kk], ql[kk]);
  }
  v4f acc[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; dt++) acc[dt] = (v4f){0, 0, 0, 0};
  float m = -INFINITY, ls = 0.f;
  // wave-private staging: [2 (hi, lo)][32 rows][LDP]
  T* vs = (T*)smem + wv * (2 * C::CH * C::LDP);
  const T* vtr = vs + (4 * g + (r >> 2)) * C::LDP + 4 * (r & 3);
  const int nch = (N + 31) >> 5;
  constexpr int PER_ROW = D / 8;
  for (int ch = wv; ch < nch; ch += C::NW) {
    const int k0 = ch * 32;
    // V chunk: 32 rows x D floats -> hi / lo tiles (each lane converts D / 16 pieces of 8)
#pragma unroll
    for (int i = 0; i < D / 16; i++) {
      const int idx = lane + 64 * i, row = idx / PER_ROW, c8 = idx % PER_ROW;
      const float* p = Vb + (int64_t)min(k0 + row, N - 1) * ld + 8 * c8;
      Frag8<T> vh, vl;
      split_frag(*(const v4f*)p, *(const v4f*)(p + 4), vh, vl);
      *(v4u*)(vs + row * C::LDP + 8 * c8) = vh.u;
      *(v4u*)(vs + (C::CH + row) * C::LDP + 8 * c8) = vl.u;
    }
    v4f s[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int kk = 0; kk < C::KK; kk++) {
        const float* p = Kb + (int64_t)min(k0 + 16 * t + r, N - 1) * ld + 32 * kk + 8 * g;
        Frag8<T> kh, kl;
        split_frag(*(const v4f*)p, *(const v4f*)(p + 4), kh, kl);
        s[t] = mma16(kh, qh[kk], s[t]);
        s[t] = mma16(kl, qh[kk], s[t]);
        s[t] = mma16(kh, ql[kk], s[t]);
      }
    float tv[8];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float x = s[t][j] * c;
        tv[4 * t + j] = (k0 + 16 * t + 4 * g + j < N) ? x : -INFINITY;
      }
    float ml = tv[0];
#pragma unroll
    for (int i = 1; i < 8; i++) ml = fmaxf(ml, tv[i]);
    const float mn = fmaxf(m, group_max(ml)), alpha = exp2f(m - mn);
    m = mn;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) { tv[i] = exp2f(tv[i] - mn); sum += tv[i]; }
    ls = ls * alpha + sum;
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) acc[dt] *= alpha;
    Frag8<T> ph, pl;
    split_frag((v4f){tv[0], tv[1], tv[2], tv[3]}, (v4f){tv[4], tv[5], tv[6], tv[7]}, ph, pl);
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();          // the wave's own LDS writes are visible to its reads
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
      const Frag8<T> vh = tr_pair<16 * C::LDP, T>(vtr + 16 * dt), vl = tr_pair<16 * C::LDP, T>(vtr + C::CH * C::LDP + 16 * dt);
      acc[dt] = mma16(vh, ph, acc[dt]);
      acc[dt] = mma16(vl, ph, acc[dt]);
      acc[dt] = mma16(vh, pl, acc[dt]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // merge the four key ranges (as in k_attn_fwd), fp32 rows out
  constexpr int ROWS = 16;
  float* part = (float*)smem;
  float* mw = (float*)(smem + (size_t)C::NW * 2 * C::CH * C::LDP * 2);
  float* lw = mw + C::NW * ROWS;
  __syncthreads();
  {
    const float l = group_sum(ls);
    if (g == 0) { mw[wv * ROWS + r] = m; lw[wv * ROWS + r] = l; }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) *(v4f*)(part + ((wv * ROWS + r) * C::CP + 16 * dt + 4 * g)) = acc[dt];
  }
  __syncthreads();
  if (tid < ROWS) {
    float M = mw[tid];
#pragma unroll
    for (int w = 1; w < C::NW; w++) M = fmaxf(M, mw[w * ROWS + tid]);
    float e[C::NW], L = 0.f;
#pragma unroll
    for (int w = 0; w < C::NW; w++) { e[w] = exp2f(mw[w * ROWS + tid] - M); L += lw[w * ROWS + tid] * e[w]; }
    const float inv = 1.f / L;
#pragma unroll
    for (int w = 0; w < C::NW; w++) lw[w * ROWS + tid] = e[w] * inv;
  }
  __syncthreads();
  float* Ob = O + (int64_t)b * N * ldo + (int64_t)h * D;
  for (int idx = tid; idx < ROWS * (D / 4); idx += 256) {
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    if (q0 + row >= N) continue;
    v4f o = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < C::NW; w++) o += *(const v4f*)(part + ((w * ROWS + row) * C::CP + 4 * c4)) * lw[w * ROWS + row];
    *(v4f*)(Ob + (int64_t)(q0 + row) * ldo + 4 * c4) = o;
  }
}

// ------------------------------------------------------------------------------------------------ backward
// delta[bh][q] = sum_d dO[q][d] O[q][d]   (= rowsum(dP o P)); rows of lse / delta have pitch Np = N rounded up to 32
template <typename T, int D>
__global__ __launch_bounds__(256) void k_attn_delta(const T* __restrict__ O, const T* __restrict__ dO, int64_t ldo, float* __restrict__ delta,
                                                    int heads, int N, int Np, int64_t rows) {
  constexpr int LPR = D / 8;
  const int64_t gid = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;   // row over (b, h, q < Np)
  const int c8 = threadIdx.x % LPR;
  float s = 0.f;
  const int64_t bh = gid / Np, q = gid % Np, b = bh / heads, h = bh % heads;
  if (gid < rows && q < N) {
    const int64_t off = (b * N + q) * ldo + h * D + 8 * c8;
    const v4u a = *(const v4u*)(O + off), d = *(const v4u*)(dO + off);
#pragma unroll
    for (int i = 0; i < 4; i++) s += unpack_lo<T>(a[i]) * unpack_lo<T>(d[i]) + unpack_hi<T>(a[i]) * unpack_hi<T>(d[i]);
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (gid < rows && c8 == 0) delta[gid] = s;   // 0 in the padding rows
}

struct BwdArgs {
  const void *Q, *K, *V, *dO;
  const float *LSE, *delta;
  void *dQ, *dK, *dV;
  int64_t ld, ldo, ldg;
  int heads, N, Np, nq_tiles, tiles;
  float c, scale;
};

// dQ: fixed = query rows, streamed = keys.  dQ^T += K^T dS^T with dS^T = P^T o (dP^T - delta) * scale.
template <typename T, int D, int NF>
__device__ __forceinline__ void attn_bwd_q(const BwdArgs& a, char* smem, int bh, int tile) {
  using C = AttCfg<D>;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const int N = a.N, b = bh / a.heads, h = bh % a.heads, q0 = tile * (16 * NF);
  const int64_t ld = a.ld, ldo = a.ldo;
  const int64_t boff = (int64_t)b * N * ld + (int64_t)h * D, ooff = (int64_t)b * N * ldo + (int64_t)h * D;
  const T *Qb = (const T*)a.Q + boff, *Kb = (const T*)a.K + boff, *Vb = (const T*)a.V + boff, *Gb = (const T*)a.dO + ooff;
  const float c = a.c, scale = a.scale;
  Frag8<T> xq[NF][C::KK], xo[NF][C::KK];
  float lse[NF], dl[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const int row = min(q0 + 16 * f + r, N - 1);
    lse[f] = a.LSE[(int64_t)bh * a.Np + row]; dl[f] = a.delta[(int64_t)bh * a.Np + row];
#pragma unroll
    for (int kk = 0; kk < C::KK; kk++) {
      xq[f][kk] = ldg_frag(Qb + (int64_t)row * ld + 32 * kk + 8 * g);
      xo[f][kk] = ldg_frag(Gb + (int64_t)row * ldo + 32 * kk + 8 * g);
    }
  }
  v4f acc[NF][C::DT];
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) acc[f][dt] = (v4f){0, 0, 0, 0};
  T* ks = (T*)smem + wv * (2 * C::CH * C::LDP);
  T* vs = ks + C::CH * C::LDP;
  const int tro = (4 * g + (r >> 2)) * C::LDP + 4 * (r & 3), fro = r * C::LDP + 8 * g;
  const int nch = (N + 31) >> 5;
  ChunkRegs<D> kn, vn;
  if (wv < nch) { kn = load_chunk<T, D>(Kb, ld, wv * 32, N, lane); vn = load_chunk<T, D>(Vb, ld, wv * 32, N, lane); }
  for (int ch = wv; ch < nch; ch += C::NW) {
    const int k0 = ch * 32;
    store_chunk<T, D>(kn, ks, lane);
    store_chunk<T, D>(vn, vs, lane);
    if (ch + C::NW < nch) {
      kn = load_chunk<T, D>(Kb, ld, k0 + 32 * C::NW, N, lane);
      vn = load_chunk<T, D>(Vb, ld, k0 + 32 * C::NW, N, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    Frag8<T> dsf[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
      v4f s[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}}, dp[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int kk = 0; kk < C::KK; kk++) {
          s[t] = mma16(frag_ld_lds(ks + fro + 16 * t * C::LDP + 32 * kk), xq[f][kk], s[t]);
          dp[t] = mma16(frag_ld_lds(vs + fro + 16 * t * C::LDP + 32 * kk), xo[f][kk], dp[t]);
        }
      float tv[8];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float p = __builtin_amdgcn_exp2f(s[t][j] * c - lse[f]);
          tv[4 * t + j] = (k0 + 16 * t + 4 * g + j < N) ? p * (dp[t][j] - dl[f]) * scale : 0.f;
        }
      dsf[f] = pack_frag<T>(tv);
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
      const Frag8<T> kf = tr_pair<16 * C::LDP, T>(ks + tro + 16 * dt);
#pragma unroll
      for (int f = 0; f < NF; f++) acc[f][dt] = mma16(kf, dsf[f], acc[f][dt]);
    }
  }
  constexpr int ROWS = 16 * NF;
  float* part = (float*)smem;
  __syncthreads();
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) *(v4f*)(part + ((wv * ROWS + 16 * f + r) * C::CP + 16 * dt + 4 * g)) = acc[f][dt];
  __syncthreads();
  combine_store<T, D, ROWS, false>(part, nullptr, (T*)a.dQ + (int64_t)b * N * a.ldg + (int64_t)h * D, a.ldg, q0, N, tid);
}

// dK / dV: fixed = key rows, streamed = queries.  dV^T += dO^T P,  dK^T += Q^T dS.
template <typename T, int D, int NF, bool PF>
__device__ __forceinline__ void attn_bwd_kv(const BwdArgs& a, char* smem, int bh, int tile) {
  using C = AttCfg<D>;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const int N = a.N, b = bh / a.heads, h = bh % a.heads, k0 = tile * (16 * NF);
  const int64_t ld = a.ld, ldo = a.ldo;
  const int64_t boff = (int64_t)b * N * ld + (int64_t)h * D, ooff = (int64_t)b * N * ldo + (int64_t)h * D;
  const T *Qb = (const T*)a.Q + boff, *Kb = (const T*)a.K + boff, *Vb = (const T*)a.V + boff, *Gb = (const T*)a.dO + ooff;
  const float *lseb = a.LSE + (int64_t)bh * a.Np + 4 * g, *dlb = a.delta + (int64_t)bh * a.Np + 4 * g;
  const float c = a.c, scale = a.scale;
  Frag8<T> xk[NF][C::KK], xv[NF][C::KK];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const int row = min(k0 + 16 * f + r, N - 1);
#pragma unroll
    for (int kk = 0; kk < C::KK; kk++) {
      xk[f][kk] = ldg_frag(Kb + (int64_t)row * ld + 32 * kk + 8 * g);
      xv[f][kk] = ldg_frag(Vb + (int64_t)row * ld + 32 * kk + 8 * g);
    }
  }
  v4f ak[NF][C::DT], av[NF][C::DT];
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) { ak[f][dt] = (v4f){0, 0, 0, 0}; av[f][dt] = (v4f){0, 0, 0, 0}; }
  T* qs = (T*)smem + wv * (2 * C::CH * C::LDP);
  T* gs = qs + C::CH * C::LDP;
  const int tro = (4 * g + (r >> 2)) * C::LDP + 4 * (r & 3), fro = r * C::LDP + 8 * g;
  const int nch = (N + 31) >> 5;
  ChunkRegs<D> qn, gn;
  v4f lsn[2], dln[2];   // row statistics of the chunk: rows 16 t + 4 g + j
  auto load_stats = [&](int q0) {
#pragma unroll
    for (int t = 0; t < 2; t++) { lsn[t] = *(const v4f*)(lseb + q0 + 16 * t); dln[t] = *(const v4f*)(dlb + q0 + 16 * t); }
  };
  if (PF && wv < nch) { qn = load_chunk<T, D>(Qb, ld, wv * 32, N, lane); gn = load_chunk<T, D>(Gb, ldo, wv * 32, N, lane); load_stats(wv * 32); }
  for (int ch = wv; ch < nch; ch += C::NW) {
    const int q0 = ch * 32;
    if (!PF) { qn = load_chunk<T, D>(Qb, ld, q0, N, lane); gn = load_chunk<T, D>(Gb, ldo, q0, N, lane); load_stats(q0); }
    store_chunk<T, D>(qn, qs, lane);
    store_chunk<T, D>(gn, gs, lane);
    const v4f lse[2] = {lsn[0], lsn[1]}, dl[2] = {dln[0], dln[1]};
    if (PF && ch + C::NW < nch) {
      qn = load_chunk<T, D>(Qb, ld, q0 + 32 * C::NW, N, lane);
      gn = load_chunk<T, D>(Gb, ldo, q0 + 32 * C::NW, N, lane);
      load_stats(q0 + 32 * C::NW);
    }
    if (PF) __builtin_amdgcn_sched_barrier(0);
    Frag8<T> pf[NF], dsf[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
      v4f s[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}}, dp[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int kk = 0; kk < C::KK; kk++) {
          s[t] = mma16(frag_ld_lds(qs + fro + 16 * t * C::LDP + 32 * kk), xk[f][kk], s[t]);
          dp[t] = mma16(frag_ld_lds(gs + fro + 16 * t * C::LDP + 32 * kk), xv[f][kk], dp[t]);
        }
      float pv[8], dv[8];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int i = 4 * t + j;
          const bool live = q0 + 16 * t + 4 * g + j < N;
          const float p = live ? __builtin_amdgcn_exp2f(s[t][j] * c - lse[t][j]) : 0.f;
          pv[i] = p; dv[i] = live ? p * (dp[t][j] - dl[t][j]) * scale : 0.f;
        }
      pf[f] = pack_frag<T>(pv); dsf[f] = pack_frag<T>(dv);
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
      const Frag8<T> gf = tr_pair<16 * C::LDP, T>(gs + tro + 16 * dt), qf = tr_pair<16 * C::LDP, T>(qs + tro + 16 * dt);
#pragma unroll
      for (int f = 0; f < NF; f++) {
        av[f][dt] = mma16(gf, pf[f], av[f][dt]);
        ak[f][dt] = mma16(qf, dsf[f], ak[f][dt]);
      }
    }
  }
  constexpr int ROWS = 16 * NF;
  float* pk = (float*)smem;
  float* pvv = pk + C::NW * ROWS * C::CP;
  __syncthreads();
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
      *(v4f*)(pk + ((wv * ROWS + 16 * f + r) * C::CP + 16 * dt + 4 * g)) = ak[f][dt];
      *(v4f*)(pvv + ((wv * ROWS + 16 * f + r) * C::CP + 16 * dt + 4 * g)) = av[f][dt];
    }
  __syncthreads();
  const int64_t goff = (int64_t)b * N * a.ldg + (int64_t)h * D;
  combine_store<T, D, ROWS, false>(pk, nullptr, (T*)a.dK + goff, a.ldg, k0, N, tid);
  combine_store<T, D, ROWS, false>(pvv, nullptr, (T*)a.dV + goff, a.ldg, k0, N, tid);
}

// One launch for both passes: per (batch, head), tiles [0, nq_tiles) produce dQ, the rest dK / dV; the two are independent
// latency chains of similar length, so side by side they take the time of one.
template <typename T, int D, int NFQ, int NFK>
__global__ __launch_bounds__(256) void k_attn_bwd(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int flat = xcd_order(blockIdx.x, gridDim.x), bh = flat / a.tiles, tile = flat % a.tiles;
  if (tile < a.nq_tiles) attn_bwd_q<T, D, NFQ>(a, smem, bh, tile);
  else attn_bwd_kv<T, D, NFK, (D * NFK < 256)>(a, smem, bh, tile - a.nq_tiles);
}

template <typename K>
int raise_lds(K kern, size_t smem, const char* name) {
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { dp_set_error("%s: cannot raise dynamic LDS to %zu: %s", name, smem, hipGetErrorString(e)); return 1; }
  }
  return 0;
}

int check_args(const char* name, const void* q, const void* k, const void* v, const void* o, int64_t ld, int64_t ldo, int B, int heads, int N,
               int d, int dtype) {
  if (dtype != DP_BF16 && dtype != DP_F16) DP_FAIL("%s: 16-bit storage only (dtype %d); fp32 uses dp_gemm_nt + dp_softmax_*", name, dtype);
  if (d != 64 && d != 128) DP_FAIL("%s: head dim %d not in {64, 128}", name, d);
  if (B <= 0 || heads <= 0 || N <= 0) DP_FAIL("%s: bad sizes B=%d heads=%d N=%d", name, B, heads, N);
  if ((ld & 7) || (ldo & 7) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15))
    DP_FAIL("%s: rows must be 16-byte aligned (ld %lld, ldo %lld)", name, (long long)ld, (long long)ldo);
  return 0;
}

template <typename T, int D, int NF>
int launch_fwd(const void* q, const void* k, const void* v, int64_t ld, void* o, int64_t ldo, float* lse, int B, int heads, int N, float scale,
               hipStream_t st) {
  auto kern = k_attn_fwd<T, D, NF>;
  constexpr size_t smem = AttSmem<D, NF>::fwd;
  if (raise_lds(kern, smem, "attention_fwd")) return 1;
  dim3 grid(cdiv(N, 16 * NF) * B * heads);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, (const T*)q, (const T*)k, (const T*)v, ld, (T*)o, ldo, lse, heads, N, scale * LOG2E);
  DP_CHECK_LAUNCH("attention_fwd");
  return 0;
}

template <typename T, int D, int NFQ, int NFK>
int launch_bwd(BwdArgs a, const void* o, int B, hipStream_t st) {
  const int64_t rows = (int64_t)B * a.heads * a.Np;
  constexpr int RPB = 256 / (D / 8);
  hipLaunchKernelGGL((k_attn_delta<T, D>), dim3(cdiv(rows, RPB)), dim3(256), 0, st, (const T*)o, (const T*)a.dO, a.ldo, (float*)a.delta, a.heads,
                     a.N, a.Np, rows);
  DP_CHECK_LAUNCH("attention_bwd(delta)");
  auto kern = k_attn_bwd<T, D, NFQ, NFK>;
  constexpr size_t sq = AttSmem<D, NFQ>::bwd_q, skv = AttSmem<D, NFK>::bwd_kv, smem = sq > skv ? sq : skv;
  if (raise_lds(kern, smem, "attention_bwd")) return 1;
  a.nq_tiles = cdiv(a.N, 16 * NFQ);
  a.tiles = a.nq_tiles + cdiv(a.N, 16 * NFK);
  hipLaunchKernelGGL(kern, dim3(a.tiles * B * a.heads), dim3(256), smem, st, a);
  DP_CHECK_LAUNCH("attention_bwd");
  return 0;
}

}  // namespace

template <int D>
int launch_fwd_x3(const void* q, const void* k, const void* v, int64_t ld, void* o, int64_t ldo, int B, int heads, int N, float scale, hipStream_t st) {
  using C = AttCfg<D>;
  auto kern = k_attn_fwd_x3<D>;
  constexpr size_t stage = (size_t)C::NW * 2 * C::CH * C::LDP * 2, comb = (size_t)C::NW * 16 * C::CP * 4;
  constexpr size_t smem = (stage > comb ? stage : comb) + (size_t)2 * C::NW * 16 * 4;
  static_assert(stage >= comb, "the combine tiles reuse the staging space; the statistics sit behind the staging space");
  if (raise_lds(kern, smem, "attention_fwd_x3")) return 1;
  hipLaunchKernelGGL(kern, dim3(cdiv(N, 16) * B * heads), dim3(256), smem, st, (const float*)q, (const float*)k, (const float*)v, ld, (float*)o, ldo, heads, N,
                     scale * LOG2E);
  DP_CHECK_LAUNCH("attention_fwd_x3");
  return 0;
}

extern "C" int dp_attention_fwd(const void* q, const void* k, const void* v, int64_t ld, void* o, int64_t ldo, float* lse, int B, int heads, int N,
                                int d, float scale, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DP_X3) {      // fp32 tensors, split-bf16 arithmetic (three products), fp32 output; lse is not written
    if (d != 64 && d != 128) DP_FAIL("attention_fwd: head dim %d not in {64, 128}", d);
    if (B <= 0 || heads <= 0 || N <= 0) DP_FAIL("attention_fwd: bad sizes B=%d heads=%d N=%d", B, heads, N);
    if ((ld & 3) || (ldo & 3) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)) DP_FAIL("attention_fwd (DP_X3): rows must be 16-byte aligned");
    return d == 64 ? launch_fwd_x3<64>(q, k, v, ld, o, ldo, B, heads, N, scale, st) : launch_fwd_x3<128>(q, k, v, ld, o, ldo, B, heads, N, scale, st);
  }
  if (check_args("attention_fwd", q, k, v, o, ld, ldo, B, heads, N, d, dtype)) return 1;
  if (dtype == DP_BF16) return d == 64 ? launch_fwd<bf16_t, 64, 4>(q, k, v, ld, o, ldo, lse, B, heads, N, scale, st)
                                       : launch_fwd<bf16_t, 128, 2>(q, k, v, ld, o, ldo, lse, B, heads, N, scale, st);
  return d == 64 ? launch_fwd<f16_t, 64, 4>(q, k, v, ld, o, ldo, lse, B, heads, N, scale, st)
                 : launch_fwd<f16_t, 128, 2>(q, k, v, ld, o, ldo, lse, B, heads, N, scale, st);
}

extern "C" int dp_attention_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* o, const void* go, int64_t ldo,
                                const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t ldg, int B, int heads, int N, int d,
                                float scale, int dtype, void* stream) {
  if (check_args("attention_bwd", q, k, v, o, ld, ldo, B, heads, N, d, dtype)) return 1;
  if ((ldg & 7) || (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)go | (uintptr_t)lse | (uintptr_t)delta) & 15))
    DP_FAIL("attention_bwd: gradient rows and the statistics must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  BwdArgs a{q, k, v, go, lse, delta, dq, dk, dv, ld, ldo, ldg, heads, N, (N + 31) & ~31, 0, 0, scale * LOG2E, scale};
  if (dtype == DP_BF16)
    return d == 64 ? launch_bwd<bf16_t, 64, 4, 2>(a, o, B, st) : launch_bwd<bf16_t, 128, 2, 2>(a, o, B, st);
  return d == 64 ? launch_bwd<f16_t, 64, 4, 2>(a, o, B, st) : launch_bwd<f16_t, 128, 2, 2>(a, o, B, st);
}
