// Batched "NT" GEMM on MFMA: C[m][n] = alpha * sum_k A[m][k] * B[n][k] (+ bias[n]).
// Both operands are k-contiguous, which is exactly the per-lane fragment shape of v_mfma_f32_16x16x32_bf16
// (lane l: row/col l&15, k = 8*(l>>4)..+7), so fragments are single 16-byte LDS reads.
// Block = 4 waves (2x2), block tile 128x128x32, wave tile 64x64 (4x4 MFMA tiles, 64 accumulator VGPRs).
// Global -> registers -> LDS with the next tile's loads in flight during the MFMA phase.
#include "common.h"
#include <stdlib.h>

#define STREAM ((hipStream_t)stream)
// BK: k-depth of one LDS stage (32 or 64).  LDP = padded LDS row (elements): keeps 16-B alignment, breaks the power-of-two stride.

#ifdef DP_GEMM_PROBE
__device__ unsigned long long dp_gemm_probe[40];      // tools/probes/gemm_probe.hip: s_memtime at block start, every K step, phases of step 2, loop end, block end
#define PROBE(i) do { if (probe) dp_gemm_probe[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PROBE(i) do { } while (0)
#endif
// TI x TJ = MFMA tiles per wave in M / N: block tile = (2*TI*16) x (2*TJ*16): 128x128 (4,4) for big problems, 64x64 (2,2)
// when a 128x128 grid would leave most of the 256 CUs idle (ViT token matrices: M = B*512).
template <typename T, int TI, int TJ, int BK>
__global__ void __launch_bounds__(256) k_gemm_nt(const T* __restrict__ A, int64_t lda, int64_t sa0, int64_t sa1,
                                                 const T* __restrict__ B, int64_t ldb, int64_t sb0, int64_t sb1,
                                                 void* __restrict__ Cv, int64_t ldc, int64_t sc0, int64_t sc1,
                                                 const float* __restrict__ bias, int M, int N, int K, int nb1,
                                                 float alpha, int out_f32, int splitk, T* __restrict__ aux, int64_t ldaux, int epi) {
  constexpr int BM = 32 * TI, BN = 32 * TJ, WMR = 16 * TI, WNR = 16 * TJ, LDP = BK + 8, CPR = BK / 8;   // CPR: 16-byte chunks per row
  constexpr int UA = BM * CPR / 256 > 0 ? BM * CPR / 256 : 1, UB = BN * CPR / 256 > 0 ? BN * CPR / 256 : 1;   // chunks per thread
  // fp32 results leave through an LDS tile too (round 4): NPF passes of BM / NPF rows x (BN + 4) floats
  constexpr int NPF = TI == 4 ? 2 : 1, CPF = BN + 4;
  constexpr int OPB = (BM + BN) * LDP * (int)sizeof(T), F32B = (BM / NPF) * CPF * 4, SMB = OPB > F32B ? OPB : F32B;
  __shared__ __attribute__((aligned(16))) char smem_raw[SMB];
  T* const smem = (T*)smem_raw;
  T* const As = smem; T* const Bs = smem + BM * LDP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * (32 * TI), n0 = blockIdx.y * (32 * TJ);
  const int bz = blockIdx.z / splitk, ks = blockIdx.z - bz * splitk;
  const int b0 = bz / nb1, b1 = bz - b0 * nb1;
  A += b0 * sa0 + b1 * sa1; B += b0 * sb0 + b1 * sb1;
  // K range of this split (multiples of BK)
  int ktiles = (K + BK - 1) / BK, per = (ktiles + splitk - 1) / splitk;
  int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);

  v4f acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  const int imax = min(TI, (M - m0 - wm * WMR + 15) / 16), jmax = min(TJ, (N - n0 - wn * WNR + 15) / 16);
  Frag8<T> ra[UA], rb[UB];
  // interior tiles of aligned operands: straight-line 16-byte loads (no per-chunk guards -> all of a stage's loads are in flight together)
  const bool fast = (BM * CPR) % 256 == 0 && (BN * CPR) % 256 == 0 && 256 % CPR == 0 && m0 + BM <= M && n0 + BN <= N && K % BK == 0 &&
                    ((lda | ldb) & 7) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((sa0 | sa1 | sb0 | sb1) & 7) == 0 &&
                    lda * BM < (1ll << 30) && ldb * BN < (1ll << 30);
  // fast path addressing: a wave-uniform 64-bit tile base per K step + 32-bit lane offsets fixed for the launch (issuing the 8
  // loads of a step used to cost ~500 cycles of 64-bit address arithmetic, a quarter of the step)
  const int g_row0 = tid / CPR, g_kc0 = (tid % CPR) * 8, g_rstep = 256 / CPR;
  const int offA = g_row0 * (int)lda + g_kc0, offB = g_row0 * (int)ldb + g_kc0, stepA = g_rstep * (int)lda, stepB = g_rstep * (int)ldb;
  const T* const tileA = A + (int64_t)m0 * lda; const T* const tileB = B + (int64_t)n0 * ldb;
  auto gload = [&](int kt) {
    if (fast) {
      const T* ab = tileA + kt * BK; const T* bb = tileB + kt * BK;
#pragma unroll
      for (int u = 0; u < UA; u++) ra[u] = frag_ld_lds(ab + (offA + u * stepA));
#pragma unroll
      for (int u = 0; u < UB; u++) rb[u] = frag_ld_lds(bb + (offB + u * stepB));
      return;
    }
#pragma unroll
    for (int u = 0; u < UA; u++) {
      int c = tid + u * 256, row = c / CPR, kc = c % CPR, k = kt * BK + kc * 8;
      int nv = K - k; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
      int gm = m0 + row;
      ra[u] = (row < BM && gm < M && nv > 0) ? frag_load(A + (int64_t)gm * lda + k, nv) : frag_zero<T>();
    }
#pragma unroll
    for (int u = 0; u < UB; u++) {
      int c = tid + u * 256, row = c / CPR, kc = c % CPR, k = kt * BK + kc * 8;
      int nv = K - k; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
      int gn = n0 + row;
      rb[u] = (row < BN && gn < N && nv > 0) ? frag_load(B + (int64_t)gn * ldb + k, nv) : frag_zero<T>();
    }
  };
#ifdef DP_GEMM_PROBE
  const bool probe = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0 && tid == 0;
#endif
  PROBE(0);
  if (kt0 < kt1) gload(kt0);
  for (int kt = kt0; kt < kt1; kt++) {
    if (kt - kt0 < 30) PROBE(1 + (kt - kt0));
    __syncthreads();
    if (kt - kt0 == 2) PROBE(33);
#pragma unroll
    for (int u = 0; u < UA; u++) {
      int c = tid + u * 256, row = c / CPR, kc = c % CPR;
      if (row < BM) frag_st_lds(As + row * LDP + kc * 8, ra[u]);
    }
#pragma unroll
    for (int u = 0; u < UB; u++) {
      int c = tid + u * 256, row = c / CPR, kc = c % CPR;
      if (row < BN) frag_st_lds(Bs + row * LDP + kc * 8, rb[u]);
    }
    __syncthreads();
    if (kt - kt0 == 2) PROBE(34);
    if (kt + 1 < kt1) gload(kt + 1);
    if (kt - kt0 == 2) PROBE(35);
    // all fragment reads of the step are issued before its first MFMA: one LDS latency per step instead of one per 32-deep
    // sub-step (reads + MFMAs took ~1000 cycles for 256 cycles of MFMA work)
    constexpr int NSUB = BK / 32;
    Frag8<T> fa[NSUB][TI], fb[NSUB][TJ];
#pragma unroll
    for (int ss = 0; ss < NSUB; ss++) {
#pragma unroll
      for (int i = 0; i < TI; i++) fa[ss][i] = frag_ld_lds(As + (wm * WMR + i * 16 + r) * LDP + ss * 32 + q * 8);
#pragma unroll
      for (int j = 0; j < TJ; j++) fb[ss][j] = frag_ld_lds(Bs + (wn * WNR + j * 16 + r) * LDP + ss * 32 + q * 8);
    }
#pragma unroll
    for (int ss = 0; ss < NSUB; ss++)
#pragma unroll
      for (int i = 0; i < TI; i++) {
        if (i >= imax) continue;      // wave-uniform: skip MFMA tiles that lie wholly outside M x N (skinny problems)
#pragma unroll
        for (int j = 0; j < TJ; j++) if (j < jmax) acc[i][j] = mma16(fa[ss][i], fb[ss][j], acc[i][j]);
      }
  }
  PROBE(31);
  // epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
  const int64_t coff = b0 * sc0 + b1 * sc1;
  // Interior tiles with a T output: transpose through LDS and write whole 16-byte chunks (2-byte scattered stores took
  // ~5800 cycles -- 30 % of a 1024 x 768 x 768 token GEMM).
  constexpr int CP = BN + 8;                                    // padded tile pitch (elements): 16-byte rows, no 4-way bank conflict
  constexpr int NPASS = (BM * CP <= (BM + BN) * LDP) ? 1 : 2;   // the two wave rows take turns when the whole tile does not fit
  constexpr bool CAN_STAGE = sizeof(T) == 2 && (BM / NPASS) * CP <= (BM + BN) * LDP;
  if (CAN_STAGE && !out_f32 && splitk == 1 && m0 + BM <= M && n0 + BN <= N && (ldc & 7) == 0 && ((sc0 | sc1) & 7) == 0 &&
      (((uintptr_t)Cv) & 15) == 0) {
    T* tile = smem;
    T* Cb = (T*)Cv + coff + (int64_t)m0 * ldc + n0;
    constexpr int PR = BM / NPASS;                              // rows per pass
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      __syncthreads();                                          // operand tiles / previous pass no longer needed
      if (NPASS == 1 || wm == ps) {
        const int rbase = NPASS == 1 ? wm * WMR : 0;
#pragma unroll
        for (int j = 0; j < TJ; j++) {
          const int col = wn * WNR + j * 16 + r;
          const float bv = bias ? bias[n0 + col] : 0.f;
#pragma unroll
          for (int i = 0; i < TI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              float v = alpha * acc[i][j][e] + bv;
              // epi 2: the product is the gradient of GELU's OUTPUT; aux holds the pre-activation (fc2 data gradient -> fc1 output gradient)
              if (epi == 2) v *= act_bwd(ld_f(aux + (int64_t)(m0 + (NPASS == 1 ? 0 : ps * PR) + rbase + i * 16 + q * 4 + e) * ldaux + n0 + col), DP_ACT_GELU);
              st_f(tile + (rbase + i * 16 + q * 4 + e) * CP + col, v);
            }
        }
      }
      __syncthreads();
      if (epi == 1) {
        // fc1 + GELU: the tile holds the pre-activation rounded to T; it goes to aux as it is and to C through GELU -- the values
        // a separate GELU kernel would read and write
        T* Ab = aux + (int64_t)m0 * ldaux + n0;
#pragma unroll
        for (int u = 0; u < PR * BN / 8 / 256; u++) {
          const int c = tid + u * 256, row = c / (BN / 8), cc = (c % (BN / 8)) * 8;
          if constexpr (sizeof(T) == 2) {
            union { v4u raw; T e[8]; } w;
            w.raw = *(const v4u*)(tile + row * CP + cc);
            *(v4u*)(Ab + (int64_t)(ps * PR + row) * ldaux + cc) = w.raw;
#pragma unroll
            for (int z = 0; z < 8; z++) st_f(&w.e[z], act_fwd(ld_f(&w.e[z]), DP_ACT_GELU));
            *(v4u*)(Cb + (int64_t)(ps * PR + row) * ldc + cc) = w.raw;
          }
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < PR * BN / 8 / 256; u++) {
        const int c = tid + u * 256, row = c / (BN / 8), cc = (c % (BN / 8)) * 8;
        *(v4u*)(Cb + (int64_t)(ps * PR + row) * ldc + cc) = *(const v4u*)(tile + row * CP + cc);
      }
    }
    PROBE(32);
    return;
  }
  // Interior tiles with an fp32 output (the fp32x3 mode's Linear layers, the exact-fp32 ConvTranspose / attention GEMMs): the same
  // transposition through LDS, 16-byte stores of four floats (the scattered version wrote 4-byte elements 64 bytes at a time: the
  // 524 288 x 128 x 32 ConvTranspose GEMM of the fp32 modes ran at 1.3 TB/s of output)
  if ((out_f32 || sizeof(T) == 4) && splitk == 1 && epi == 0 && m0 + BM <= M && n0 + BN <= N && (ldc & 3) == 0 && ((sc0 | sc1) & 3) == 0 &&
      (((uintptr_t)Cv) & 15) == 0) {
    float* tile = (float*)smem_raw;
    float* Cb = (float*)Cv + coff + (int64_t)m0 * ldc + n0;
    constexpr int PR = BM / NPF;
#pragma unroll
    for (int ps = 0; ps < NPF; ps++) {
      __syncthreads();
      if (NPF == 1 || wm == ps) {
        const int rbase = NPF == 1 ? wm * WMR : 0;
#pragma unroll
        for (int j = 0; j < TJ; j++) {
          const int col = wn * WNR + j * 16 + r;
          const float bv = bias ? bias[n0 + col] : 0.f;
#pragma unroll
          for (int i = 0; i < TI; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) tile[(rbase + i * 16 + q * 4 + e) * CPF + col] = alpha * acc[i][j][e] + bv;
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PR * BN / 4 / 256; u++) {
        const int c = tid + u * 256, row = c / (BN / 4), cc = (c % (BN / 4)) * 4;
        *(v4f*)(Cb + (int64_t)(ps * PR + row) * ldc + cc) = *(const v4f*)(tile + row * CPF + cc);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TJ; j++) {
    int col = n0 + wn * WNR + j * 16 + r;
    if (col >= N) continue;
    float bv = (bias && ks == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; i++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int row = m0 + wm * WMR + i * 16 + q * 4 + e;
        if (row >= M) continue;
        float v = alpha * acc[i][j][e] + bv;
        int64_t idx = coff + (int64_t)row * ldc + col;
        if (epi == 1) { T h; st_f(&h, v); aux[(int64_t)row * ldaux + col] = h; v = act_fwd(ld_f(&h), DP_ACT_GELU); }
        else if (epi == 2) v *= act_bwd(ld_f(aux + (int64_t)row * ldaux + col), DP_ACT_GELU);
        if (out_f32) { if (splitk > 1) atomicAdd((float*)Cv + idx, v); else ((float*)Cv)[idx] = v; }
        else st_f((T*)Cv + idx, v);
      }
    }
  }
}

static int gemm_nt_impl(const void* A, int64_t lda, int64_t sa0, int64_t sa1, const void* B, int64_t ldb, int64_t sb0, int64_t sb1,
                        void* C, int64_t ldc, int64_t sc0, int64_t sc1, const float* bias, int M, int N, int K, int nb0, int nb1,
                        float alpha, int out_f32, int splitk, void* aux, int64_t ldaux, int epi, int dtype, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) DP_FAIL("gemm_nt: empty problem %d %d %d", M, N, K);
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && !out_f32) DP_FAIL("gemm_nt: split-K needs fp32 (atomic) output");
  if (dp_det(DET_SPLITK)) splitk = 1;            // deterministic mode: no atomic meeting of K shares (the whole K axis in one block per tile)
  int64_t big_blocks = (int64_t)cdiv(M, 128) * cdiv(N, 128) * nb0 * nb1 * splitk;
  bool small = big_blocks < 256 && N > 32;        // 64x64 tiles: 4x the blocks (skinny N keeps the 128-row tile: it is a row stream)
  // (a 128 x 64 tile for the qkv / fc1 token GEMMs -- 288-384 blocks, a third less L2 -> LDS traffic -- was measured in round 4: 16.8 / 18.0 us
  // against 14.9 / 15.7 for the 64 x 64 tiles: these launches are latency-, not traffic-bound)
  // (32 x 64 tiles for the 192-block out-projection / fc2 GEMMs: measured 11.6 / 29.7 us against 10.0 / 24.5 -- round 4)
  int bm = small ? 64 : 128, bn = small ? 64 : 128;
  dim3 g(cdiv(M, bm), cdiv(N, bn), nb0 * nb1 * splitk);
  if (g.y > 65535 || g.z > 65535) DP_FAIL("gemm_nt: grid too large");
  // bf16 and deep K: 64-deep LDS stages halve the number of barrier pairs (the token GEMMs are latency-, not MFMA-bound)
  bool deep = dtype != DP_F32 && K >= 256;
#define GEMM_ARGS g, dim3(256), 0, STREAM, (const T*)A, lda, sa0, sa1, (const T*)B, ldb, sb0, sb1, C, ldc, sc0, sc1, bias, M, N, K, nb1, alpha, out_f32, splitk, (T*)aux, ldaux, epi
#define GEMM_GO(TI_, TJ_, BK_) DP_DISPATCH(dtype, hipLaunchKernelGGL((k_gemm_nt<T, TI_, TJ_, BK_>), GEMM_ARGS))
  if (deep && dtype == DP_BF16) {   // 16-bit types only: the deeper stage keeps (BM + BN) * BK * 2 bytes of loads in flight per block
    typedef bf16_t T;
    if (small) hipLaunchKernelGGL((k_gemm_nt<T, 2, 2, 128>), GEMM_ARGS); else hipLaunchKernelGGL((k_gemm_nt<T, 4, 4, 64>), GEMM_ARGS);
  } else if (deep) {
    typedef f16_t T;
    if (small) hipLaunchKernelGGL((k_gemm_nt<T, 2, 2, 128>), GEMM_ARGS); else hipLaunchKernelGGL((k_gemm_nt<T, 4, 4, 64>), GEMM_ARGS);
  } else if (small) GEMM_GO(2, 2, 32);
  else GEMM_GO(4, 4, 32);
#undef GEMM_ARGS
#undef GEMM_GO
  DP_CHECK_LAUNCH("gemm_nt"); return 0;
}
extern "C" int dp_gemm_nt(const void* A, int64_t lda, int64_t sa0, int64_t sa1, const void* B, int64_t ldb, int64_t sb0, int64_t sb1,
                          void* C, int64_t ldc, int64_t sc0, int64_t sc1, const float* bias, int M, int N, int K, int nb0, int nb1,
                          float alpha, int out_f32, int splitk, int dtype, void* stream) {
  return gemm_nt_impl(A, lda, sa0, sa1, B, ldb, sb0, sb1, C, ldc, sc0, sc1, bias, M, N, K, nb0, nb1, alpha, out_f32, splitk, nullptr, 0, 0, dtype, stream);
}
// Token GEMM with the transformer MLP's GELU in its epilogue (one matrix, storage-type output):
//   mode 1: aux = A B^T + bias (pre-activation, kept for the backward pass), C = GELU(aux)
//   mode 2: C = (A B^T) * GELU'(aux)             (the data gradient of the layer AFTER the GELU, taken through it)
extern "C" int dp_gemm_nt_gelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, void* aux,
                               int64_t ldaux, int M, int N, int K, int mode, int dtype, void* stream) {
  if (mode != 1 && mode != 2) DP_FAIL("gemm_nt_gelu: mode must be 1 (forward) or 2 (backward)");
  if (!aux) DP_FAIL("gemm_nt_gelu: the pre-activation matrix is missing");
  return gemm_nt_impl(A, lda, 0, 0, B, ldb, 0, 0, C, ldc, 0, 0, mode == 1 ? bias : nullptr, M, N, K, 1, 1, 1.f, 0, 1, aux, ldaux, mode, dtype, stream);
}

// ------------------------------------------------------------------------------------------------ TN GEMM
// C[m][n] (fp32) = sum_k A[k][m] * B[k][n]: both operands are k-major in MEMORY (rows = k), which is what every weight
// gradient of a row-major layer looks like (dW[out][in] = gy^T x with k = token / voxel rows).  The 64-row k slabs are
// staged as they lie; the MFMA fragments (8 consecutive k of one column per lane) come out of LDS through
// ds_read_b64_tr_b16 (fp32: scalar column reads).  Replaces two k_transpose launches + k_gemm_nt per weight gradient.
// One 64 x 64 output tile of C = A^T B (the body shared by k_gemm_tn and k_gemm_tn_grouped).  colsum != nullptr: the tile also
// leaves colsum[m] = sum_k A[k][m] for its 64 columns m (the bias gradient of a Linear layer: A = gy), from the staged A tiles.
template <typename T>
__device__ __forceinline__ void gemm_tn_tile(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                             int64_t ldc, int M, int N, int K, int splitk, int m0, int n0, int ks, float* __restrict__ colsum,
                                             T* As, T* Bs) {
  constexpr int BT = 64, BKT = 64, LDPT = BT + 8, CPRT = BT / 8, UT = BKT * CPRT / 256;     // 72-element LDS rows; 2 chunks per thread per operand
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1, r = lane & 15, q = lane >> 4;
  const int ktiles = (K + BKT - 1) / BKT, per = (ktiles + splitk - 1) / splitk, kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  v4f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  Frag8<T> ra[UT], rb[UT];
  float csum = 0.f;
  const bool fast = m0 + BT <= M && n0 + BT <= N && ((lda | ldb) & 7) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  auto gload = [&](int kt) {
#pragma unroll
    for (int u = 0; u < UT; u++) {
      const int c = tid + u * 256, row = c / CPRT, cc = (c % CPRT) * 8, k = kt * BKT + row;
      if (fast && k < K) { ra[u] = frag_ld_lds(A + (int64_t)k * lda + m0 + cc); rb[u] = frag_ld_lds(B + (int64_t)k * ldb + n0 + cc); }
      else {
        int nva = M - m0 - cc, nvb = N - n0 - cc; nva = nva < 0 ? 0 : (nva > 8 ? 8 : nva); nvb = nvb < 0 ? 0 : (nvb > 8 ? 8 : nvb);
        ra[u] = (k < K && nva > 0) ? frag_load(A + (int64_t)k * lda + m0 + cc, nva) : frag_zero<T>();
        rb[u] = (k < K && nvb > 0) ? frag_load(B + (int64_t)k * ldb + n0 + cc, nvb) : frag_zero<T>();
      }
    }
  };
  // lane part of the k-major fragment address: MFMA k group q = lane>>4 (8 consecutive k), column r = lane&15
  const int i16 = lane & 15, tr_lane = (8 * q + (i16 >> 2)) * LDPT + 4 * (i16 & 3);
  auto frag = [&](const T* img, int kk, int col0) {
    if constexpr (sizeof(T) == 2) return tr_pair<4 * LDPT, T>(img + kk * LDPT + col0 + tr_lane);
    else { Frag8<float> f;
#pragma unroll
      for (int j = 0; j < 8; j++) f.v[j] = img[(kk + 8 * q + j) * LDPT + col0 + r];
      return f; }
  };
  if (kt0 < kt1) gload(kt0);
  for (int kt = kt0; kt < kt1; kt++) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UT; u++) {
      const int c = tid + u * 256, row = c / CPRT, cc = (c % CPRT) * 8;
      frag_st_lds(As + row * LDPT + cc, ra[u]); frag_st_lds(Bs + row * LDPT + cc, rb[u]);
    }
    __syncthreads();
    if (kt + 1 < kt1) gload(kt + 1);
    if (colsum) {                                  // block-uniform: thread t adds rows (t >> 6) * 16 .. + 15 of column t & 63
      const int cc = tid & 63, r0 = (tid >> 6) * 16;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) csum += ld_f(As + (r0 + rr) * LDPT + cc);
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 32) {
      Frag8<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) { fa[i] = frag(As, kk, wm * 32 + i * 16); fb[i] = frag(Bs, kk, wn * 32 + i * 16); }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    }
  }
  // C/D layout: col (n) = lane&15, row (m) = 4*(lane>>4) + reg
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int col = n0 + wn * 32 + j * 16 + r;
    if (col >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int row = m0 + wm * 32 + i * 16 + q * 4 + e;
        if (row >= M) continue;
        if (splitk > 1) atomicAdd(C + (int64_t)row * ldc + col, acc[i][j][e]); else C[(int64_t)row * ldc + col] = acc[i][j][e];
      }
  }
  if (colsum) {
    __syncthreads();
    float* red = (float*)As;
    red[tid] = csum;
    __syncthreads();
    if (tid < 64 && m0 + tid < M) colsum[m0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
  }
}
// The same for a 128 x 128 output tile (round 4): a 64 x 64 tile fetches 16 KB of operands per 0.5 MFLOP, and the grouped launch of the
// transformer's + patch embedding's weight gradients (33 k tiles x 16 k-slabs) pulled 8.4 GB through L2 -> LDS for 0.3 GB of operands:
// it ran at the L2's rate (0.72 ms), not the matrix cores' (16.7 % MFMA-busy).  Four waves x (64 x 64) per block halve that traffic.
// Whole tiles and aligned rows only (M, N multiples of 128, 16-byte rows): the caller picks the tile per problem.
template <typename T>
__device__ __forceinline__ void gemm_tn_tile128(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                int64_t ldc, int K, int m0, int n0, float* __restrict__ colsum, T* As, T* Bs) {
  constexpr int BT = 128, BKT = 64, LDPT = BT + 8, CPRT = BT / 8, UT = BKT * CPRT / 256;     // 136-element LDS rows; 4 chunks per thread per operand
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1, r = lane & 15, q = lane >> 4;
  const int ktiles = (K + BKT - 1) / BKT;
  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  Frag8<T> ra[UT], rb[UT];
  float csum = 0.f;
  auto gload = [&](int kt) {
#pragma unroll
    for (int u = 0; u < UT; u++) {
      const int c = tid + u * 256, row = c / CPRT, cc = (c % CPRT) * 8, k = kt * BKT + row;
      if (k < K) { ra[u] = frag_ld_lds(A + (int64_t)k * lda + m0 + cc); rb[u] = frag_ld_lds(B + (int64_t)k * ldb + n0 + cc); }
      else { ra[u] = frag_zero<T>(); rb[u] = frag_zero<T>(); }
    }
  };
  const int i16 = lane & 15, tr_lane = (8 * q + (i16 >> 2)) * LDPT + 4 * (i16 & 3);
  auto frag = [&](const T* img, int kk, int col0) {
    if constexpr (sizeof(T) == 2) return tr_pair<4 * LDPT, T>(img + kk * LDPT + col0 + tr_lane);
    else { Frag8<float> f;
#pragma unroll
      for (int j = 0; j < 8; j++) f.v[j] = img[(kk + 8 * q + j) * LDPT + col0 + r];
      return f; }
  };
  gload(0);
  for (int kt = 0; kt < ktiles; kt++) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UT; u++) {
      const int c = tid + u * 256, row = c / CPRT, cc = (c % CPRT) * 8;
      frag_st_lds(As + row * LDPT + cc, ra[u]); frag_st_lds(Bs + row * LDPT + cc, rb[u]);
    }
    __syncthreads();
    if (kt + 1 < ktiles) gload(kt + 1);
    if (colsum) {                                  // block-uniform: thread t adds rows (t >> 7) * 32 .. + 31 of column t & 127
      const int cc = tid & 127, r0 = (tid >> 7) * 32;
#pragma unroll
      for (int rr = 0; rr < 32; rr++) csum += ld_f(As + (r0 + rr) * LDPT + cc);
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 32) {
      Frag8<T> fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { fa[i] = frag(As, kk, wm * 64 + i * 16); fb[i] = frag(Bs, kk, wn * 64 + i * 16); }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int col = n0 + wn * 64 + j * 16 + r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) C[(int64_t)(m0 + wm * 64 + i * 16 + q * 4 + e) * ldc + col] = acc[i][j][e];
  }
  if (colsum) {
    __syncthreads();
    float* red = (float*)As;
    red[tid] = csum;
    __syncthreads();
    if (tid < 128) colsum[m0 + tid] = red[tid] + red[128 + tid];
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k_gemm_tn(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                 int64_t ldc, int M, int N, int K, int splitk) {
  __shared__ __attribute__((aligned(16))) T As[64 * 72];
  __shared__ __attribute__((aligned(16))) T Bs[64 * 72];
  gemm_tn_tile<T>(A, lda, B, ldb, C, ldc, M, N, K, splitk, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z, nullptr, As, Bs);
}
// Grouped form: ONE launch computes the weight (and bias) gradients of many Linear layers (the 8 transformer layers x 4 weights
// + the patch embedding: every one a separate, latency-bound 25-us launch before).  table row p (12 x int64): A, B, C, colsum
// pointers, lda, ldb, ldc, M, N, K, first tile id, tiles along M (| 1 << 32: 128 x 128 tiles instead of 64 x 64: M, N multiples of
// 128, lda / ldb multiples of 8, 16-byte aligned operands).  Tile ids are dealt problem-major, n-tile-major, m fastest.
struct TnProblem { const void* A; const void* B; float* C; float* colsum; int64_t lda, ldb, ldc, M, N, K, tile0, tiles_m; };
// Three blocks per CU for the 16-bit types (round 6): a 128 x 128 tile is 16 K steps of one register-prefetched slab each, i.e. bound by the
// round trip of its loads; at 172 VGPRs two blocks shared a CU, at 164 (no spills) three: 648 -> 399 us for the transformer's 41 problems.
template <typename T>
__global__ void __launch_bounds__(256, sizeof(T) == 2 ? 3 : 2) k_gemm_tn_grouped(const TnProblem* __restrict__ tab, int nprob, int64_t total_tiles) {
  __shared__ __attribute__((aligned(16))) T As[64 * 136];
  __shared__ __attribute__((aligned(16))) T Bs[64 * 136];
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, consecutive TILES share an operand panel (the row tiles
  // of one 64-column panel of B: for the patch embedding 12 tiles x 128 KB of x).  Give every XCD a contiguous range of tiles, so
  // that a panel is fetched into ONE L2 instead of eight (FETCH_SIZE of the launch: 2.6 GB for 0.3 GB of operands, round 3).
  // (the grid is padded to 8 x ceil(total / 8) blocks so that the map is a bijection; ids >= total leave)
  int64_t bid = blockIdx.x;
  {
    const int64_t per = gridDim.x >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= total_tiles) return;
  }
  int p = 0;
  while (p + 1 < nprob && tab[p + 1].tile0 <= bid) p++;          // (uniform scalar loads; <= a few dozen problems)
  const TnProblem P = tab[p];
  const int t = (int)(bid - P.tile0), tm = (int)(P.tiles_m & 0xffffffff), mi = t % tm, ni = t / tm;
  if (P.tiles_m >> 32) {      // (bit 32 of tiles_m: the problem was tiled 128 x 128 -- whole tiles, aligned rows)
    gemm_tn_tile128<T>((const T*)P.A, P.lda, (const T*)P.B, P.ldb, P.C, P.ldc, (int)P.K, mi * 128, ni * 128, (P.colsum && ni == 0) ? P.colsum : nullptr, As, Bs);
    return;
  }
  gemm_tn_tile<T>((const T*)P.A, P.lda, (const T*)P.B, P.ldb, P.C, P.ldc, (int)P.M, (int)P.N, (int)P.K, 1, mi * 64, ni * 64, 0,
                  (P.colsum && ni == 0) ? P.colsum : nullptr, As, Bs);
}
extern "C" int dp_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int splitk,
                          int dtype, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) DP_FAIL("gemm_tn: empty problem %d %d %d", M, N, K);
  if (splitk < 1 || dp_det(DET_SPLITK)) splitk = 1;      // (deterministic mode: unsplit)
  dim3 g(cdiv(M, 64), cdiv(N, 64), splitk);
  if (g.y > 65535 || g.z > 65535) DP_FAIL("gemm_tn: grid too large");
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_gemm_tn<T>, g, dim3(256), 0, STREAM, (const T*)A, lda, (const T*)B, ldb, C, ldc, M, N, K, splitk));
  DP_CHECK_LAUNCH("gemm_tn"); return 0;
}
extern "C" int dp_gemm_tn_grouped(const void* table, int nproblems, int64_t total_tiles, int dtype, void* stream) {
  if (nproblems <= 0 || total_tiles <= 0) return 0;
  if (total_tiles > 2000000000LL) DP_FAIL("gemm_tn_grouped: too many tiles");
  const unsigned grid = (unsigned)(((total_tiles + 7) >> 3) << 3);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_gemm_tn_grouped<T>, dim3(grid), dim3(256), 0, STREAM, (const TnProblem*)table, nproblems, total_tiles));
  DP_CHECK_LAUNCH("gemm_tn_grouped"); return 0;
}
