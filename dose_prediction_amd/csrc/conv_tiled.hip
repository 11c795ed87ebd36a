// LDS-tiled implicit-GEMM convolution for the hot layers: stride 1, dilation 1, "same" padding, k in {3, 7}.
// Used for the forward pass and (with transposed + flipped weights) for the data gradient.
//
// MFMA: v_mfma_f32_32x32x16_bf16 (f32: 8 x v_mfma_f32_32x32x2_f32 on the same 8-wide fragments).
//   M = 32 consecutive output positions along W,  K = 16 input channels (one channel chunk),
//   N = 32 output columns = 32 output channels, or -- when Cout <= 16 -- TWO vertically adjacent taps (kh = 2j, 2j+1)
//       x 16 channels ("tap pairing"): with only 16 output channels every A fragment would feed a single 16-wide MFMA
//       and the kernel would be bound by LDS bandwidth (1 KiB of LDS reads per 16-cycle MFMA per SIMD = the full
//       256 B/clk of the CU).  Pairing two taps in N halves the LDS traffic per FLOP.  The odd tap of a pair belongs to
//       the output row above, so a wave accumulates RW = RWO+1 rows and combines Z0[row t] + Z1[row t+1] (a 16-lane
//       shuffle) once in the epilogue.
// Block = 4 waves arranged TWC (32-position columns) x TRG (row groups); it owns one depth slice d, TRG*RWO output rows
// and TWC*32 positions.  For each kd and each 16-channel chunk the input slab [rows+halo][positions+halo][16] is staged
// in LDS (zero-filled outside the volume) and every wave sweeps its (kh, kw) taps: B fragments (packed weights) come
// straight from global/L2 (one 16-byte load per lane per tap, reused over RW rows), A fragments are single
// ds_read_b128 of 8 consecutive channels of one voxel.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#ifndef DP_TILED_TAB
#define DP_TILED_TAB 1        // k_conv_tiled: slab staging from a per-block table of voxel indices + buffer loads (round 6); 0: the round-2 form
#endif
#ifndef STAGE_UNROLL
#define STAGE_UNROLL 9
#endif
#define STREAM ((hipStream_t)stream)
typedef float v16f __attribute__((ext_vector_type(16)));
// Scratch contract (dp_scratch_contract): 0 = the fp32 scratch `ws` of dp_conv3d_tiled* (split-kd) and dp_conv3d_wgrad_tiled* may hold
// anything and is cleared with a memset launch per call; 1 = the caller guarantees it is all zero on entry, and the finish /
// unpack kernels put zeros back while they read it (a persistent scratch buffer then never needs a memset).
static int g_scratch_zeroed = 0;
extern "C" int dp_scratch_contract(int zeroed) { g_scratch_zeroed = zeroed ? 1 : 0; return 0; }
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v16f mma32(const Frag8<bf16_t>& a, const Frag8<bf16_t>& b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a.u), __builtin_bit_cast(v8bf, b.u), c, 0, 0, 0);
}
__device__ __forceinline__ v16f mma32(const Frag8<f16_t>& a, const Frag8<f16_t>& b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a.u), __builtin_bit_cast(v8h, b.u), c, 0, 0, 0);
}
__device__ __forceinline__ v16f mma32(const Frag8<float>& a, const Frag8<float>& b, v16f c) {
#pragma unroll
  for (int j = 0; j < 8; j++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}

struct TiledGeom {
  int N, D, H, W, Cin, Cout, ldx, ldy;
  int TWC, TRG;        // wave arrangement (TWC*TRG == 4)
  int LR, LP;          // slab rows / positions (with halo)
  int NCH;             // 16-channel chunks of Cin
  int NTT;             // total 32-column N tiles in the packed weights
  int tiles_h, tiles_w;
  int dbg;             // experiments only (env DP_DBG): 1 = skip staging, 2 = skip the MFMA sweep, 3 = both, 4 = also skip the epilogue
  int splitkd;         // 1: blockIdx.z selects ONE kd (and one of `chsplit` shares of the input chunks); results are atomically accumulated into the fp32 scratch `ws`
  int chsplit;
  int64_t det_slab;    // deterministic mode: every (kd, chunk share) = blockIdx.z accumulates into ITS OWN slab of the scratch (det_slab elements apart; one contributor per address) and k_conv_split_finish adds the slabs in order; 0 = one shared scratch
  // "virtual concat": input channels >= csplit come from x2 (pitch ldx2), output channels >= osplit go to y2 (pitch ldy2).
  // torch.cat((a, b), dim=1) feeding a convolution is never materialised -- and each 16-channel chunk pass then reads whole,
  // contiguous voxel rows of ONE tensor (a 32-channel row read 16 channels at a time touches every cache line twice and
  // held the L2 hit rate of the 32->16 7x7x7 layer at ~25 % against ~90 % for 16-channel inputs).
  const void* x2; void* y2; int ldx2, csplit, ldy2, osplit;
  // Normalisation statistics of the OUTPUT, taken in the epilogue from the accumulators rounded to the storage type (the
  // InstanceNorm / BatchNorm that follows every convolution of the path then needs no read pass over y, and its mean / variance
  // are exactly those of the tensor it reads -- the same numbers the row pass dp_stats_partial(y) would produce): block (n, d, tile_h, tile_w) writes its per-channel (sum, sum of squares) over the voxels it owns to
  // stat_part[((n * stat_nblk + blk) * 2 + {0,1}) * Cout + c]; the partial rows are combined in fp64 by dp_stats_finalize.
  // Only with the wide (16-byte) epilogue and without split-kd (dp_conv3d_tiled_stat_blocks tells).
  float* stat_part; int stat_nblk; int wide;
  // DP_X3 launches (see Cc16Geom::x3 in conv_cc16.hip): the input holds 2 * x3 chunks [x_hi | x_lo], the packed weights NCH = 3 * x3
  // chunks [w_hi | w_hi | w_lo]; an x_hi slab is staged once and swept with weight chunks ch and 2 * x3 + ch.  0 = ordinary launch.
  int x3;
};

// ---------------------------------------------------------------------------------------------- weight packing
// dst[kd][jh][kw][chunk][ntile][col 32][ci 16];  NPAIR==2: col = s*16 + co, kh = 2*jh + s;  NPAIR==1: kh = jh, co = ntile*32 + col.
// transposed_flipped != 0: use w'[co'][ci'][t'] = w[ci'][co'][taps-1-t'] (data gradient as a forward convolution); then
// "Cout"/"Cin" below are the roles in the convolution being computed.
template <typename T>
__global__ void k_pack_w_tiled(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int KS, int NPAIR, int tf) {
  int JH = NPAIR == 2 ? (KS + 1) / 2 : KS, NCH = (Cin + 15) / 16, NTT = (Cout * NPAIR + 31) / 32, taps = KS * KS * KS;
  int64_t total = (int64_t)KS * JH * KS * NCH * NTT * 512;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i & 15), col = (int)((i >> 4) & 31); int64_t t = i >> 9;
    int nt = (int)(t % NTT); t /= NTT; int ch = (int)(t % NCH); t /= NCH; int kw = (int)(t % KS); t /= KS;
    int jh = (int)(t % JH); int kd = (int)(t / JH);
    int kh, co;
    if (NPAIR == 2) { int s = col >> 4; kh = 2 * jh + s; co = nt * 16 + (col & 15); }   // NTT == 1 when paired
    else { kh = jh; co = nt * 32 + col; }
    int ci = ch * 16 + c;
    float v = 0.f;
    if (kh < KS && co < Cout && ci < Cin) {
      int tap = (kd * KS + kh) * KS + kw;
      v = tf ? w[((int64_t)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * Cin + ci) * taps + tap];
    }
    st_f(dst + i, v);
  }
}

// configuration chosen from (Cout): returns NPAIR, and RW / NT through pointers
static inline int tiled_config(int Cout, int* rw, int* nt) {
  if (Cout <= 16) { *rw = 9; *nt = 1; return 2; }
  if (Cout <= 32) { *rw = 8; *nt = 1; return 1; }
  *rw = 4; *nt = 2; return 1;
}
static inline bool tiled_applicable(int Cin, int Cout, int k, int stride, int pad, int dil, int W) {
  // (W >= 8: the 12^3 level of OAR-TRANSEG's 96^3 sliding-window crop must not fall to the generic kernel: 2.4-4.9 ms per launch)
  return (k == 3 || k == 7) && stride == 1 && dil == 1 && pad == k / 2 && W >= 8 && Cin >= 1 && Cout >= 8;
}

// 0: the shape takes the generic kernel; 1: k_conv_tiled layout (dp_pack_conv_weight_tiled); 2: k_conv_cc16 layout
// (dp_pack_conv_weight_cc16).  A function of the shape alone, so the packed copies can be cached per (Cin, Cout, k, W class).
extern "C" int dp_conv3d_tiled_layout(int Cin, int Cout, int k, int stride, int pad, int dil, int W) {
  if (!tiled_applicable(Cin, Cout, k, stride, pad, dil, W)) return 0;
  return cc16_applicable(Cin, Cout, k, W) ? 2 : 1;
}
extern "C" int dp_pack_conv_weight_cc16(const float* w, void* dst, int Cout, int Cin, int k, int transposed_flipped, int dtype, void* stream) {
  return cc16_pack(w, dst, Cout, Cin, k, transposed_flipped, dtype, STREAM);
}
extern "C" int dp_conv3d_tiled_weight_elems(int Cin, int Cout, int k, int stride, int pad, int dil, int W) {
  if (!tiled_applicable(Cin, Cout, k, stride, pad, dil, W)) return 0;
  if (cc16_applicable(Cin, Cout, k, W)) return cc16_weight_elems(Cin, Cout, k);
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  int JH = np == 2 ? (k + 1) / 2 : k;
  return k * JH * k * ((Cin + 15) / 16) * ((Cout * np + 31) / 32) * 512;
}
extern "C" int dp_conv3d_tiled_npair(int Cout) { int rw, nt; return tiled_config(Cout, &rw, &nt); }
extern "C" int dp_pack_conv_weight_tiled(const float* w, void* dst, int Cout, int Cin, int k, int transposed_flipped, int dtype, void* stream) {
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  int JH = np == 2 ? (k + 1) / 2 : k;
  int64_t total = (int64_t)k * JH * k * ((Cin + 15) / 16) * ((Cout * np + 31) / 32) * 512;
  int g = (int)((total + 255) / 256); if (g > 8192) g = 8192;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_w_tiled<T>, dim3(g), dim3(256), 0, STREAM, w, (T*)dst, Cout, Cin, k, np, transposed_flipped));
  DP_CHECK_LAUNCH("pack_conv_weight_tiled"); return 0;
}

// ---------------------------------------------------------------------------------------------- the kernel
// TWP: wave columns (1, 2 or 4 x 32 positions; the other waves stack as row groups), or 0 = "W16": volumes with W <= 16,
// where one 32-row MFMA tile covers TWO image rows x 16 positions (NPAIR == 1 only).  A template parameter so that the slab
// pitch is a compile-time constant: every LDS row address is then one base register + an immediate offset.
// TO: type of the output tensor (T, or float for DP_X3 launches: bf16 hi / lo operand halves, fp32 result -- see conv_cc16.hip).
// WN: waves along N.  1: the four waves are TWC columns x TRG row groups over the SAME NT channel tiles.  2 (W16 tiles on planes of
// <= 16 rows, Cout >= 128: the 16^3 level of the decoder): two row groups x two channel-tile groups -- with four row groups of
// 4 x 2 image rows half of every block's MFMAs fell on rows >= H (7^3 at 16^3: 0.67 PFLOP/s against 1.1-1.4 at the other levels).
#ifndef DP_TILED_MINB8
#define DP_TILED_MINB8 2
#endif
#ifndef DP_TILED_MINB4
#define DP_TILED_MINB4 2
#endif
template <typename T, int KS, int NPAIR, int RW, int NT, int TWP, typename TO, int WN = 1>
__global__ void __launch_bounds__(256, (RW == 8 ? DP_TILED_MINB8 : DP_TILED_MINB4)) k_conv_tiled(const T* __restrict__ x, const T* __restrict__ wq, const float* __restrict__ bias,
                                                    TO* __restrict__ y, float* __restrict__ ws, TiledGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* slab = (T*)smem_raw;
  constexpr int PAD = KS / 2, JH = NPAIR == 2 ? (KS + 1) / 2 : KS, RWO = RW - (NPAIR - 1), NTAP = JH * KS, CK = 16;
  constexpr bool W16 = TWP == 0;
  constexpr int TWC = W16 ? 1 : TWP, TRG = 4 / (TWC * WN);
  constexpr int LP = ((W16 ? 16 : TWC * 32) + KS - 1 + 7) & ~7;   // multiple of 8: the swizzle bit of a row differs from row 0 by (row * LP/8) & 1
  constexpr int LR = TRG * RWO * (W16 ? 2 : 1) + (NPAIR - 1) + (NPAIR == 2 ? 2 * (JH - 1) : KS - 1);
  // bf16: a voxel is 32 B (two 16-B halves); a ds_read_b128 lane group only ever asks for ONE half of 16 voxels, i.e. 8 of
  // the 16 slots of the 256-B bank row twice (2-way conflict).  Storing the halves of every other run of 8 voxels swapped
  // (half ^= bit 3 of the voxel index; LP is a multiple of 8, so rows differ by a known parity) spreads a
  // group over all 16 slots.
  constexpr bool SWZ = sizeof(T) == 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int wc = wv % TWC, rg = (wv / TWC) % TRG, wn = wv / (TWC * TRG);
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), while the decode below
  // puts consecutive ids on neighbouring tiles / depth slices.  Give every XCD one CONTIGUOUS id range (= a range of depth
  // slices): the kd-neighbour slabs a block re-reads then come from its own L2 (the 3x3x3 layers fetched 2.7x and the
  // 7x7x7 layers 4-8x their input through the fabric with the round-robin order).
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; b /= g.tiles_h; const int d = b % g.D; const int n = b / g.D;
  constexpr int RPA = W16 ? 2 : 1;                 // image rows per accumulator row
  const int h0 = th * (TRG * RWO * RPA), w0 = W16 ? 0 : tw * (TWC * 32);
  const int nt0 = (blockIdx.y * WN + wn) * NT;     // first N tile of this wave

  v16f acc[RW][NT];
#pragma unroll
  for (int i = 0; i < RW; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int pieces = LR * LP * 2;              // 8-channel pieces of the slab
  const int wts = g.NCH * g.NTT * 512;             // elements between consecutive (kd,jh,kw) taps (a pass spans < 2^31)
  // lane-constant part of the A address: position (wc*32 + r), channel half hh, first row of this wave's row group
  const int v_lane = W16 ? (rg * RWO * 2 + (r >> 4)) * LP + (r & 15) : (rg * RWO) * LP + wc * 32 + r;
  // fast staging path: bf16, whole 16-channel chunks, 16-byte aligned voxel rows (block-uniform)
  constexpr int SU = STAGE_UNROLL;
  // whole 16-channel chunks must EXIST in memory (row pitch), not be logical channels: channels >= Cin of a padded row meet
  // zero packed weights (25 = 16 + 9 -> 16 + 16 channel rows of the first skip block take the fast path)
  const int cin_in = g.x3 ? 32 * g.x3 : g.Cin, nstage = g.x3 ? 2 * g.x3 : g.NCH;     // channels / chunks of the INPUT tensor
  const bool fast = SWZ && (nstage * 16 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx)) && (g.ldx % 8 == 0) && (((uintptr_t)x & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 16 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0))) &&
                    (int64_t)g.H * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30);
  const int lp_par = SWZ ? ((LP >> 3) & 1) : 0;
  const int st_half = tid & 1, st_lp0 = (tid >> 1) % LP, st_lr0 = (tid >> 1) / LP;
#if DP_TILED_TAB
  // Round 6: WHICH voxel each of a thread's slab pieces comes from never changes during the block's life (one tile, all kd, all chunks), so
  // it is worked out once -- divisions, bounds and all -- into a table in LDS behind the slab: tab[j][tid] = voxel index inside the
  // (n, depth) plane of piece j * 256 + tid, or H * W (the first voxel BEHIND the plane: out of range of the plane's buffer descriptor,
  // which then returns zeros) for padding and for pieces that do not exist.  Staging a slab is then, per piece, one ds_read_b32, one
  // v_mad_u32_u24 (voxel x row pitch: the two concat operands may differ in pitch), one buffer_load_dwordx4 and one ds_write_b128 at an
  // immediate offset; the round-2 form spent ~20 VALU per piece on carries, bounds, a 64-bit address and the zero select.
  constexpr int NJ = (LR * LP * 2 + 255) / 256;
  unsigned* const tab = (unsigned*)(smem_raw + (size_t)LR * LP * CK * sizeof(T));
  const bool fast_tab = fast && (int64_t)g.H * g.W < (1 << 24);
  if (fast_tab) {
    int lp = st_lp0, lr = st_lr0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int ih = h0 - PAD + lr, iw = w0 - PAD + lp;
      const bool ok = (j * 256 + tid < LR * LP * 2) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      tab[j * 256 + tid] = ok ? (unsigned)(ih * g.W + iw) : (unsigned)(g.H * g.W);
      lp += 128;
#pragma unroll
      for (int c_ = 0; c_ < (128 + LP - 1) / LP; c_++) if (lp >= LP) { lp -= LP; lr++; }
    }
  }
  // (the table is read by the thread that wrote it: DS operations of one wave execute in order, no barrier)
  unsigned char* const st_dst = smem_raw + (tid >> 1) * (CK * (int)sizeof(T)) + ((st_half ^ ((tid >> 4) & 1)) * 16);      // + j * 4096: piece j * 256 + tid
#endif

  constexpr int RS = W16 ? 2 : 1, TS = NPAIR == 2 ? 2 : 1;        // slab rows per accumulator row / per kh tap (pair)
  constexpr int ROWS = RS * (RW - 1) + TS * (JH - 1) + 1;          // slab rows one wave sweeps per kw column
  // the JH weight fragments (all kh taps / pairs) of one kw column and one N tile.  `wb` is wave-uniform (SGPR base); the
  // lane part is ONE 32-bit offset shared by every load, so the 2*JH pointers do not live in VGPRs.
  const int lane_off = r * 16 + hh * 8;
  auto load_bk = [&](const T* wb, int kw, int j, Frag8<T>* bb) {
#pragma unroll
    for (int jh = 0; jh < JH; jh++) bb[jh] = frag_ld_lds(wb + ((jh * KS + kw) * wts + j * 512) + lane_off);   // 32-bit uniform + lane offsets: saddr loads
  };
  // split-kd launches: blockIdx.z = kd * chsplit + (which share of the staged chunks); partial sums meet through fp32 atomics
  const int zsub = g.splitkd ? (int)blockIdx.z % g.chsplit : 0;
  const int kd_lo = g.splitkd ? (int)blockIdx.z / g.chsplit : 0, kd_hi = g.splitkd ? kd_lo + 1 : KS;
  const int ch_lo = g.splitkd ? (nstage * zsub) / g.chsplit : 0, ch_hi = g.splitkd ? (nstage * (zsub + 1)) / g.chsplit : nstage;
  for (int kd = kd_lo; kd < kd_hi; kd++) {
    const int id = d + kd - PAD;
    if (id < 0 || id >= g.D) continue;             // block-uniform: the whole depth slice is zero padding
    for (int ch = ch_lo; ch < ch_hi; ch++) {
      // packed weights of this (kd, chunk) pass; the first kw column is requested BEFORE the slab is staged so that its
      // L2 latency hides behind the staging loads.
      const T* wbase = wq + ((int64_t)kd * NTAP * g.NCH + ch) * g.NTT * 512 + (int64_t)nt0 * 512;      // (not const: a DP_X3 x_hi slab is swept twice)
      Frag8<T> b0[JH], b1[JH];
      load_bk(wbase, 0, 0, b0);
      lds_barrier();                    // LDS-only: __syncthreads() would drain the weight loads just issued (vmcnt(0))
      if (g.dbg == 1 || g.dbg == 3 || g.dbg == 4) {
      }
#if DP_TILED_TAB
      else if (fast_tab) {
        const bool second = g.x2 && ch * CK >= g.csplit;                    // block-uniform: which concat operand this chunk reads
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + id) * g.H) * (int64_t)g.W * ldsrc + c0;
        const uint64_t pa = (uint64_t)(uintptr_t)xplane;
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa), phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const unsigned pbytes = __builtin_amdgcn_readfirstlane((unsigned)(g.H * g.W * ldsrc * 2 - c0 * 2));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)phi << 32) | plo), 0, (int)pbytes, 0x00020000);
        const unsigned ldb = (unsigned)ldsrc * 2u, hoff = (unsigned)st_half * 16u;
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += SU) {
          v4u buf[SU];
#pragma unroll
          for (int j = 0; j < SU; j++)
            if (j0 + j < NJ) buf[j] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, __umul24(tab[(j0 + j) * 256 + tid], ldb) + hoff, 0, 0));
#pragma unroll
          for (int j = 0; j < SU; j++)
            if (j0 + j < NJ) {
              if ((j0 + j + 1) * 256 <= LR * LP * 2 || (j0 + j) * 256 + tid < LR * LP * 2) *(v4u*)(st_dst + (j0 + j) * 4096) = buf[j];
            }
        }
      }
#endif
      else if (fast && !DP_TILED_TAB) {          // (with the table: planes of >= 2^24 voxels take the guarded loop below)
        // straight-line staging: SU independent 16-byte loads are in flight before the first LDS store (a per-piece
        // load->store loop serialises on HBM/L2 latency and dominated the kernel); voxel coordinates advance incrementally
        // (256 threads = 128 voxels per step), no divisions in the loop.
        const bool second = g.x2 && ch * CK >= g.csplit;                    // block-uniform: which concat operand this chunk reads
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + id) * g.H) * (int64_t)g.W * ldsrc + c0 + st_half * 8;
        int lp = st_lp0, lr = st_lr0, v = tid >> 1;
        for (int p0 = 0; p0 < pieces; p0 += 256 * SU) {
          v4u buf[SU]; int vv[SU];
#pragma unroll
          for (int j = 0; j < SU; j++) {
            const int ih = h0 - PAD + lr, iw = w0 - PAD + lp;
            const bool ok = (p0 + j * 256 + tid < pieces) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            v4u t = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));     // 32-bit offset inside one (n, depth) plane
            buf[j] = ok ? t : (v4u){0, 0, 0, 0};
            vv[j] = (p0 + j * 256 + tid < pieces) ? v : -1;
            v += 128; lp += 128;
#pragma unroll
            for (int wq_ = 0; wq_ < (128 + LP - 1) / LP; wq_++) if (lp >= LP) { lp -= LP; lr++; }   // branch-free carry (LP is a constant)
          }
#pragma unroll
          for (int j = 0; j < SU; j++)
            if (vv[j] >= 0) *(v4u*)(slab + (int64_t)vv[j] * CK + (st_half ^ ((vv[j] >> 3) & 1)) * 8) = buf[j];
        }
      } else {
        for (int p = tid; p < pieces; p += 256) {
          int half = p & 1, v = p >> 1, lp = v % LP, lr = v / LP;
          int ih = h0 - PAD + lr, iw = w0 - PAD + lp, c = ch * CK + half * 8;
          int nv = cin_in - c; nv = nv > 8 ? 8 : nv;
          bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
          const bool second = g.x2 && c >= g.csplit;
          if (!second && g.x2 && c + nv > g.csplit) nv = g.csplit - c;        // a piece never straddles the two operands (csplit % 8 == 0 is required)
          const T* xsrc = second ? (const T*)g.x2 : x;
          const int ldsrc = second ? g.ldx2 : g.ldx, cc = c - (second ? g.csplit : 0);
          Frag8<T> f = ok ? frag_load(xsrc + ((((int64_t)n * g.D + id) * g.H + ih) * g.W + iw) * ldsrc + cc, nv) : frag_zero<T>();
          frag_st_lds(slab + (int64_t)v * CK + (half ^ (SWZ ? ((v >> 3) & 1) : 0)) * 8, f);
        }
      }
      lds_barrier();
      if (g.dbg >= 2) continue;
      const int nrep = (g.x3 && ch < g.x3) ? 2 : 1;     // DP_X3: the x_hi slab also meets its w_lo block (weight chunk 2 x3 + ch)
#pragma unroll 1
      for (int rep = 0; rep < nrep; rep++) {
      if (rep) { wbase += (int64_t)2 * g.x3 * g.NTT * 512; load_bk(wbase, 0, 0, b0); }
      // Sweep: for one kw column, EVERY slab row is read from LDS once and feeds all the kh taps (pairs) that use it
      // (output row i and tap jh share the input row RS*i + TS*jh).  A tap-outer loop read the A fragment again for every
      // (row, tap) pair -- 1 KiB of LDS per 32-cycle MFMA per SIMD, i.e. exactly the CU's whole LDS bandwidth -- and held
      // the sweep near half the MFMA rate; row-outer needs ROWS reads for RW*JH MFMAs (15 for 36 with 7x7x7 pairs, 14 for 56
      // unpaired).  The JH weight fragments of the NEXT column are in flight (two named register sets) while this one runs.
      // The A rows are software-pipelined through THREE register sets: the reads of rows rho+1 and rho+2 are in flight while the
      // MFMAs of row rho issue (a read -> s_waitcnt lgkmcnt(0) -> MFMA chain per row left the matrix pipe idle for most of every
      // LDS round trip: ~2.4 MFMAs = 77 cycles of work per 100+ cycle read; the sweep alone ran at 43 % of the MFMA peak).  The last
      // two steps of a column already request rows 0 and 1 of the next column (`pre_next`), so only the first column of a pass
      // exposes the latency.  On entry fa[0], fa[1] hold (requests for) rows 0, 1 of column kw.
      Frag8<T> fa[3];
      auto a_addr = [&](int kw, int rho) -> const T* {
        const int v_tap = v_lane + kw;
        const int sw0 = SWZ ? (hh ^ ((v_tap >> 3) & 1)) : hh;
        return slab + v_tap * CK + (((rho & 1) ? (sw0 ^ lp_par) : sw0) * 8) + rho * LP * CK;   // odd rows flip the half iff LP/8 is odd
      };
      auto do_kw = [&](int kw, auto jc, const Frag8<T>* bb, bool pre_next) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int rho = 0; rho < ROWS; rho++) {
          const int nxt = rho + 2;
          if (nxt < ROWS) fa[nxt % 3] = frag_ld_lds(a_addr(kw, nxt));
          else if (pre_next) fa[nxt % 3] = frag_ld_lds(a_addr(kw + 1, nxt - ROWS));
          __builtin_amdgcn_sched_barrier(0);          // the prefetch stays above this row's MFMAs
#pragma unroll
          for (int jh = 0; jh < JH; jh++) {
            const int num = rho - TS * jh;
            if (num < 0 || num % RS != 0 || num / RS >= RW) continue;      // compile-time after unrolling
            acc[num / RS][j] = mma32(fa[rho % 3], bb[jh], acc[num / RS][j]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ROWS % 3 != 0) {                            // next column's rows 0, 1 sit in slots ROWS % 3, (ROWS + 1) % 3: bring them to 0, 1
          const Frag8<T> t0 = fa[ROWS % 3], t1 = fa[(ROWS + 1) % 3];
          fa[0] = t0; fa[1] = t1;
        }
      };
      auto do_kw_same = [&](int kw, auto jc, const Frag8<T>* bb) {      // as do_kw, but the column that follows is kw again (two N tiles)
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int rho = 0; rho < ROWS; rho++) {
          const int nxt = rho + 2;
          fa[nxt % 3] = frag_ld_lds(a_addr(kw, nxt < ROWS ? nxt : nxt - ROWS));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jh = 0; jh < JH; jh++) {
            const int num = rho - TS * jh;
            if (num < 0 || num % RS != 0 || num / RS >= RW) continue;
            acc[num / RS][j] = mma32(fa[rho % 3], bb[jh], acc[num / RS][j]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ROWS % 3 != 0) {
          const Frag8<T> t0 = fa[ROWS % 3], t1 = fa[(ROWS + 1) % 3];
          fa[0] = t0; fa[1] = t1;
        }
      };
      fa[0] = frag_ld_lds(a_addr(0, 0)); fa[1] = frag_ld_lds(a_addr(0, 1));
      std::integral_constant<int, 0> J0;
      // sched_barrier: without it the scheduler sinks the prefetch loads down to their first use (to shorten live ranges)
      // and every column pays the full L2 latency.
      if (NT == 1) {
#pragma unroll 1
        for (int kw = 0; kw < KS; kw += 2) {        // KS is odd: the pass ends on b0, which is free again at the next pass's top
          if (kw + 1 < KS) { load_bk(wbase, kw + 1, 0, b1); __builtin_amdgcn_sched_barrier(0); }
          do_kw(kw, J0, b0, kw + 1 < KS);
          if (kw + 1 < KS) {
            if (kw + 2 < KS) { load_bk(wbase, kw + 2, 0, b0); __builtin_amdgcn_sched_barrier(0); }
            do_kw(kw + 1, J0, b1, kw + 2 < KS);
          }
        }
      } else {                                      // two N tiles: they alternate between the register sets (A rows are read once per tile)
        std::integral_constant<int, NT - 1> J1;
#pragma unroll 1
        for (int kw = 0; kw < KS; kw++) {              // (the second N tile sweeps the same column again: its "next column" is kw itself)
          load_bk(wbase, kw, 1, b1); __builtin_amdgcn_sched_barrier(0);
          do_kw_same(kw, J0, b0);
          if (kw + 1 < KS) { load_bk(wbase, kw + 1, 0, b0); __builtin_amdgcn_sched_barrier(0); }
          do_kw(kw, J1, b1, kw + 1 < KS);
        }
      }
      }
    }
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  // A lane holds ONE channel of 16 positions, so direct stores are 2-byte scatters (measured: 0.8 TB/s, 0.16 ms of a 0.49 ms
  // 3x3x3 layer).  Each wave therefore transposes its 32-position x NC-channel tile through a private LDS patch (the slab
  // is dead by now) and writes 16 contiguous bytes per lane: whole voxel rows, 1 KiB per wave-instruction.
  if (g.dbg == 4) return;                                         // experiment: loop skeleton only
  const int hw0 = h0 + rg * RWO * RPA, wbase_o = w0 + wc * 32;
  constexpr int NC = NPAIR == 2 ? 16 : 32;                       // channels per output tile
  constexpr int EPC = 16 / (int)sizeof(TO);                       // elements per 16-byte chunk
  constexpr int CPP = NC / EPC;                                  // chunks per position
  constexpr int PASSES = 32 * CPP / 64;
  __syncthreads();                                               // every wave is done reading the slab
  TO* patch = (TO*)smem_raw + (wv & 3) * (2 * 32 * 32);                      // two alternating [32 positions][NC] patches per wave
  TO* y2 = (TO*)g.y2;
  const bool wide = g.wide != 0;      // (host-computed: no split-kd, 16-byte aligned output rows)
  const int dd = d, nn = n;
  const int stat_blk = (dd * g.tiles_h + th) * g.tiles_w + tw;
  // destination of output channel c of voxel `vox` (virtual concat: channels >= osplit live in y2)
  auto out_ptr = [&](int64_t vox, int c) -> TO* { return (y2 && c >= g.osplit) ? y2 + vox * g.ldy2 + (c - g.osplit) : y + vox * g.ldy + c; };
  auto store_tile = [&](const TO* pt, int oh_lo, int nt_idx) {   // tile in `pt` as [32 positions][NC channels]; oh_lo: image row of position 0
    const int cbase = NPAIR == 2 ? 0 : nt_idx * 32;
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
      const int q = ps * 64 + lane, m = q / CPP, cc = (q % CPP) * EPC;
      const int oh = W16 ? oh_lo + (m >> 4) : oh_lo, ow = W16 ? (m & 15) : wbase_o + m;
      if (oh < g.H && ow < g.W && cbase + cc < g.Cout) {
        TO* dst = out_ptr((((int64_t)n * g.D + d) * g.H + oh) * g.W + ow, cbase + cc);
        if (cbase + cc + EPC <= g.Cout) *(v4u*)dst = *(const v4u*)(pt + m * NC + cc);
        else for (int k = 0; k < EPC; k++) if (cbase + cc + k < g.Cout) dst[k] = pt[m * NC + cc + k];
      }
    }
  };
  // The wide path is straight-line code: one runtime `wide` test per element used to put a branch and a full
  // ds_bpermute -> s_waitcnt round trip between consecutive LDS writes (16 serialised LDS latencies per row: the epilogue
  // wrote at 1.7 TB/s and cost 40 % of a 3x3x3 layer).  Two patches per wave alternate so that a row's LDS writes never
  // wait for the previous row's read.
  if (NPAIR == 2) {
    const int co = lane & 15;
    const bool low = (lane & 16) == 0;
    const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
    if (wide) {
      float st1 = 0.f, st2 = 0.f;
#pragma unroll
      for (int t = 0; t < RWO; t++) {
        TO* pp = patch + (t & 1) * (32 * NC);
        const bool rowok = hw0 + t < g.H;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          // v_permlane16_swap: lanes 16-31 / 48-63 of the first operand <-> lanes 0-15 / 32-47 of the second.  The low lanes
          // receive the odd tap (row t+1, element k) they must add to their own even-tap sum; the high lanes receive the even
          // tap (row t, element k+8) that completes THEIR odd-tap sum: one VALU swap finishes two output elements.
          v2u sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[t + 1][0][k]), __float_as_uint(acc[t][0][k + 8]), false, false);
          const float v = (low ? acc[t][0][k] + __uint_as_float(sw[1]) : acc[t + 1][0][k + 8] + __uint_as_float(sw[0])) + bv;
          const int m = (k & 3) + 8 * (k >> 2) + 4 * hh + (low ? 0 : 16);
          st_f(pp + m * NC + co, v);
          const float vs = (rowok && wbase_o + m < g.W) ? as_stored<TO>(v) : 0.f;      // branch-free: voxels outside the volume count as 0
          st1 += vs; st2 += vs * vs;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
        store_tile(pp, hw0 + t, 0);
        __builtin_amdgcn_wave_barrier();
      }
      if (g.stat_part) {          // block-uniform
        // lanes co, co+16, co+32, co+48 hold the same channel; then the four waves meet in LDS (the patches are dead)
        st1 += __shfl_xor(st1, 16, 64); st2 += __shfl_xor(st2, 16, 64);
        st1 += __shfl_xor(st1, 32, 64); st2 += __shfl_xor(st2, 32, 64);
        __syncthreads();
        float* sred = (float*)smem_raw;
        if (lane < 16) { sred[(wv * 2) * 16 + lane] = st1; sred[(wv * 2 + 1) * 16 + lane] = st2; }
        __syncthreads();
        if (tid < 32) {
          const int which = tid >> 4, c = tid & 15;
          if (c < g.Cout)
            g.stat_part[(((int64_t)nn * g.stat_nblk + stat_blk) * 2 + which) * g.Cout + c] =
                (sred[(0 + which) * 16 + c] + sred[(2 + which) * 16 + c]) + (sred[(4 + which) * 16 + c] + sred[(6 + which) * 16 + c]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < RWO; t++) {
        const int oh = hw0 + t;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          float z1 = __shfl_down(acc[t + 1][0][e], 16, 64);            // odd tap of the pair: accumulated one row below
          float v = acc[t][0][e] + z1 + bv;
          const int m = (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (low && co < g.Cout && oh < g.H && wbase_o + m < g.W)
            st_f(out_ptr((((int64_t)n * g.D + d) * g.H + oh) * g.W + wbase_o + m, co), v);
        }
      }
    }
  } else if (wide) {
    float st1[NT], st2[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int co = (nt0 + j) * 32 + r;
      const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
      st1[j] = 0.f; st2[j] = 0.f;
#pragma unroll
      for (int i = 0; i < RW; i++) {
        TO* pp = patch + (i & 1) * (32 * NC);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = (e & 3) + 8 * (e >> 2) + 4 * hh;
          const float v = acc[i][j][e] + bv;
          st_f(pp + m * NC + r, v);
          const int oh = W16 ? hw0 + 2 * i + (m >> 4) : hw0 + i, ow = W16 ? (m & 15) : wbase_o + m;
          const float vs = (oh < g.H && ow < g.W) ? as_stored<TO>(v) : 0.f;
          st1[j] += vs; st2[j] += vs * vs;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
        store_tile(pp, W16 ? hw0 + 2 * i : hw0 + i, nt0 + j);
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (g.stat_part) {            // block-uniform
      __syncthreads();
      float* sred = (float*)smem_raw;            // [wave][j][which][32]
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const float a1 = st1[j] + __shfl_xor(st1[j], 32, 64), a2 = st2[j] + __shfl_xor(st2[j], 32, 64);   // the two position halves of a channel
        if (lane < 32) { sred[((wv * NT + j) * 2) * 32 + lane] = a1; sred[((wv * NT + j) * 2 + 1) * 32 + lane] = a2; }
      }
      __syncthreads();
      for (int o = tid; o < WN * NT * 2 * 32; o += 256) {
        const int c32 = o & 31, which = (o >> 5) & 1, jj = o >> 6, wn2 = jj / NT, j = jj % NT, c = ((blockIdx.y * WN + wn2) * NT + j) * 32 + c32;
        if (c < g.Cout) {
          float sum = 0.f;
#pragma unroll
          for (int w4 = 0; w4 < 4 / WN; w4++) sum += sred[(((wn2 * (4 / WN) + w4) * NT + j) * 2 + which) * 32 + c32];
          g.stat_part[(((int64_t)nn * g.stat_nblk + stat_blk) * 2 + which) * g.Cout + c] = sum;
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int co = (nt0 + j) * 32 + r;
      const float bv = (bias && co < g.Cout && !g.splitkd) ? bias[co] : 0.f;
#pragma unroll
      for (int i = 0; i < RW; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = (e & 3) + 8 * (e >> 2) + 4 * hh;
          const int oh = W16 ? hw0 + 2 * i + (m >> 4) : hw0 + i;
          const int ow = W16 ? (m & 15) : wbase_o + m;
          if (co < g.Cout && oh < g.H && ow < g.W) {
            const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
            if (g.splitkd) atomicAdd(ws + (int64_t)blockIdx.z * g.det_slab + vox * g.Cout + co, acc[i][j][e]);
            else st_f(out_ptr(vox, co), acc[i][j][e] + bv);
          }
        }
      }
    }
  }
}

// split-kd epilogue: y = T(ws + bias)
template <typename T>
__global__ void k_conv_split_finish(float* __restrict__ ws, const float* __restrict__ bias, T* __restrict__ y, int64_t rows, int C, int ldy,
                                    T* __restrict__ y2, int ldy2, int osplit, int rezero, int nslab, int64_t slab) {
  int64_t total = rows * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t rrow = i / C; int c = (int)(i - rrow * C);
    T* dst = (y2 && c >= osplit) ? y2 + rrow * ldy2 + (c - osplit) : y + rrow * ldy + c;
    float v = ws[i];
    if (rezero) ws[i] = 0.f;
    for (int sl = 1; sl < nslab; sl++) { v += ws[i + sl * slab]; if (rezero) ws[i + sl * slab] = 0.f; }     // (deterministic mode: the kd / chunk shares in order)
    st_f(dst, v + (bias ? bias[c] : 0.f));
  }
}

template <typename T, int KS, int NPAIR, int RW, int NT, int TWP, typename TO = T, int WN = 1>
static int launch_tiled(const void* x, const void* wq, const float* bias, void* y, float* ws, TiledGeom g, int ygrid, hipStream_t s) {
  size_t smem = (size_t)g.LR * g.LP * 16 * sizeof(T);
#if DP_TILED_TAB
  smem += (size_t)(((size_t)g.LR * g.LP * 2 + 255) / 256) * 256 * sizeof(unsigned);      // the staging table behind the slab
#endif
  if (smem < 8 * 32 * 32 * sizeof(TO)) smem = 8 * 32 * 32 * sizeof(TO);     // the epilogue transposes through 2 patches per wave
  auto kern = k_conv_tiled<T, KS, NPAIR, RW, NT, TWP, TO, WN>;
  if (smem > 160 * 1024) { dp_set_error("conv3d_tiled: slab %zu B exceeds LDS", smem); return 1; }
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { dp_set_error("conv3d_tiled: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return 1; }
  }
  dim3 grid(g.N * g.D * g.tiles_h * g.tiles_w, ygrid, g.splitkd ? KS * g.chsplit : 1);
  if (getenv("DP_DEBUG_OCC")) {
    int nb = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, smem);
    fprintf(stderr, "[dp] conv_tiled KS=%d NPAIR=%d RW=%d NT=%d TWP=%d smem=%zu grid=%u x %u x %u occupancy(blocks/CU)=%d (%s)\n", KS, NPAIR, RW, NT, TWP, smem,
            grid.x, grid.y, grid.z, nb, hipGetErrorString(e));
  }
  const int nslab = (g.splitkd && g.det_slab) ? KS * g.chsplit : 1;
  if (g.splitkd && !g_scratch_zeroed) {
    hipError_t me = hipMemsetAsync(ws, 0, (size_t)g.N * g.D * g.H * g.W * g.Cout * sizeof(float) * nslab, s);
    if (me != hipSuccess) { dp_set_error("conv3d_tiled: memset failed"); return 1; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (TO*)y, ws, g);
  if (g.splitkd) {
    int64_t rows = (int64_t)g.N * g.D * g.H * g.W;
    int gb = (int)((rows * g.Cout + 255) / 256); if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_conv_split_finish<TO>, dim3(gb), dim3(256), 0, s, ws, bias, (TO*)y, rows, g.Cout, g.ldy, (TO*)g.y2, g.ldy2, g.osplit, g_scratch_zeroed, nslab, g.det_slab);
  }
  return 0;
}

static void tiled_geometry(TiledGeom& g, int k, int np, int rw, int& nt, int* ygrid, bool* w16, int* wn_out = nullptr, bool allow_wn = true) {
  int rwo = rw - (np - 1), JH = np == 2 ? (k + 1) / 2 : k;
  *w16 = (g.W <= 16 && np == 1);
  const int ntt = (g.Cout * np + 31) / 32;
  // W16 tiles on planes of <= 16 rows with >= 4 channel tiles: two of the four waves take other channel tiles instead of rows >= H
  static const bool no_wn = getenv("DP_NO_WN") != nullptr;
  int wn = (*w16 && nt == 2 && g.H <= 16 && ntt % 4 == 0 && !no_wn && allow_wn) ? 2 : 1;
  // One-column tiles (16 < W <= 32, the 32^3 level) with two channel tiles per wave: 4 row groups x 4 rows x 32 positions is a
  // 16-row block, i.e. 128 blocks for 2 x 32^3 -- half of the compute units, or (what round 1-2 did) one block per kd with fp32
  // atomics, whose epilogue + finish pass were 36 % of a 64 -> 64 launch (knock-outs, round 3: 0.231 ms of which 0.085 ms; without
  // the split, on HALF of the chip, the same launch took 0.290 ms).  Two row groups x two channel-tile groups instead: 8-row
  // blocks, every wave ONE channel tile of 4 rows (64 accumulator registers), 256 blocks, no split, no scratch, statistics from
  // the epilogue.  Measured at 2 x 32^3 (tools/bench_conv.py): 7^3 64->64 0.223 -> 0.201 ms, 64->128 0.361 -> 0.292; 3^3 64->64
  // 0.074 -> 0.032, 128->64 0.185 -> 0.054, 64->128 0.129 -> 0.043 (their split epilogues cost more than their sweeps); 7^3
  // 128->64 0.380 -> 0.403: with 8 input chunks and one output-tile pair the split amortises, that shape keeps it.
  static const bool no_wn32 = getenv("DP_NO_WN32") != nullptr;
  if (!*w16 && np == 1 && nt == 2 && g.W <= 32 && ntt % 2 == 0 && allow_wn && !no_wn && !no_wn32 &&
      !(k == 7 && ntt == 2 && g.Cin > 64) && (int64_t)g.N * g.D * cdiv(g.H, 16) * (ntt / 2) < 400) { wn = 2; nt = 1; }
  if (wn_out) *wn_out = wn;
  if (*w16) { g.TWC = 1; g.TRG = 4 / wn; }
  else if (g.W > 32) {
    // 32-position wave columns: 4 x 1, 2 x 2 or 1 x 4 (columns x row groups) -- whichever covers the row with the least padding, ties
    // to the wider tile (round 5: W = 96 took one 128-position tile and W = 192 two, a quarter of the MFMAs on padding; now three
    // 32 x 32-row tiles and three 64 x 16-row tiles.  A/B on one box, 7^3: 16 -> 32 at 4 x 96^3 1.328 -> 0.954 ms, 64 -> 32 at 64 x 96 x 96
    // 0.855 -> 0.777, 32 -> 32 0.412 -> 0.387)
    static const bool fixed = getenv("DP_TILED_TWC_OLD") != nullptr;
    const int p4 = cdiv(g.W, 128) * 128, p2 = cdiv(g.W, 64) * 64, p1 = cdiv(g.W, 32) * 32;
    int twc = g.W > 64 ? 4 : 2;
    // (7^3 only: the 3^3 launches are staging-bound and lose 5-13 % to the wider halo of narrower tiles)
    if (!fixed && k == 7) { twc = 4; int best = p4; if (p2 < best) { twc = 2; best = p2; } if (p1 < best) twc = 1; if (g.W <= 64 && twc == 4) twc = 2; }
    g.TWC = twc; g.TRG = 4 / twc;
  } else { g.TWC = 1; g.TRG = 4 / wn; }
  int rpa = *w16 ? 2 : 1;
  g.LR = g.TRG * rwo * rpa + (np - 1) + (np == 2 ? 2 * (JH - 1) : k - 1);
  g.LP = ((*w16 ? 16 : g.TWC * 32) + k - 1 + 7) & ~7;   // multiple of 8: the LDS swizzle bit of a row differs from row 0 by (row * LP/8) & 1
  g.NCH = (g.Cin + 15) / 16; g.NTT = (g.Cout * np + 31) / 32;
  g.tiles_h = cdiv(g.H, g.TRG * rwo * rpa); g.tiles_w = *w16 ? 1 : cdiv(g.W, g.TWC * 32);
  *ygrid = cdiv(g.NTT, nt * wn);
  int64_t blocks = (int64_t)g.N * g.D * g.tiles_h * g.tiles_w * *ygrid;
  const int split_below = (!*w16 && wn == 2) ? 200 : 400;   // (the 8-row arrangement is there to AVOID the split: 256 blocks are enough)
  g.splitkd = (np == 1 && blocks < split_below) ? 1 : 0;    // small volumes: one block per kd, fp32 atomic accumulation (deterministic mode: one scratch slab per share, added in order)
  // ... and, when even k blocks per tile leave the chip half empty, per share of the input chunks (>= 4 chunks per share)
  g.chsplit = 1;
  if (g.splitkd) { while (blocks * k * g.chsplit < 400 && g.NCH / (g.chsplit * 2) >= 4) g.chsplit *= 2; }
}

// fp32 scratch elements dp_conv3d_tiled needs for this shape (0 = none)
extern "C" int dp_conv3d_tiled_ws_elems(int N, int D, int H, int W, int Cin, int Cout, int k) {
  if (!tiled_applicable(Cin, Cout, k, 1, k / 2, 1, W)) return 0;
  if (cc16_applicable(Cin, Cout, k, W)) return 0;
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  TiledGeom g; g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.dbg = 0; g.x2 = nullptr; g.y2 = nullptr;
  int ygrid; bool w16; tiled_geometry(g, k, np, rw, nt, &ygrid, &w16);
  if (!g.splitkd) return 0;
  int64_t e = (int64_t)N * D * H * W * Cout;
  if (dp_det(DET_SPLITKD)) e *= k * g.chsplit;
  return e > 2000000000LL ? -1 : (int)e;
}

static bool tiled_wide(const TiledGeom& g, const void* y, int dtype) {
  const int es = (dtype == DP_F32 || dtype == DP_X3 || dtype == DP_X1) ? 4 : 2, epc = 16 / es;      // element size of the OUTPUT
  return !g.splitkd && (g.ldy * es) % 16 == 0 && (((uintptr_t)y & 15) == 0) &&
         (!g.y2 || ((g.ldy2 * es) % 16 == 0 && (((uintptr_t)g.y2 & 15) == 0) && g.osplit % epc == 0));
}
// Number of per-sample partial statistics rows dp_conv3d_tiled_stats writes for this shape (= D x output tiles), or 0 when the
// launch cannot produce statistics (split-kd volumes, output rows that are not 16-byte aligned: the caller then runs dp_stats_partial).
extern "C" int dp_conv3d_tiled_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int k, int ldy, int dtype) {
  if (!tiled_applicable(Cin, Cout, k, 1, k / 2, 1, W)) return 0;
  if (cc16_applicable(Cin, Cout, k, W)) return cc16_wide(nullptr, ldy, nullptr, 0, 0, dtype) ? cc16_stat_blocks(D, H, W, k, dtype) : 0;
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  TiledGeom g; g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.dbg = 0; g.x2 = nullptr; g.y2 = nullptr; g.ldy = ldy;
  int ygrid; bool w16; tiled_geometry(g, k, np, rw, nt, &ygrid, &w16);
  if (!tiled_wide(g, nullptr, dtype)) return 0;
  return D * g.tiles_h * g.tiles_w;
}
// Tiled convolution with weights packed by dp_pack_conv_weight_tiled.  Same-size output ("same" padding).
// ws: fp32 scratch of dp_conv3d_tiled_ws_elems() elements (may be NULL when that is 0).
extern "C" int dp_conv3d_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                                void* y, int ldy, void* y2, int ldy2, int osplit, float* ws, int N, int D, int H, int W,
                                int Cin, int Cout, int k, int dtype, void* stream);
static int conv3d_tiled_impl(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                             void* y, int ldy, void* y2, int ldy2, int osplit, float* ws, float* stat_part, int N, int D, int H, int W,
                             int Cin, int Cout, int k, int dtype, void* stream);
extern "C" int dp_conv3d_tiled_stats(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                                     void* y, int ldy, float* ws, float* stat_part, int N, int D, int H, int W, int Cin, int Cout, int k,
                                     int dtype, void* stream) {
  if (!stat_part) DP_FAIL("conv3d_tiled_stats: stat_part is NULL");
  return conv3d_tiled_impl(x, ldx, x2, ldx2, csplit, wq, bias, y, ldy, nullptr, 0, 0, ws, stat_part, N, D, H, W, Cin, Cout, k, dtype, stream);
}
extern "C" int dp_conv3d_tiled(const void* x, int ldx, const void* wq, const float* bias, void* y, int ldy, float* ws, int N, int D, int H, int W,
                               int Cin, int Cout, int k, int dtype, void* stream) {
  return dp_conv3d_tiled2(x, ldx, nullptr, 0, 0, wq, bias, y, ldy, nullptr, 0, 0, ws, N, D, H, W, Cin, Cout, k, dtype, stream);
}
extern "C" int dp_conv3d_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                                void* y, int ldy, void* y2, int ldy2, int osplit, float* ws, int N, int D, int H, int W,
                                int Cin, int Cout, int k, int dtype, void* stream) {
  return conv3d_tiled_impl(x, ldx, x2, ldx2, csplit, wq, bias, y, ldy, y2, ldy2, osplit, ws, nullptr, N, D, H, W, Cin, Cout, k, dtype, stream);
}
static int conv3d_tiled_impl(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                             void* y, int ldy, void* y2, int ldy2, int osplit, float* ws, float* stat_part, int N, int D, int H, int W,
                             int Cin, int Cout, int k, int dtype, void* stream) {
  if (!tiled_applicable(Cin, Cout, k, 1, k / 2, 1, W)) DP_FAIL("conv3d_tiled: shape not supported");
  if (x2 && (csplit <= 0 || csplit >= Cin || csplit % 8)) DP_FAIL("conv3d_tiled: input split must be a multiple of 8 inside (0, Cin)");
  if (y2 && (osplit <= 0 || osplit >= Cout || osplit % 8)) DP_FAIL("conv3d_tiled: output split must be a multiple of 8 inside (0, Cout)");
  if (cc16_applicable(Cin, Cout, k, W))
    return cc16_launch(x, ldx, x2, ldx2, csplit, wq, bias, y, ldy, y2, ldy2, osplit, stat_part, N, D, H, W, Cin, Cout, k, dtype, STREAM);
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  TiledGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ldx = ldx; g.ldy = ldy;
  g.x2 = x2; g.ldx2 = ldx2; g.csplit = csplit; g.y2 = y2; g.ldy2 = ldy2; g.osplit = osplit;
  int ygrid; bool w16; int wn; tiled_geometry(g, k, np, rw, nt, &ygrid, &w16, &wn);
  g.det_slab = (g.splitkd && dp_det(DET_SPLITKD)) ? (int64_t)N * D * H * W * Cout : 0;
  g.x3 = 0;
  if (dtype == DP_X3) {
    if (x2 || Cin % 48 || ldx < 2 * (Cin / 3)) DP_FAIL("conv3d_tiled: a DP_X3 launch takes ONE [x_hi | x_lo] tensor of 2/3 Cin channels (Cin = 3 x a multiple of 16)");
    g.x3 = Cin / 48;
  }
  g.wide = tiled_wide(g, y, dtype) ? 1 : 0;
  g.stat_part = stat_part; g.stat_nblk = D * g.tiles_h * g.tiles_w;
  if (stat_part && !g.wide) DP_FAIL("conv3d_tiled_stats: this launch cannot produce statistics (dp_conv3d_tiled_stat_blocks == 0)");
  { const char* e = getenv("DP_DBG"); g.dbg = e ? atoi(e) : 0; }
  if (getenv("DP_DEBUG_SLOW")) {      // report launches that will take the guarded (slow) staging path
    int nch = (Cin + 15) / 16;
    bool fast = dtype != DP_F32 && nch * 16 <= (x2 ? csplit + ldx2 : ldx) && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
                (!x2 || (csplit % 16 == 0 && ldx2 % 8 == 0 && ((uintptr_t)x2 & 15) == 0));
    bool wide = (ldy * 2) % 16 == 0 && ((uintptr_t)y & 15) == 0 && (!y2 || ((ldy2 * 2) % 16 == 0 && ((uintptr_t)y2 & 15) == 0 && osplit % 8 == 0));
    if (!fast || !wide) fprintf(stderr, "[dp slow] conv3d_tiled k=%d Cin=%d Cout=%d %dx%dx%d ldx=%d ldx2=%d csplit=%d ldy=%d: staging %s, epilogue %s\n", k, Cin, Cout,
                                D, H, W, ldx, ldx2, csplit, ldy, fast ? "fast" : "GUARDED", wide ? "wide" : "SCALAR");
  }
  if (g.splitkd && !ws) DP_FAIL("conv3d_tiled: this shape needs the fp32 scratch (dp_conv3d_tiled_ws_elems)");
  if ((int64_t)N * D * g.tiles_h * g.tiles_w > 2000000000LL) DP_FAIL("conv3d_tiled: grid too large");
  int rc = 0;
  hipStream_t s = STREAM;
#define GO(TT, KS_, NP, RW_, NT_, TWP_) rc = x3 ? launch_tiled<bf16_t, KS_, NP, RW_, NT_, TWP_, float>(x, wq, bias, y, ws, g, ygrid, s) \
                                             : launch_tiled<TT, KS_, NP, RW_, NT_, TWP_>(x, wq, bias, y, ws, g, ygrid, s)
#define GOW2(TT, KS_) rc = x3 ? launch_tiled<bf16_t, KS_, 1, 4, 2, 0, float, 2>(x, wq, bias, y, ws, g, ygrid, s) \
                              : launch_tiled<TT, KS_, 1, 4, 2, 0, TT, 2>(x, wq, bias, y, ws, g, ygrid, s)
#define BYTW(TT, KS_, NP, RW_, NT_) do { if (g.TWC == 4) GO(TT, KS_, NP, RW_, NT_, 4); else if (g.TWC == 2) GO(TT, KS_, NP, RW_, NT_, 2); \
                                         else GO(TT, KS_, NP, RW_, NT_, 1); } while (0)
#define GOW32(TT, KS_) rc = x3 ? launch_tiled<bf16_t, KS_, 1, 4, 1, 1, float, 2>(x, wq, bias, y, ws, g, ygrid, s) \
                               : launch_tiled<TT, KS_, 1, 4, 1, 1, TT, 2>(x, wq, bias, y, ws, g, ygrid, s)
#define BYCFG(TT, KS_) do { if (np == 2) BYTW(TT, KS_, 2, 9, 1); else if (!w16 && wn == 2) GOW32(TT, KS_); \
                            else if (nt == 1) { if (w16) GO(TT, KS_, 1, 8, 1, 0); else BYTW(TT, KS_, 1, 8, 1); } \
                            else { if (w16 && wn == 2) GOW2(TT, KS_); else if (w16) GO(TT, KS_, 1, 4, 2, 0); else BYTW(TT, KS_, 1, 4, 2); } } while (0)
  const bool x3 = dtype == DP_X3 || dtype == DP_X1;          // (float output; g.x3 says whether the operands are split)
  if (dtype == DP_BF16 || x3) { if (k == 7) BYCFG(bf16_t, 7); else BYCFG(bf16_t, 3); }
  else if (dtype == DP_F16) { if (k == 7) BYCFG(f16_t, 7); else BYCFG(f16_t, 3); }
  else if (dtype == DP_F32) { if (k == 7) BYCFG(float, 7); else BYCFG(float, 3); }
  else DP_FAIL("conv3d_tiled: bad dtype");
#undef BYCFG
#undef BYTW
#undef GOW2
#undef GOW32
#undef GO
  if (rc) return rc;
  DP_CHECK_LAUNCH("conv3d_tiled"); return 0;
}

// hooks used by the generic entry points (conv_generic.hip): the tiled kernels need their own weight packing, so the
// generic dp_conv3d never dispatches to them implicitly.
int dp_conv3d_tiled_try(const void*, int, const void*, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, int, int, int,
                        int, int, void*) { return -1; }
int dp_wgrad_tiled_try(const void*, int, const void*, int, float*, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int,
                       int64_t, int64_t, int64_t, int, void*) { return -1; }

// ================================================================================================ tiled weight gradient
// dW[co][ci][kd][kh][kw] = sum_{n,d,h,w} gy[n,d,h,w,co] * x[n, d+kd-P, h+kh-P, w+kw-P, ci]      (stride 1, "same" padding)
// MFMA 32x32x16 with K = 16 consecutive voxels along W:
//   A = x^T : M = 32 input channels (Cin <= 16: two horizontal taps kw = 2j, 2j+1 x 16 channels, "M pairing")
//   B = gy  : N = 32 output channels (Cout <= 16: two vertical taps kh = 2j, 2j+1 x 16 channels, "N pairing")
// Both operands are voxel-major in memory, so their k-major fragments come from LDS through ds_read_b64_tr_b16 (bf16) or
// scalar reads (f32).  A block fixes kd and one (M tile, N tile, group of up to 4 kh); its 4 waves own one kh (pair) each
// -- or split the 16-voxel chunks when there are fewer than 3 kh (pairs) -- and keep the KWT per-kw accumulators in
// registers while the block sweeps its share of (n, d, h-tile, w-tile) voxel tiles; one atomic pass at the end into a
// tap-major fp32 scratch [tap][ci][co] (co contiguous => full-rate atomics), unpacked into the caller's layout afterwards.
struct WgtGeom {
  int N, D, H, W, Cin, Cout, ldx, ldgy;
  int tiles_h, tiles_w, MT, NTn, KHG, ydim, zdim;
  const void* x2; int ldx2, csplit;   // virtual concat of the input (see TiledGeom)
  int dbg;             // experiments only (env DP_DBG): 1 = skip staging, 2 = skip the MFMA sweep, 3 = both, 4 = also skip the epilogue
  int64_t slab;        // deterministic mode: voxel share yb adds into ITS OWN scratch slab (yb * slab elements further; every address then has one contributor and k_wgrad_unpack adds the slabs in order); 0 = one shared scratch
};

template <typename T, int KS, int NPAIR, int MPAIR>
struct WgtCfg {
  static constexpr int PAD = KS / 2;
  static constexpr int JHN = NPAIR == 2 ? (KS + 1) / 2 : KS;      // kh (pairs)
  static constexpr int KWT = MPAIR == 2 ? (KS + 1) / 2 : KS;      // kw (pairs) = accumulators per wave
  static constexpr int WKH = JHN >= 3 ? 4 : (JHN == 2 ? 2 : 1), WCH = 4 / WKH;
  static constexpr int XC = MPAIR == 2 ? 16 : 32, GC = NPAIR == 2 ? 16 : 32;
  // 64-voxel tiles only while x slab + gy tile stay near 64 KB: with 32 input channels per voxel the wide tile needs ~90 KB,
  // i.e. ONE block (one wave per SIMD) per CU, and the sweep ran at a third of the MFMA rate with nothing to overlap
  static constexpr int TH = 8, TW = (sizeof(T) == 2 && MPAIR == 2) ? 64 : 32, NCHK = TW / 16;
  static constexpr int GR = NPAIR == 2 ? TH + 2 : TH;
  static constexpr int LR = NPAIR == 2 ? TH + 2 * JHN - 1 : TH + KS - 1, LP = TW + KS - 1;
  static constexpr int STEPS = NPAIR == 2 ? TH + 1 : TH;
  // gy row pitch in LDS.  Paired N reads rows i and i+1 in the same instruction: +128 B keeps them on different banks.
  static constexpr int GRP = TW * GC + (NPAIR == 2 ? 128 / (int)sizeof(T) : 0);
  static constexpr size_t SMEM = ((size_t)LR * LP * XC + (size_t)GR * GRP) * sizeof(T);
};

// k-major fragment of a [rows = voxels][pitch] LDS image: lane (col = lane&31, hh = lane>>5) gets, for column `colbase + (col&15)`,
// the 8 voxels vox0(+shift for the upper 16 columns when SHIFT) + 8*hh .. +7.
template <int PITCH, typename T16> requires (sizeof(T16) == 2)
__device__ __forceinline__ Frag8<T16> ld_kmajor32(const T16* img, int vox_lo, int vox_hi, int col_lo, int col_hi, int lane) {
  // lanes 16..31 / 48..63 (upper 16 columns) use (vox_hi, col_hi); the others (vox_lo, col_lo)
  const int upper = (lane >> 4) & 1, hh = lane >> 5, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
  const T16* a = img + ((upper ? vox_hi : vox_lo) + 8 * hh + qq) * PITCH + (upper ? col_hi : col_lo) + 4 * p;
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + 4 * PITCH));
  Frag8<T16> f;
  f.u[0] = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  f.u[1] = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  f.u[2] = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  f.u[3] = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return f;
}
template <int PITCH>
__device__ __forceinline__ Frag8<float> ld_kmajor32(const float* img, int vox_lo, int vox_hi, int col_lo, int col_hi, int lane) {
  const int upper = (lane >> 4) & 1, hh = lane >> 5, i16 = lane & 15;
  const float* a = img + ((upper ? vox_hi : vox_lo) + 8 * hh) * PITCH + (upper ? col_hi : col_lo) + i16;
  Frag8<float> f;
#pragma unroll
  for (int j = 0; j < 8; j++) f.v[j] = a[j * PITCH];
  return f;
}

// 16 consecutive voxels (k-major) of one column per lane: the window every kw tap of a 16-voxel chunk reads from.  Two
// aligned fragment loads (voxels +0 and +8) fill it; the per-tap fragments are then register sub-ranges (even voxel
// offsets) or sub-ranges of a copy shifted by one voxel (odd offsets; 7 v_perm/v_alignbit per window) instead of one more
// pair of LDS transpose reads per tap -- the x operand was 7 of the 8 fragment loads per 7 MFMAs of a 7x7x7 layer.
// With M pairing the odd tap of a pair lives on the upper 16 columns, whose lanes simply READ their window one voxel later.
template <typename T> struct Win16;
template <> struct Win16<bf16_t> { unsigned w[8], wo[7]; };
template <> struct Win16<f16_t> { unsigned w[8], wo[7]; };
template <> struct Win16<float> { float w[16]; };
template <typename T16> requires (sizeof(T16) == 2)
__device__ __forceinline__ Win16<T16> make_win(const Frag8<T16>& a, const Frag8<T16>& b) {
  Win16<T16> W;
#pragma unroll
  for (int r = 0; r < 4; r++) { W.w[r] = a.u[r]; W.w[4 + r] = b.u[r]; }
  return W;
}
__device__ __forceinline__ Win16<float> make_win(const Frag8<float>& a, const Frag8<float>& b) {
  Win16<float> W;
#pragma unroll
  for (int j = 0; j < 8; j++) { W.w[j] = a.v[j]; W.w[8 + j] = b.v[j]; }
  return W;
}
// the copy shifted by one voxel; called AFTER the MFMAs of the previous window have been issued, so that the LDS
// latency of the window's loads is not exposed in front of them
template <bool ODD, typename T16> requires (sizeof(T16) == 2)
__device__ __forceinline__ void win_finish(Win16<T16>& W) {
#pragma unroll
  for (int r = 0; r < 7; r++) W.wo[r] = ODD ? __builtin_amdgcn_alignbit(W.w[r + 1], W.w[r], 16) : 0u;
}
template <bool ODD> __device__ __forceinline__ void win_finish(Win16<float>&) {}
// fragment = voxels [S, S + 8) of the window (S compile-time)
template <int S, typename T16> requires (sizeof(T16) == 2)
__device__ __forceinline__ Frag8<T16> win_frag(const Win16<T16>& W) {
  Frag8<T16> f;
#pragma unroll
  for (int r = 0; r < 4; r++) f.u[r] = (S & 1) ? W.wo[S / 2 + r] : W.w[S / 2 + r];
  return f;
}
template <int S>
__device__ __forceinline__ Frag8<float> win_frag(const Win16<float>& W) {
  Frag8<float> f;
#pragma unroll
  for (int j = 0; j < 8; j++) f.v[j] = W.w[S + j];
  return f;
}

template <typename T, int KS, int NPAIR, int MPAIR>
__global__ void __launch_bounds__(256, 2) k_wgrad_tiled(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ dwt, WgtGeom g) {
  using C = WgtCfg<T, KS, NPAIR, MPAIR>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = (T*)smem_raw;
  T* gs = xs + (size_t)C::LR * C::LP * C::XC;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;   // wv in an SGPR: the (i, c) counters stay scalar
  // 1-D grid, decoded XCD-aware: the KS * zdim blocks that sweep the SAME unit range (all kd, channel tiles and kh groups)
  // sit on one XCD (ids with equal id % 8), so the gy tile and the overlapping x slices they all read are fetched into
  // that XCD's L2 once (round-robin order: 4.3 GB of fabric reads for 0.4 GB of operands on the 32->16 7x7x7 layer).
  int kd, z, yb;
  {
    const int L = blockIdx.x, inner = KS * g.zdim;
    if (g.ydim % 8 == 0) { const int xcd = L & 7, slot = L >> 3, c = slot % inner; yb = (slot / inner) * 8 + xcd; kd = c % KS; z = c / KS; }
    else { kd = L % KS; const int q = L / KS; yb = q % g.ydim; z = q / g.ydim; }
  }
  const int khg = z % g.KHG; z /= g.KHG; const int nt = z % g.NTn; const int mt = z / g.NTn;
  const int khw = wv % C::WKH, chw = wv / C::WKH;
  const int jh = khg * C::WKH + khw;
  const bool active = jh < C::JHN;
  // lane-constant element offsets of the k-major (transposing) LDS reads: lane = 16*g + 4*qq + p supplies row qq, columns
  // 4p..4p+3 of an 4x16 block; the upper 16 columns of a 32-wide operand are the second channel half, or (pairing) a
  // shifted copy: x by one voxel (done in registers, see Win16), gy by one image row.
  const int tr_i16 = lane & 15, tr_up = (lane >> 4) & 1;
  const int x_lane = (8 * hh + (tr_i16 >> 2) + (MPAIR == 2 ? tr_up : 0)) * C::XC + (MPAIR == 2 ? 0 : 16 * tr_up) + 4 * (tr_i16 & 3);
  const int g_lane = (8 * hh + (tr_i16 >> 2)) * C::GC + 4 * (tr_i16 & 3) + (NPAIR == 2 ? (tr_up ? 0 : C::GRP) : 16 * tr_up);
  const int units = g.N * g.D * g.tiles_h, per = (units + g.ydim - 1) / g.ydim;
  const int u0 = yb * per, u1 = min(units, u0 + per);
  const int cbase_x = mt * C::XC, cbase_g = nt * C::GC;

  v16f acc[C::KWT];
#pragma unroll
  for (int k = 0; k < C::KWT; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[k][e] = 0.f;

  constexpr int XPV = C::XC / 8, GPV = C::GC / 8;
  // fast staging: bf16, whole channel tiles, 16-byte aligned voxel rows (block-uniform)
  // 16-byte pieces that do not exist in memory (beyond the row pitch) are staged as zeros; gradients of channels >= Cin / Cout
  // that DO exist as padding are computed and discarded below
  const bool fast = sizeof(T) == 2 && (g.ldx % 8 == 0) && (g.ldgy % 8 == 0) &&
                    (((uintptr_t)x & 15) == 0) && (((uintptr_t)gy & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 8 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0))) &&
                    (g.tiles_w == 1 || g.W % C::TW == 0) &&                   // one tile width per launch (piece coordinates are precomputed)
                    (int64_t)g.H * g.W * max(max(g.ldx, g.ldx2), g.ldgy) < (1ll << 30);   // 32-bit offsets inside a plane
  // ---- tile iteration: (u, tw) pairs whose input depth slice exists
  struct Tile { int n, d, id, h0, w0, nchk, lpn; };
  // A unit names the INPUT depth slice: the KS blocks (one per kd) that sweep the same unit range then stage the same x slab
  // at the same time (and gy tiles KS steps apart), so that slab is fetched into the XCD's L2 once; enumerating output slices
  // spread the KS uses of a slab over KS steps and the L2 could not hold them (3.4 GB fetched for 0.4 GB of operands).
  auto tile_ok = [&](int u) { const int id = (u / g.tiles_h) % g.D, d = id - kd + C::PAD; return d >= 0 && d < g.D; };
  auto tile_of = [&](int u, int tw) {
    Tile t; const int th = u % g.tiles_h, nd = u / g.tiles_h;
    t.id = nd % g.D; t.n = nd / g.D; t.d = t.id - kd + C::PAD; t.h0 = th * C::TH; t.w0 = tw * C::TW;
    t.nchk = min(C::NCHK, (g.W - t.w0 + 15) >> 4);             // 16-voxel chunks that hold real output positions
    t.lpn = t.nchk * 16 + KS - 1;                                // slab positions actually read
    return t;
  };
  auto advance = [&](int& u, int& tw) { if (++tw >= g.tiles_w) { tw = 0; u++; while (u < u1 && !tile_ok(u)) u++; } };
  int cu = u0, ctw = 0;
  while (cu < u1 && !tile_ok(cu)) cu++;

  // ---- register-staged tile loads (fast path): ALL 16-byte pieces of one tile live in registers, requested while the
  // previous tile is being swept and written to LDS after the barrier that ends that sweep (the synchronous
  // load -> store -> sweep sequence left the MFMAs idle for the whole staging time: 0.5 of 1.25 ms on 16->16 at 128^3)
  constexpr int PX = (C::LR * C::LP * XPV + 255) / 256, PG = (C::GR * C::TW * GPV + 255) / 256;
  v4u rx[PX], rg[PG];
  const int cpiece = cbase_x + (tid % XPV) * 8;                  // this thread's 8-channel piece of x: fixed for the launch
  const bool xsecond = g.x2 && cpiece >= g.csplit;
  const T* xsrc = (xsecond ? (const T*)g.x2 : x) + cpiece - (xsecond ? g.csplit : 0);
  const int ldsrc = xsecond ? g.ldx2 : g.ldx;
  const T* gsrc = gy + cbase_g + (tid % GPV) * 8;
  const bool x_exists = cpiece + 8 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx), g_exists = cbase_g + (tid % GPV) * 8 + 8 <= g.ldgy;
  // Piece coordinates are launch constants on the fast path (it requires one tile width for the whole launch): walk them
  // ONCE here, packed (row << 16 | position); -1 = no piece.  Per tile a piece then costs two range checks and one 32-bit
  // multiply-add -- the per-tile coordinate walk (divergent carry loops, 64-bit address products) used to issue as many
  // instructions as the MFMA sweep it was supposed to hide behind.
  const int nchk_c = min(C::NCHK, (g.W + 15) >> 4), lpn_c = nchk_c * 16 + KS - 1, gwn_c = nchk_c * 16;
  int pkx[PX], pkg[PG];
  {
    const int totx = C::LR * lpn_c * XPV, totg = C::GR * gwn_c * GPV;
    int lp = (tid / XPV) % lpn_c, lr = (tid / XPV) / lpn_c;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      pkx[j] = (j * 256 + tid < totx) ? (lr << 16 | lp) : -1;
      lp += 256 / XPV; while (lp >= lpn_c) { lp -= lpn_c; lr++; }
    }
    int gw = (tid / GPV) % gwn_c, gr = (tid / GPV) / gwn_c;
#pragma unroll
    for (int j = 0; j < PG; j++) {
      pkg[j] = (j * 256 + tid < totg) ? (gr << 16 | gw) : -1;
      gw += 256 / GPV; while (gw >= gwn_c) { gw -= gwn_c; gr++; }
    }
  }
  auto issue = [&](const Tile& t) {
    const T* xplane = xsrc + (((int64_t)t.n * g.D + t.id) * g.H) * (int64_t)g.W * ldsrc;      // one (n, depth) plane: 32-bit offsets inside
    const int ihb = t.h0 - C::PAD, iwb = t.w0 - C::PAD;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      const int ih = ihb + (pkx[j] >> 16), iw = iwb + (pkx[j] & 0xffff);
      const bool ok = x_exists && pkx[j] >= 0 && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      v4u v = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
      rx[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
    const T* gplane = gsrc + (((int64_t)t.n * g.D + t.d) * g.H) * (int64_t)g.W * g.ldgy;
    const int ohb = t.h0 - (NPAIR == 2 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < PG; j++) {
      const int gr = pkg[j] >> 16, oh = ohb + gr, ow = t.w0 + (pkg[j] & 0xffff);
      const bool ok = g_exists && pkg[j] >= 0 && oh >= t.h0 && oh < t.h0 + C::TH && oh < g.H && ow < g.W;
      v4u v = *(const v4u*)(gplane + (ok ? (oh * g.W + ow) * g.ldgy : 0));
      rg[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < PX; j++)
      if (pkx[j] >= 0) *(v4u*)(xs + ((pkx[j] >> 16) * C::LP + (pkx[j] & 0xffff)) * C::XC + (tid % XPV) * 8) = rx[j];
#pragma unroll
    for (int j = 0; j < PG; j++)
      if (pkg[j] >= 0) *(v4u*)(gs + (pkg[j] >> 16) * C::GRP + (pkg[j] & 0xffff) * C::GC + (tid % GPV) * 8) = rg[j];
  };
  auto stage_slow = [&](const Tile& t) {                           // guarded loads (fp32, ragged channel tiles, unaligned rows)
    const int n = t.n, d = t.d, id = t.id, h0 = t.h0, w0 = t.w0;
        for (int p = tid; p < C::LR * C::LP * XPV; p += 256) {
          int part = p % XPV, v = p / XPV, lp = v % C::LP, lr = v / C::LP;
          int ih = h0 - C::PAD + lr, iw = w0 - C::PAD + lp, c = cbase_x + part * 8;
          int nv = g.Cin - c; nv = nv > 8 ? 8 : nv;
          bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
          const bool second = g.x2 && c >= g.csplit;
          if (!second && g.x2 && c + nv > g.csplit) nv = g.csplit - c;
          const T* xsrc = second ? (const T*)g.x2 : x;
          const int ldsrc = second ? g.ldx2 : g.ldx, cc = c - (second ? g.csplit : 0);
          Frag8<T> f = ok ? frag_load(xsrc + ((((int64_t)n * g.D + id) * g.H + ih) * g.W + iw) * ldsrc + cc, nv) : frag_zero<T>();
          frag_st_lds(xs + (size_t)v * C::XC + part * 8, f);
        }
        for (int p = tid; p < C::GR * C::TW * GPV; p += 256) {
          int part = p % GPV, v = p / GPV, gw = v % C::TW, gr = v / C::TW;
          int oh = h0 + gr - (NPAIR == 2 ? 1 : 0), ow = w0 + gw, c = cbase_g + part * 8;
          int nv = g.Cout - c; nv = nv > 8 ? 8 : nv;
          bool ok = oh >= h0 && oh < h0 + C::TH && oh < g.H && ow < g.W && nv > 0;
          Frag8<T> f = ok ? frag_load(gy + ((((int64_t)n * g.D + d) * g.H + oh) * g.W + ow) * g.ldgy + c, nv) : frag_zero<T>();
          frag_st_lds(gs + (size_t)gr * C::GRP + gw * C::GC + part * 8, f);
        }
  };
  auto sweep_tile = [&](const Tile& tl) {
    const int nchk = tl.nchk;
      if (active && g.dbg != 2) {
        // flat loop over (gy row i, 16-voxel chunk c) with the NEXT pair's operands in flight: two named register sets
        const int ncw = (nchk - chw + C::WCH - 1) / C::WCH, nit = C::STEPS * ncw;     // chunks of this wave x rows
        constexpr bool WODD = MPAIR == 1 && KS > 1;        // odd voxel shifts are needed only without M pairing
        // (i, c) advance with wave-uniform (scalar) counters; the lane part of every LDS address is constant for the launch,
        // so one VALU add per image and immediate offsets cover the six transpose reads of an iteration (the address
        // arithmetic used to cost several times the issue slots of the MFMAs it fed).
        int it_i = 0, it_c = chw;
        auto load_ic = [&](Frag8<T>& fb, Win16<T>& W, Frag8<T>& r0, Frag8<T>& r1) {
          const int i = it_i, c = it_c;
          it_c += C::WCH; if (it_c >= nchk) { it_c = chw; it_i++; }
          const int lr = i + (NPAIR == 2 ? 2 * jh : jh), vbase = lr * C::LP + 16 * c;
          if constexpr (sizeof(T) == 2) {
            const T* ga = gs + g_lane + (i * C::GRP + 16 * c * C::GC);
            const T* xa = xs + x_lane + vbase * C::XC;
            fb = tr_pair<4 * C::GC, T>(ga);
            r0 = tr_pair<4 * C::XC, T>(xa);
            if (KS > 1) r1 = tr_pair<4 * C::XC, T>(xa + 8 * C::XC);                    // reads past a row end only feed unused window slots
          } else {
            // gy rows are GRP apart: express the row offset in "voxels" of pitch GC (GRP is a multiple of GC)
            if (NPAIR == 2) fb = ld_kmajor32<C::GC>(gs, (i + 1) * (C::GRP / C::GC) + 16 * c, i * (C::GRP / C::GC) + 16 * c, 0, 0, lane);
            else fb = ld_kmajor32<C::GC>(gs, i * (C::GRP / C::GC) + 16 * c, i * (C::GRP / C::GC) + 16 * c, 0, 16, lane);
            const int chi = MPAIR == 2 ? 0 : 16;
            const int vup = vbase + (MPAIR == 2 ? 1 : 0);
            r0 = ld_kmajor32<C::XC>(xs, vbase, vup, 0, chi, lane);
            if (KS > 1) r1 = ld_kmajor32<C::XC>(xs, vbase + 8, vup + 8, 0, chi, lane);
          }
          if (!WODD) W = make_win(r0, KS == 1 ? r0 : r1);          // nothing to derive: the loads ARE the window
        };
        auto sweep = [&](const Frag8<T>& fb, const Win16<T>& W) {
          [&]<int... K>(std::integer_sequence<int, K...>) {
            ((acc[K] = mma32(win_frag<(MPAIR == 2 ? 2 * K : K)>(W), fb, acc[K])), ...);
          }(std::make_integer_sequence<int, C::KWT>{});
        };
        // Per set: the raw loads (fb, r0, r1) are issued one iteration ahead; the window (copies + one-voxel-shifted copy)
        // is built from them AFTER the previous window's MFMAs have been issued, so no LDS latency sits in front of an MFMA.
        Frag8<T> fbA, fbB, r0, r1; Win16<T> WA, WB;
        auto finish = [&](Win16<T>& W) { if (WODD) { W = make_win(r0, r1); win_finish<true>(W); } };
        if (nit > 0) { load_ic(fbA, WA, r0, r1); finish(WA); }
#pragma unroll 1
        for (int t = 0; t < nit; t += 2) {
          if (t + 1 < nit) load_ic(fbB, WB, r0, r1);
          sweep(fbA, WA);
          if (t + 1 < nit) {
            finish(WB);
            if (t + 2 < nit) load_ic(fbA, WA, r0, r1);
            sweep(fbB, WB);
            if (t + 2 < nit) finish(WA);
          }
        }
      }
  };

  if (fast && g.dbg != 1) {
    if (cu < u1) issue(tile_of(cu, ctw));
    while (cu < u1) {
      const Tile t = tile_of(cu, ctw);
      lds_barrier();                                               // every wave is done sweeping the previous tile
      commit();
      int nu = cu, ntw = ctw; advance(nu, ntw);
      if (nu < u1) issue(tile_of(nu, ntw));                        // in flight during the sweep below (lds_barrier does not drain it)
      lds_barrier();
      sweep_tile(t);
      cu = nu; ctw = ntw;
    }
  } else {
    while (cu < u1) {
      const Tile t = tile_of(cu, ctw);
      __syncthreads();
      if (g.dbg != 1) stage_slow(t);
      __syncthreads();
      sweep_tile(t);
      advance(cu, ctw);
    }
  }
  if (!active) return;
  // C/D: col (N index) = lane&31, row (M index) = (e&3) + 8*(e>>2) + 4*hh
  const int ncol = lane & 31;
  int co, kh;
  if (NPAIR == 2) { co = nt * 16 + (ncol & 15); kh = 2 * jh + (ncol >> 4); } else { co = nt * 32 + ncol; kh = jh; }
  if (kh >= KS || co >= g.Cout) return;
  dwt += ((int64_t)yb * C::WCH + chw) * g.slab;          // (deterministic mode: one slab per voxel share AND per wave group that shares its taps)
#pragma unroll
  for (int k = 0; k < C::KWT; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = (e & 3) + 8 * (e >> 2) + 4 * hh;
      int ci, kw;
      if (MPAIR == 2) { ci = mt * 16 + (m & 15); kw = 2 * k + (m >> 4); } else { ci = mt * 32 + m; kw = k; }
      if (kw < KS && ci < g.Cin) atomicAdd(dwt + ((int64_t)((kd * KS + kh) * KS + kw) * g.Cin + ci) * g.Cout + co, acc[k][e]);
    }
}

// ================================================================================================ weight gradient, Cout <= 16
// The 128^3-level layers have 16 output channels.  The kernel above fills the 32-wide N of a 32x32x16 MFMA with two kh taps
// (and M with two kw taps when Cin <= 16): 49 taps occupy 64 tap slots, and the shifted tap costs a ninth accumulator row --
// 32 % of the MFMA cycles are padding.  Here the MFMA is v_mfma_f32_16x16x32: M = 16 input channels, N = 16 output channels,
// K = 32 voxels along W, ONE tap per MFMA and no padding at all.  Operand traffic stays low because (a) the x operand of all
// KS kw taps of a (row, 32-voxel chunk) is ONE 16-voxel register window (two transpose-read pairs, shifts by v_alignbit /
// register sub-ranges) and (b) a gy fragment (one transpose-read pair) feeds the KS taps of a kh: 12 LDS reads per 28 MFMAs.  A wave owns a kh subset (4 + 3 of 7, 2 + 1 of 3) of one 32-voxel chunk and keeps its nkh x KS accumulators
// (4 VGPRs each) for the block's whole voxel share; staging, tile order, grid decode and the scratch layout are those of
// k_wgrad_tiled.
template <typename T, int KS, int NT16>
struct Wg16Cfg {
  static constexpr int PAD = KS / 2, TH = 8, TW = 64, XC = 16, GC = 16 * NT16;      // NT16 = 2: Cout in (16, 32], Cin <= 16
  static constexpr int LP = TW + KS - 1, LR = TH + KS - 1, GRP = TW * GC + 64;      // +128 B: gy rows on different banks
  static constexpr int KPW = 2;                                                    // kh taps per wave (accumulators: KPW x KS x 4 VGPRs)
  static constexpr size_t SMEM = ((size_t)LR * LP * XC + (size_t)TH * GRP) * sizeof(T);
};

template <typename T, int KS, int NT16>
__global__ void __launch_bounds__(256, 2) k_wgrad_cc16(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ dwt, WgtGeom g) {
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  using C = Wg16Cfg<T, KS, NT16>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = (T*)smem_raw;
  T* gs = xs + (size_t)C::LR * C::LP * C::XC;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4;
  int kd, mt, yb;
  {
    const int L = blockIdx.x, inner = KS * g.zdim;
    if (g.ydim % 8 == 0) { const int xcd = L & 7, slot = L >> 3, c = slot % inner; yb = (slot / inner) * 8 + xcd; kd = c % KS; mt = c / KS; }
    else { kd = L % KS; const int r = L / KS; yb = r % g.ydim; mt = r / g.ydim; }
  }
  // wave roles: kh0 / nkh = the wave's kh taps, [c_lo, c_hi) = its 32-voxel chunks of a tile row
  const int kh0 = KS == 7 ? 2 * wv : 2 * (wv & 1), nkh = min(C::KPW, KS - kh0);
  const int c_lo = KS == 7 ? 0 : (wv >> 1), c_hi = KS == 7 ? 2 : c_lo + 1;
  const int units = g.N * g.D * g.tiles_h, per = (units + g.ydim - 1) / g.ydim;
  const int u0 = yb * per, u1 = min(units, u0 + per);
  const int cbase_x = mt * C::XC;
  // lane part of the k-major (transposing) reads: k group q = lane>>4 holds voxels 8q..8q+7, lane&15 = channel
  const int i16 = lane & 15, x_lane = (8 * q + (i16 >> 2)) * C::XC + 4 * (i16 & 3), g_lane = (8 * q + (i16 >> 2)) * C::GC + 4 * (i16 & 3);

  v4f acc[C::KPW][NT16][KS];
#pragma unroll
  for (int a = 0; a < C::KPW; a++)
#pragma unroll
    for (int n_ = 0; n_ < NT16; n_++)
#pragma unroll
      for (int b = 0; b < KS; b++) acc[a][n_][b] = (v4f){0.f, 0.f, 0.f, 0.f};

  struct Tile { int n, d, id, h0, w0, nch; };
  auto tile_ok = [&](int u) { const int id = (u / g.tiles_h) % g.D, d = id - kd + C::PAD; return d >= 0 && d < g.D; };
  auto tile_of = [&](int u, int tw) {
    Tile t; const int th = u % g.tiles_h, nd = u / g.tiles_h;
    t.id = nd % g.D; t.n = nd / g.D; t.d = t.id - kd + C::PAD; t.h0 = th * C::TH; t.w0 = tw * C::TW;
    t.nch = min(2, (g.W - t.w0 + 31) >> 5);
    return t;
  };
  auto advance = [&](int& u, int& tw) { if (++tw >= g.tiles_w) { tw = 0; u++; while (u < u1 && !tile_ok(u)) u++; } };
  int cu = u0, ctw = 0;
  while (cu < u1 && !tile_ok(cu)) cu++;

  // register-staged tiles (see k_wgrad_tiled): all 16-byte pieces of the next tile are in flight during the sweep
  constexpr int XPV = 2, GPV = 2 * NT16;
  constexpr int PX = (C::LR * C::LP * XPV + 255) / 256, PG = (C::TH * C::TW * GPV + 255) / 256;
  v4u rx[PX], rg[PG];
  const int cpiece = cbase_x + (tid % XPV) * 8;
  const bool xsecond = g.x2 && cpiece >= g.csplit;
  const T* xsrc = (xsecond ? (const T*)g.x2 : x) + cpiece - (xsecond ? g.csplit : 0);
  const int ldsrc = xsecond ? g.ldx2 : g.ldx;
  const T* gsrc = gy + (tid % GPV) * 8;
  const bool x_exists = cpiece + 8 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx), g_exists = (tid % GPV) * 8 + 8 <= g.ldgy;
  int pkx[PX], pkg[PG];
  {
    int lp = (tid / XPV) % C::LP, lr = (tid / XPV) / C::LP;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      pkx[j] = (j * 256 + tid < C::LR * C::LP * XPV) ? (lr << 16 | lp) : -1;
      lp += 256 / XPV;
#pragma unroll
      for (int w_ = 0; w_ < (128 + C::LP - 1) / C::LP; w_++) if (lp >= C::LP) { lp -= C::LP; lr++; }
    }
#pragma unroll
    for (int j = 0; j < PG; j++) { const int v = tid / GPV + j * (256 / GPV); pkg[j] = (v < C::TH * C::TW) ? ((v / C::TW) << 16 | (v % C::TW)) : -1; }
  }
  auto issue = [&](const Tile& t) {
    const T* xplane = xsrc + (((int64_t)t.n * g.D + t.id) * g.H) * (int64_t)g.W * ldsrc;
    const int ihb = t.h0 - C::PAD, iwb = t.w0 - C::PAD;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      const int ih = ihb + (pkx[j] >> 16), iw = iwb + (pkx[j] & 0xffff);
      const bool ok = x_exists && pkx[j] >= 0 && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      v4u v = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
      rx[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
    const T* gplane = gsrc + (((int64_t)t.n * g.D + t.d) * g.H) * (int64_t)g.W * g.ldgy;
#pragma unroll
    for (int j = 0; j < PG; j++) {
      const int oh = t.h0 + (pkg[j] >> 16), ow = t.w0 + (pkg[j] & 0xffff);
      const bool ok = g_exists && pkg[j] >= 0 && oh < g.H && ow < g.W;
      v4u v = *(const v4u*)(gplane + (ok ? (oh * g.W + ow) * g.ldgy : 0));
      rg[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < PX; j++)
      if (pkx[j] >= 0) *(v4u*)(xs + ((pkx[j] >> 16) * C::LP + (pkx[j] & 0xffff)) * C::XC + (tid % XPV) * 8) = rx[j];
#pragma unroll
    for (int j = 0; j < PG; j++)
      if (pkg[j] >= 0) *(v4u*)(gs + (pkg[j] >> 16) * C::GRP + (pkg[j] & 0xffff) * C::GC + (tid % GPV) * 8) = rg[j];
  };
  // sweep of one 32-voxel chunk of one tile: every x row rho is loaded once as a register window and meets the gy rows i = rho - kh
  // of the wave's taps (wave-uniform predicates; the accumulator index kk is static)
  auto sweep = [&](int chunk) {
    const T* gb = gs + g_lane + 32 * chunk * C::GC;
    const T* xb = xs + x_lane + 32 * chunk * C::XC;
    // (a hand-pipelined variant -- next row's loads issued before this row's MFMAs, rolling gy registers -- measured 8 % slower:
    // the second wave on the SIMD already hides the LDS latency and the register copies cost issue slots)
#pragma unroll 1
    for (int rho = kh0; rho < min(C::LR, kh0 + nkh - 1 + C::TH); rho++) {
      Win16<T> W = make_win(tr_pair<4 * C::XC, T>(xb + rho * C::LP * C::XC), tr_pair<4 * C::XC, T>(xb + (rho * C::LP + 8) * C::XC));
      win_finish<(KS > 1)>(W);
#pragma unroll
      for (int kk = 0; kk < C::KPW; kk++) {
        const int i = rho - kh0 - kk;
        if (kk >= nkh || i < 0 || i >= C::TH) continue;
#pragma unroll
        for (int n_ = 0; n_ < NT16; n_++) {
          const Frag8<T> gf = tr_pair<4 * C::GC, T>(gb + i * C::GRP + 16 * n_);
          [&]<int... KW>(std::integer_sequence<int, KW...>) {
            ((acc[kk][n_][KW] = mma16(win_frag<KW>(W), gf, acc[kk][n_][KW])), ...);
          }(std::make_integer_sequence<int, KS>{});
        }
      }
    }
  };
  const bool fast = (g.ldx % 8 == 0) && (g.ldgy % 8 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)gy & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 8 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0)));
  if (!fast) return;                                                // the launcher only selects this kernel for aligned operands
  const bool dbg_nostage = g.dbg & 1, dbg_nosweep = g.dbg & 2;       // experiments only (env DP_DBG)
  if (cu < u1 && !dbg_nostage) issue(tile_of(cu, ctw));
  while (cu < u1) {
    const Tile t = tile_of(cu, ctw);
    lds_barrier();
    if (!dbg_nostage) commit();
    int nu = cu, ntw = ctw; advance(nu, ntw);
    if (nu < u1 && !dbg_nostage) issue(tile_of(nu, ntw));
    lds_barrier();
    if (nkh > 0 && !dbg_nosweep)
      for (int c = c_lo; c < min(c_hi, t.nch); c++) sweep(c);
    cu = nu; ctw = ntw;
  }
  if (g.dbg & 4) return;
  dwt += (int64_t)yb * g.slab;
  // C/D of the 16x16 MFMA: col (co) = lane&15, row (ci) = 4*(lane>>4) + e
#pragma unroll
  for (int kk = 0; kk < C::KPW; kk++) {
    if (kk >= nkh) continue;
#pragma unroll
    for (int n_ = 0; n_ < NT16; n_++) {
      const int co = 16 * n_ + (lane & 15);
      if (co >= g.Cout) continue;
#pragma unroll
      for (int kw = 0; kw < KS; kw++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int ci = mt * 16 + 4 * q + e;
          if (ci < g.Cin) atomicAdd(dwt + ((int64_t)((kd * KS + kh0 + kk) * KS + kw) * g.Cin + ci) * g.Cout + co, acc[kk][n_][kw][e]);
        }
    }
  }
}

template <typename T, int KS, int NT16>
static int launch_wg16(const void* x, const void* gy, float* ws, WgtGeom g, hipStream_t s, int max_slabs, int* nslab) {
  using C = Wg16Cfg<T, KS, NT16>;
  auto kern = k_wgrad_cc16<T, KS, NT16>;
  if (C::SMEM > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    if (e != hipSuccess) { dp_set_error("wgrad_cc16: cannot raise dynamic LDS to %zu: %s", C::SMEM, hipGetErrorString(e)); return 1; }
  }
  g.tiles_h = cdiv(g.H, C::TH); g.tiles_w = cdiv(g.W, C::TW);
  g.MT = cdiv(g.Cin, C::XC); g.NTn = 1; g.KHG = 1;
  int zdim = g.MT, units = g.N * g.D * g.tiles_h;
  static int occ = 0, ncu = 0;
  if (!occ) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, C::SMEM) != hipSuccess || occ < 1) occ = 2;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  int want = (ncu * occ) / (KS * zdim); if (want < 1) want = 1;
  int ydim = units < want ? units : want;
  // (always rounding to a multiple of 8 for the XCD-aware decode: 0.84 -> 0.81 ms on 16->16, nothing on 32->16; rounding up: 1.5x slower)
  if (max_slabs && ydim > max_slabs) ydim = max_slabs;
  if (ydim >= 8 && (ydim & 7) * 20 <= ydim) ydim &= ~7;
  g.ydim = ydim; g.zdim = zdim; *nslab = max_slabs ? ydim : 1;
  hipLaunchKernelGGL(kern, dim3(KS * ydim * zdim), dim3(256), C::SMEM, s, (const T*)x, (const T*)gy, ws, g);
  return 0;
}

__global__ void __launch_bounds__(256) k_wgrad_unpack(float* __restrict__ dwt, float* __restrict__ dw, int taps, int Cin, int Cout, int64_t s_co, int64_t s_ci,
                                                      int64_t s_tap, int rezero, int nslab, int64_t slab) {
  // [tap][ci][co] scratch -> dw[co*s_co + ci*s_ci + tap*s_tap] as an LDS-tiled transpose of 32 taps x 32 (ci, co) pairs: reads are
  // 128-byte runs along co, writes 128-byte runs along the taps (either side alone is a 4-byte scatter: 207 us for 343 x 256 x 128).
  __shared__ float tile[32][33];
  const int64_t pairs = (int64_t)Cin * Cout;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t pair0 = (int64_t)blockIdx.x * 32; const int tap0 = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int tap = tap0 + ty + 8 * r; const int64_t pr = pair0 + tx;
    const bool in = tap < taps && pr < pairs;
    // (deterministic mode: the voxel shares' slabs added in share order)
    float v = 0.f;
    if (in) {
      float* a = dwt + (int64_t)tap * pairs + pr;
      v = *a;
      if (rezero) *a = 0.f;
      for (int sl = 1; sl < nslab; sl++) { v += a[sl * slab]; if (rezero) a[sl * slab] = 0.f; }
    }
    tile[ty + 8 * r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int64_t pr = pair0 + ty + 8 * r; const int tap = tap0 + tx;
    if (pr < pairs && tap < taps) { const int co = (int)(pr % Cout), ci = (int)(pr / Cout); dw[co * s_co + ci * s_ci + tap * s_tap] = tile[tx][ty + 8 * r]; }
  }
}

static inline bool wgt_applicable(int Cin, int Cout, int k, int stride, int pad, int dil, int shift, int W) {
  if (k == 1) return stride == 1 && pad == 0 && W >= 16 && Cin >= 8 && Cout >= 8 && Cin <= 256 && Cout <= 256;   // pointwise: HBM-bound row stream
  // (W >= 8, round 5: the 12^3 level of the 96^3 crop fell to the generic kernel -- 2 ms per 7x7x7 128 -> 128 launch at 4 x 12^3, 8.2 of the
  // 31 ms of an OAR-TRANSEG step at the reference's own crop size; a tile of 16 / 32 positions with 12 real ones still runs on the tiled kernel)
  static const int minw = [] { const char* e = getenv("DP_WGRAD_MINW"); return e ? atoi(e) : 8; }();
  return shift && (k == 3 || k == 7) && stride == 1 && dil == 1 && pad == k / 2 && W >= minw && Cin >= 1 && Cout >= 8;
}
// fp32 scratch elements needed by dp_conv3d_wgrad_tiled (0: shape not supported, use dp_conv3d_wgrad)
extern "C" int dp_conv3d_wgrad_tiled_ws_elems(int Cin, int Cout, int k, int stride, int pad, int dil, int shift, int W) {
  if (!wgt_applicable(Cin, Cout, k, stride, pad, dil, shift, W)) return 0;
  int64_t base = (int64_t)k * k * k * Cin * Cout; const int64_t hk = wgrad_hk_ws_elems(Cin, Cout, k);
  if (dp_det(DET_WGRAD)) base *= det_slabs(base);               // one slab per voxel share (see WgtGeom::slab)
  int64_t n = base > hk ? base : hk;
  if (k == 1 && Cin <= 64 && Cout <= 32) { const int64_t rw = wgrad_rows_ws_elems(Cin, Cout); if (rw > n) n = rw; }   // per-block partials of k_wgrad_rows
  return n > 2000000000LL ? 0 : (int)n;
}

template <typename T, int KS, int NPAIR, int MPAIR>
static int launch_wgt(const void* x, const void* gy, float* ws, WgtGeom g, hipStream_t s, int max_slabs, int* nslab) {
  using C = WgtCfg<T, KS, NPAIR, MPAIR>;
  auto kern = k_wgrad_tiled<T, KS, NPAIR, MPAIR>;
  if (C::SMEM > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    if (e != hipSuccess) { dp_set_error("wgrad_tiled: cannot raise dynamic LDS to %zu: %s", C::SMEM, hipGetErrorString(e)); return 1; }
  }
  g.tiles_h = cdiv(g.H, C::TH); g.tiles_w = cdiv(g.W, C::TW);
  g.MT = cdiv(g.Cin, C::XC); g.NTn = cdiv(g.Cout, C::GC); g.KHG = cdiv(C::JHN, C::WKH);
  int zdim = g.MT * g.NTn * g.KHG;
  int units = g.N * g.D * g.tiles_h;
  // one resident wave of blocks: occupancy x CUs blocks in total (a 1.5-round grid left a quarter of the chip idle in round two)
  static int occ = 0, ncu = 0;
  if (!occ) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, C::SMEM) != hipSuccess || occ < 1) occ = 2;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  int want = (ncu * occ) / (KS * zdim); if (want < 1) want = 1;
  int ydim = units < want ? units : want;
  // multiple of 8 => XCD-aware decode in the kernel; only when rounding down idles <= 5 % of the block slots (the locality is
  // worth a few per cent, an empty eighth of the chip is not)
  if (max_slabs) {
    if (max_slabs < C::WCH) { dp_set_error("wgrad_tiled: deterministic scratch too small for this shape"); return 1; }
    if (ydim > max_slabs / C::WCH) ydim = max_slabs / C::WCH;
  }
  if (ydim >= 8 && (ydim & 7) * 20 <= ydim) ydim &= ~7;
  g.ydim = ydim; g.zdim = zdim; *nslab = max_slabs ? ydim * C::WCH : 1;
  dim3 grid(KS * ydim * zdim, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), C::SMEM, s, (const T*)x, (const T*)gy, ws, g);
  return 0;
}

// shape test of the streaming pointwise weight-gradient kernel (k = 1); dtype DP_X1: fp32 rows, dW = x_hi gy_hi (the fp32x3 mode)
extern "C" int dp_conv3d_wgrad_rows_ok(int ldx, int ldgy, int64_t rows, int Cin, int Cout, int dtype) { return wgrad_rows_ok(ldx, ldgy, rows, Cin, Cout, dtype) ? 1 : 0; }
// ws: fp32 scratch of dp_conv3d_wgrad_tiled_ws_elems() elements (need not be initialised).  OVERWRITES dw (need not be initialised).
extern "C" int dp_conv3d_wgrad_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* gy, int ldgy, float* dw, float* ws,
                                      int N, int D, int H, int W, int Cin, int Cout, int k, int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream);
extern "C" int dp_conv3d_wgrad_tiled(const void* x, int ldx, const void* gy, int ldgy, float* dw, float* ws, int N, int D, int H, int W,
                                     int Cin, int Cout, int k, int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream) {
  return dp_conv3d_wgrad_tiled2(x, ldx, nullptr, 0, 0, gy, ldgy, dw, ws, N, D, H, W, Cin, Cout, k, s_co, s_ci, s_tap, dtype, stream);
}
extern "C" int dp_conv3d_wgrad_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* gy, int ldgy, float* dw, float* ws,
                                      int N, int D, int H, int W, int Cin, int Cout, int k, int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream) {
  if (!wgt_applicable(Cin, Cout, k, 1, k / 2, 1, 1, W)) DP_FAIL("wgrad_tiled: shape not supported");
  if (x2 && (csplit <= 0 || csplit >= Cin || csplit % 8)) DP_FAIL("wgrad_tiled: input split must be a multiple of 8 inside (0, Cin)");
  if (k == 1 && H * (int64_t)D * N > 2000000000LL) DP_FAIL("wgrad_tiled: too many rows");
  hipStream_t s = STREAM;
  int taps = k * k * k;
  const int64_t base = (int64_t)taps * Cin * Cout;
  const int max_slabs = dp_det(DET_WGRAD) ? det_slabs(base) : 0;
  int nslab = 1;
  if (!g_scratch_zeroed) {
    hipError_t me = hipMemsetAsync(ws, 0, (size_t)base * (max_slabs ? max_slabs : 1) * sizeof(float), s);
    if (me != hipSuccess) DP_FAIL("wgrad_tiled: memset failed: %s", hipGetErrorString(me));
  }
  WgtGeom g; g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ldx = ldx; g.ldgy = ldgy;
  g.x2 = x2; g.ldx2 = ldx2; g.csplit = csplit; g.slab = max_slabs ? base : 0;
  { const char* e = getenv("DP_DBG"); g.dbg = e ? atoi(e) : 0; }
  const int np = (Cout <= 16 && k > 1) ? 2 : 1, mp = (Cin <= 16 && k > 1) ? 2 : 1;
  if (getenv("DP_DEBUG_SLOW")) {
    int xc = mp == 2 ? 16 : 32, gc = np == 2 ? 16 : 32, tw = (dtype != DP_F32 && mp == 2) ? 64 : 32;
    int avail = x2 ? csplit + ldx2 : ldx;
    (void)xc; (void)gc; (void)avail;
    bool fast = dtype != DP_F32 && ldx % 8 == 0 && ldgy % 8 == 0 &&
                ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0 && (!x2 || (csplit % 8 == 0 && ldx2 % 8 == 0 && ((uintptr_t)x2 & 15) == 0)) &&
                (W <= tw || W % tw == 0);
    if (!fast) fprintf(stderr, "[dp slow] wgrad_tiled k=%d Cin=%d Cout=%d %dx%dx%d ldx=%d ldx2=%d csplit=%d ldgy=%d: some channel tiles stage GUARDED\n", k, Cin, Cout,
                       D, H, W, ldx, ldx2, csplit, ldgy);
  }
  if (dtype == DP_X1 && !(k == 1 && !x2 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0 && wgrad_rows_ok(ldx, ldgy, (int64_t)N * D * H * W, Cin, Cout, dtype)))
    DP_FAIL("wgrad_tiled: DP_X1 (fp32 rows, one bf16 product) is the pointwise row kernel's shape class only (dp_conv3d_wgrad_rows_ok)");
  if (k == 1 && !x2 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0 && wgrad_rows_ok(ldx, ldgy, (int64_t)N * D * H * W, Cin, Cout, dtype)) {
    // millions of voxel rows x a few dozen channels: the streaming kernel of elementwise.hip (deterministic two-stage sum, dW written directly)
    wgrad_rows_launch(x, ldx, gy, ldgy, dw, s_co, s_ci, ws, (int64_t)N * D * H * W, Cin, Cout, dtype, g_scratch_zeroed, s);
    DP_CHECK_LAUNCH("wgrad_rows"); return 0;
  }
  int rc = 0;
  const bool aligned = ldx % 8 == 0 && ldgy % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0 &&
                       (!x2 || (csplit % 8 == 0 && ldx2 % 8 == 0 && ((uintptr_t)x2 & 15) == 0)) &&
                       (int64_t)H * W * (ldx > ldgy ? (ldx > ldx2 ? ldx : ldx2) : (ldgy > ldx2 ? ldgy : ldx2)) < (1ll << 30);
  const bool cc16 = k > 1 && dtype != DP_F32 && aligned && !getenv("DP_NO_CC16");
  bool hk_done = false;
  const bool hk_wide = getenv("DP_HK_NARROW") == nullptr;         // (experiments: restrict the K-along-H kernel to Cout <= 16)
  if (k > 1 && dtype != DP_F32 && aligned && (np == 2 || hk_wide) && wgrad_hk_applicable(Cout, k, H, W, dtype)) {     // 7^3, planes >= 32 x 32: K along H (conv_wgrad_hk.hip)
    WgHkGeom hg; hg.N = N; hg.D = D; hg.H = H; hg.W = W; hg.Cin = Cin; hg.Cout = Cout; hg.ldx = ldx; hg.ldgy = ldgy;
    hg.tiles_h = hg.tiles_w = hg.MT = hg.NTn = hg.ydim = 0; hg.dbg = g.dbg; hg.x2 = x2; hg.ldx2 = ldx2; hg.csplit = csplit;
    hg.dw = dw; hg.s_co = s_co; hg.s_ci = s_ci; hg.s_tap = s_tap; hg.rezero = g_scratch_zeroed;
    hg.slab = g.slab; hg.max_slabs = max_slabs; hg.nslab_out = &nslab;
    int finished = 0;
    rc = wgrad_hk_launch(x, gy, ws, hg, k, dtype, s, &finished);
    if (rc > 0) return rc;
    if (finished) { DP_CHECK_LAUNCH("wgrad_hk"); return 0; }
    hk_done = rc == 0;                      // rc < 0: shape outside that kernel's limits, take the K-along-W kernels below
    rc = 0;
  }
  if (hk_done) {
  } else if (cc16 && np == 2) {           // Cout <= 16: one tap per 16x16x32 MFMA, no tap-pairing padding
    if (dtype == DP_BF16) rc = k == 7 ? launch_wg16<bf16_t, 7, 1>(x, gy, ws, g, s, max_slabs, &nslab) : launch_wg16<bf16_t, 3, 1>(x, gy, ws, g, s, max_slabs, &nslab);
    else rc = k == 7 ? launch_wg16<f16_t, 7, 1>(x, gy, ws, g, s, max_slabs, &nslab) : launch_wg16<f16_t, 3, 1>(x, gy, ws, g, s, max_slabs, &nslab);
  } else if (cc16 && mp == 2 && Cout <= 32) {   // Cin <= 16, Cout <= 32: two 16-wide N tiles share every x window (kw pairing wasted an eighth)
    if (dtype == DP_BF16) rc = k == 7 ? launch_wg16<bf16_t, 7, 2>(x, gy, ws, g, s, max_slabs, &nslab) : launch_wg16<bf16_t, 3, 2>(x, gy, ws, g, s, max_slabs, &nslab);
    else rc = k == 7 ? launch_wg16<f16_t, 7, 2>(x, gy, ws, g, s, max_slabs, &nslab) : launch_wg16<f16_t, 3, 2>(x, gy, ws, g, s, max_slabs, &nslab);
  } else {
#define GO(TT, KS_) do { if (np == 2 && mp == 2) rc = launch_wgt<TT, KS_, 2, 2>(x, gy, ws, g, s, max_slabs, &nslab); else if (np == 2) rc = launch_wgt<TT, KS_, 2, 1>(x, gy, ws, g, s, max_slabs, &nslab); \
                         else if (mp == 2) rc = launch_wgt<TT, KS_, 1, 2>(x, gy, ws, g, s, max_slabs, &nslab); else rc = launch_wgt<TT, KS_, 1, 1>(x, gy, ws, g, s, max_slabs, &nslab); } while (0)
  if (dtype == DP_BF16) { if (k == 7) GO(bf16_t, 7); else if (k == 3) GO(bf16_t, 3); else rc = launch_wgt<bf16_t, 1, 1, 1>(x, gy, ws, g, s, max_slabs, &nslab); }
  else if (dtype == DP_F16) { if (k == 7) GO(f16_t, 7); else if (k == 3) GO(f16_t, 3); else rc = launch_wgt<f16_t, 1, 1, 1>(x, gy, ws, g, s, max_slabs, &nslab); }
  else if (dtype == DP_F32) { if (k == 7) GO(float, 7); else if (k == 3) GO(float, 3); else rc = launch_wgt<float, 1, 1, 1>(x, gy, ws, g, s, max_slabs, &nslab); }
  else DP_FAIL("wgrad_tiled: bad dtype");
#undef GO
  }
  if (rc) return rc;
  DP_CHECK_LAUNCH("wgrad_tiled");
  int64_t pairs = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(k_wgrad_unpack, dim3((unsigned)((pairs + 31) / 32), (taps + 31) / 32), dim3(256), 0, s, ws, dw, taps, Cin, Cout, s_co, s_ci, s_tap, g_scratch_zeroed, nslab, base);
  DP_CHECK_LAUNCH("wgrad_unpack"); return 0;
}
