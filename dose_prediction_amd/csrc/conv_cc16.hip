// Convolution for the 16-output-channel layers at the full-resolution level (stride 1, "same" padding, k in {3, 7}, Cout <= 16,
// W >= 96): forward and -- with transposed + flipped weights -- data gradient.  These are the largest launches of the network
// (conv_block_7 / conv_block_3 of decoder1, blocks_MDUNet.py:64-78,98-112; UnetResBlock of skip1; C3D encoder1 / decoder1,
// c3d.py:11-22).
//
// v_mfma_f32_16x16x32_bf16 (f32: 8 x v_mfma_f32_16x16x4_f32), one instruction = 16 output positions along W x 16 output channels
// x K = 32 = TWO horizontal taps (kw = 2p, 2p+1) x one 16-channel chunk.  Against the 32x32x16 kernel of conv_tiled.hip, which for
// Cout <= 16 pairs two VERTICAL taps in N and recombines them in the epilogue:
//   * no extra accumulator row per wave (RW = 9 rows of MFMAs for 8 output rows there) and no recombination shuffles,
//   * the pairing padding is 1/8 of the MFMAs for 7x7x7 either way, so the MFMA count falls by 8/9, and the 16x16x32 shape holds
//     a higher clock under load (MI355X_MICROARCH.md, "Shape": 1.12-1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP),
//   * an accumulator tile is 4 registers per 16 positions: 8 rows x 32 positions = 64 VGPRs per depth slice, so a block owns DT = 2
//     output depth slices and every staged input slab serves both of them (7 + 1 slabs per 2 slices instead of 7 per slice for
//     7x7x7, 2 instead of 3 for 3x3x3: the 3x3x3 layers are bound by exactly this re-staging).
// LDS traffic stays low because one A fragment (16 positions of slab row rho, tap pair p) feeds the MFMAs of EVERY vertical tap kh
// whose output row rho - kh is inside the tile: 1 ds_read_b128 per 4 MFMAs.
//
// Block = 4 waves side by side (32 positions each = two 16-position M tiles), 8 output rows, DT depth slices.  Slab layout,
// swizzle, staging and the transposing epilogue are those of conv_tiled.hip.
#include "common.h"
#ifndef DP_CC16_BUF
#define DP_CC16_BUF 1      // slab staging through buffer loads with block-constant offsets (round 6); 0: the round-2 form
#endif
#include <stdlib.h>

#define STREAM ((hipStream_t)stream)

struct Cc16Geom {
  int N, D, H, W, Cin, Cout, ldx, ldy, NCH, tiles_h, tiles_w, dtiles;
  const void* x2; int ldx2, csplit;       // virtual concat of the input (channels >= csplit come from x2)
  float* stat_part; int stat_nblk;        // normalisation statistics of the output (see TiledGeom in conv_tiled.hip)
  void* y2; int ldy2, osplit;             // output channels >= osplit go to y2 (data gradient of a virtual concat)
  int wide;                               // 16-byte aligned output rows: transposing epilogue; else scalar stores
  int dbg;                                // experiments only (env DP_DBG): bit 0 skip staging, bit 1 skip the MFMA sweep, bit 2 skip the epilogue
  // DP_X3 launches: the input tensor holds 2 * x3 chunks ([x_hi | x_lo], x3 = real 16-channel chunks) while the packed weights hold
  // NCH = 3 * x3 chunks ([w_hi | w_hi | w_lo]).  Every x_hi slab is staged ONCE and swept with both of its weight blocks (chunk ch
  // and chunk 2 * x3 + ch); 0 = ordinary launch (one sweep per staged chunk).
  int x3;
};

bool cc16_applicable(int Cin, int Cout, int k, int W) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("DP_NO_CC16"); off = (e && atoi(e)) ? 1 : 0; }
  return !off && (k == 3 || k == 7) && Cout >= 8 && Cout <= 16 && Cin >= 1 && W >= 96;
}
// tap pairs (K = 32 MFMA slots) per (kd, input chunk).  3x3x3: [kwp 2][kh 3] (kw = 2 kwp, 2 kwp + 1; the odd tap of kwp 1 is padding).
// 7x7x7 (round 4): the 49 (kh, kw) taps in LINEAR order t = 7 kh + kw, slot j = taps 2j, 2j + 1: 25 slots, one padding half, instead
// of the 28 slots of [kwp 4][kh 7] (an eighth of every 7x7x7 sweep was multiplication by zero weights).  Slots with kw = 6 on the
// left straddle two kernel rows: (kh, 6) | (kh + 1, 0) -- both taps feed output row rho - kh when the right half of the A fragment
// comes from slab row rho + 1.
__host__ __device__ inline int cc16_slots(int k) { return k == 7 ? 25 : ((k + 1) / 2) * k; }
int cc16_weight_elems(int Cin, int Cout, int k) { return k * ((Cin + 15) / 16) * cc16_slots(k) * 512; }

// KS = 3: dst[kd][chunk][kwp][kh][co 16][k 32], k < 16: tap kw = 2 kwp, ci = chunk*16 + k; k >= 16: kw = 2 kwp + 1, ci = chunk*16 + k - 16.
// KS = 7: dst[kd][chunk][slot j][co 16][k 32], k < 16: tap t = 2j, k >= 16: tap t = 2j + 1 (t = 7 kh + kw; t = 49: zero).
// transposed_flipped: w'[co][ci][tap] = w[ci][co][taps-1-tap] (data gradient as a forward convolution).
template <typename T>
__global__ void k_pack_w_cc16(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int KS, int tf) {
  const int KWP = (KS + 1) / 2, NCH = (Cin + 15) / 16, taps = KS * KS * KS, NS = cc16_slots(KS);
  const int total = KS * NCH * NS * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i & 31, co = (i >> 5) & 15; int t = i >> 9;
    int kh, kw;
    if (KS == 7) { const int j = t % NS; t /= NS; const int tt = 2 * j + (k >> 4); kh = tt / 7; kw = tt % 7; if (tt >= 49) kw = KS; }
    else { kh = t % KS; t /= KS; const int kwp = t % KWP; t /= KWP; kw = 2 * kwp + (k >> 4); }
    const int ch = t % NCH; const int kd = t / NCH;
    const int ci = ch * 16 + (k & 15);
    float v = 0.f;
    if (kw < KS && co < Cout && ci < Cin) {
      const int tap = (kd * KS + kh) * KS + kw;
      v = tf ? w[((int64_t)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * Cin + ci) * taps + tap];
    }
    st_f(dst + i, v);
  }
}
int cc16_pack(const float* w, void* dst, int Cout, int Cin, int k, int tf, int dtype, hipStream_t s) {
  const int total = cc16_weight_elems(Cin, Cout, k);
  int g = (total + 255) / 256; if (g > 8192) g = 8192;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_w_cc16<T>, dim3(g), dim3(256), 0, s, w, (T*)dst, Cout, Cin, k, tf));
  DP_CHECK_LAUNCH("pack_conv_weight_cc16"); return 0;
}

// TO = type of the OUTPUT tensor: T, or float for DP_X3 launches (bf16 operands that are the hi / lo halves of fp32 values, see
// dp_split_rows: the three partial products are separate 16-channel chunks of a 3x wider "virtual" input, the sum is kept in fp32)
template <typename T, int KS, int DT, int OCC, typename TO>
__global__ void __launch_bounds__(256, OCC) k_conv_cc16(const T* __restrict__ x, const T* __restrict__ wq, const float* __restrict__ bias,
                                                   TO* __restrict__ y, Cc16Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* slab = (T*)smem_raw;
  constexpr int PAD = KS / 2, KWP = (KS + 1) / 2, RWO = 8, ROWS = RWO + KS - 1, CK = 16, TW = 128;
  constexpr int LP = (TW + KS - 1 + 7) & ~7, LR = ROWS;
  constexpr bool SWZ = sizeof(T) == 2;
  constexpr int lp_par = SWZ ? ((LP >> 3) & 1) : 0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  // XCD-aware block order (see conv_tiled.hip): every XCD owns a contiguous range of depth tiles
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; b /= g.tiles_h; const int dt = b % g.dtiles; const int n = b / g.dtiles;
  const int h0 = th * RWO, w0 = tw * TW, d0 = dt * DT;
  if (g.dbg & 8) {          // experiment (DP_DBG=8): stagger the blocks that start together on a CU so that their phases do not coincide
    const int lag = blockIdx.x < 768 ? ((blockIdx.x >> 8) % 3) * (g.dbg >> 4) : 0;       // (only the blocks of the first resident round: the later ones start when a slot frees up)
    for (int i = 0; i < lag; i++) __builtin_amdgcn_s_sleep(127);
  }
  v4f acc[DT][RWO][2];
#pragma unroll
  for (int a = 0; a < DT; a++)
#pragma unroll
    for (int i = 0; i < RWO; i++)
#pragma unroll
      for (int m = 0; m < 2; m++) acc[a][i][m] = (v4f){0.f, 0.f, 0.f, 0.f};

  constexpr int pieces = LR * LP * 2;              // 8-channel pieces of the slab
  constexpr int SU = 8;
  const int cin_in = g.x3 ? 32 * g.x3 : g.Cin;       // channels of the INPUT tensor (a DP_X3 launch reads [x_hi | x_lo])
  const bool fast = SWZ && ((g.x3 ? 2 * g.x3 : g.NCH) * 16 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx)) && (g.ldx % 8 == 0) && (((uintptr_t)x & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 16 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0))) &&
                    (int64_t)g.H * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30) &&
                    (int64_t)LR * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30);      // (a parked offset + LR row pitches stays above 2^31)
  const int st_half = tid & 1, st_lp0 = (tid >> 1) % LP, st_lr0 = (tid >> 1) / LP;
  // A operand: lane (r, q) = position r of the M tile, k-group q: q < 2 -> left tap of the pair, q >= 2 -> right tap (one voxel
  // further), q & 1 = which 8-channel half of the voxel
  const int v_lane = wv * 32 + r + (q >> 1), hsel = q & 1;
  const int lane_off = r * 32 + q * 8;             // B operand: output channel r, k-group q of the packed [16][32] tile
  constexpr int WT = 512;                          // elements per (kwp, kh) weight tile

  for (int z = d0 - PAD; z < d0 + DT + PAD; z++) {
    if (z < 0 || z >= g.D) continue;               // block-uniform: the whole depth slice is zero padding
    const int nstage = g.x3 ? 2 * g.x3 : g.NCH;
    for (int ch = 0; ch < nstage; ch++) {
      lds_barrier();                               // every wave is done with the previous slab
      if (g.dbg & 1) {
      }
#if DP_CC16_BUF
      else if (fast) {
        // Round 6: the slab as BUFFER loads with block-constant offsets.  A block owns one output tile for its whole life, so what a thread
        // stages never moves: column lp = tid / 2 of the slab (8-channel half tid & 1), rows 0 .. LR-1, plus one piece of the LP - 128 extra
        // columns.  Per piece that is ONE v_add (row pitch from an SGPR) and the load: rows above / below the volume are out of range of the
        // PLANE's descriptor (unsigned wrap-around -> zeros from the hardware), a column outside the volume parks the thread's offset at
        // 2^31.  The round-2 form (below, -DDP_CC16_BUF=0) spent ~18 VALU per piece on carry loops, four bounds compares, a 64-bit address and
        // four v_cndmask -- in a phase of its own, which the knock-outs of profiles/r06_a_3x3x3_phase_counters.md show is NOT hidden behind the
        // other blocks' sweeps.
        const bool second = g.x2 && ch * CK >= g.csplit;
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + z) * g.H) * (int64_t)g.W * ldsrc + c0;
        const uint64_t pa = (uint64_t)(uintptr_t)xplane;
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa), phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const unsigned pbytes = __builtin_amdgcn_readfirstlane((unsigned)(g.H * g.W * ldsrc * 2 - c0 * 2));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)phi << 32) | plo), 0, (int)pbytes, 0x00020000);
        const unsigned rowb = (unsigned)(g.W * ldsrc * 2);
        constexpr int EX = LP - 128;                               // extra columns beyond the 128 a thread pair owns
        const int t2 = tid >> 1, iw = w0 - PAD + t2;
        const unsigned vb = ((unsigned)iw < (unsigned)g.W) ? (unsigned)((((h0 - PAD) * g.W + iw) * ldsrc + st_half * 8) * 2) : 0x80000000u;
        const int par0 = (t2 >> 3) & 1;
        T* const dA = slab + t2 * CK + ((st_half ^ par0) * 8);        // even slab rows: + r * LP * CK (immediate)
        T* const dB = slab + t2 * CK + ((st_half ^ par0 ^ 1) * 8);    // odd slab rows (LP / 8 is odd: the swizzle bit flips with the row)
        static_assert(((LP >> 3) & 1) == 1, "row-parity form of the slab swizzle");
        // the extra piece: e = tid < EX * LR * 2 -> (half e & 1, column 128 + (e >> 1) % EX, row (e >> 1) / EX)
        const int e_half = tid & 1, e_col = 128 + (t2 % EX), e_row = t2 / EX, e_iw = w0 - PAD + e_col;
        const bool e_on = tid < EX * LR * 2;
        // (threads without an extra piece repeat their row-0 piece -- same value to the same address: no branch, no wait of its own)
        const unsigned ve = !e_on ? vb : ((unsigned)e_iw < (unsigned)g.W) ? (unsigned)((((h0 - PAD + e_row) * g.W + e_iw) * ldsrc + e_half * 8) * 2) : 0x80000000u;
        const int e_v = e_row * LP + e_col;
        T* const dE = e_on ? slab + e_v * CK + ((e_half ^ ((e_v >> 3) & 1)) * 8) : dA;
        constexpr int HALF = (LR + 1) / 2;
        v4u bufa[HALF], bufb[LR - HALF + 1];
#pragma unroll
        for (int rr = 0; rr < HALF; rr++) bufa[rr] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vb + (unsigned)rr * rowb, 0, 0));
#pragma unroll
        for (int rr = HALF; rr < LR; rr++) bufb[rr - HALF] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vb + (unsigned)rr * rowb, 0, 0));
        bufb[LR - HALF] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, ve, 0, 0));
#pragma unroll
        for (int rr = 0; rr < HALF; rr++) *(v4u*)(((rr & 1) ? dB : dA) + rr * LP * CK) = bufa[rr];
#pragma unroll
        for (int rr = HALF; rr < LR; rr++) *(v4u*)(((rr & 1) ? dB : dA) + rr * LP * CK) = bufb[rr - HALF];
        *(v4u*)dE = bufb[LR - HALF];
      }
#else
      else if (fast) {
        const bool second = g.x2 && ch * CK >= g.csplit;
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + z) * g.H) * (int64_t)g.W * ldsrc + c0 + st_half * 8;
        int lp = st_lp0, lr = st_lr0, v = tid >> 1;
        for (int p0 = 0; p0 < pieces; p0 += 256 * SU) {
          v4u buf[SU]; int vv[SU];
#pragma unroll
          for (int j = 0; j < SU; j++) {
            const int ih = h0 - PAD + lr, iw = w0 - PAD + lp;
            const bool ok = (p0 + j * 256 + tid < pieces) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            v4u t = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
            buf[j] = ok ? t : (v4u){0, 0, 0, 0};
            vv[j] = (p0 + j * 256 + tid < pieces) ? v : -1;
            v += 128; lp += 128;
#pragma unroll
            for (int c_ = 0; c_ < (128 + LP - 1) / LP; c_++) if (lp >= LP) { lp -= LP; lr++; }
          }
#pragma unroll
          for (int j = 0; j < SU; j++)
            if (vv[j] >= 0) *(v4u*)(slab + (int64_t)vv[j] * CK + (st_half ^ ((vv[j] >> 3) & 1)) * 8) = buf[j];
        }
      }
#endif
      else {
        for (int p = tid; p < pieces; p += 256) {
          int half = p & 1, v = p >> 1, lp = v % LP, lr = v / LP;
          int ih = h0 - PAD + lr, iw = w0 - PAD + lp, c = ch * CK + half * 8;
          int nv = cin_in - c; nv = nv > 8 ? 8 : nv;
          bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
          const bool second = g.x2 && c >= g.csplit;
          if (!second && g.x2 && c + nv > g.csplit) nv = g.csplit - c;
          const T* xsrc = second ? (const T*)g.x2 : x;
          const int ldsrc = second ? g.ldx2 : g.ldx, cc = c - (second ? g.csplit : 0);
          Frag8<T> f = ok ? frag_load(xsrc + ((((int64_t)n * g.D + z) * g.H + ih) * g.W + iw) * ldsrc + cc, nv) : frag_zero<T>();
          frag_st_lds(slab + (int64_t)v * CK + (half ^ (SWZ ? ((v >> 3) & 1) : 0)) * 8, f);
        }
      }
      lds_barrier();
      const int nrep = (g.x3 && ch < g.x3) ? 2 : 1;      // an x_hi slab meets w_hi (chunk ch) and w_lo (chunk 2 x3 + ch)
#pragma unroll 1
      for (int rep = 0; rep < nrep; rep++) {
      const int wch = rep ? ch + 2 * g.x3 : ch;
#pragma unroll
      for (int od = 0; od < DT; od++) {
        const int kd = z - (d0 + od) + PAD;
        if (kd < 0 || kd >= KS || d0 + od >= g.D || (g.dbg & 2)) continue;        // block-uniform
        if constexpr (KS == 7) {
          // 25 tap-pair slots in linear tap order (see cc16_slots).  Seven A-fragment TYPES per slab row rho and M tile:
          //   E_p (p = 0..2): taps (kh, 2p) | (kh, 2p + 1) for the even kh -> positions v + 2p | v + 2p + 1 of row rho,
          //   O_p (p = 0..2): taps (kh, 2p + 1) | (kh, 2p + 2) for the odd kh -> positions v + 2p + 1 | v + 2p + 2 of row rho,
          //   S = "E_3":      taps (kh, 6) | (kh + 1, 0) for kh = 0, 2, 4 and the lone (6, 6) | zero -> position v + 6 of row rho | v of row rho + 1.
          // Every fragment feeds the MFMAs of all its kh whose output row rho - kh lies in the tile (up to 4 for E / S, 3 for O).
          const T* wbase = wq + ((int64_t)(kd * g.NCH + wch) * 25) * WT + lane_off;
          Frag8<T> bE[4], bO[3];
          auto load_bE = [&](int p) {          // slots of taps (kh, 2p) for kh = 0, 2, 4, 6: j = (7 kh + 2p) / 2 = 7 i + p  (p = 3: the straddling slots)
#pragma unroll
            for (int i = 0; i < 4; i++) bE[i] = frag_ld_lds(wbase + (7 * i + p) * WT);
          };
          auto load_bO = [&](int p) {          // slots of taps (kh, 2p + 1) for kh = 1, 3, 5: j = (7 kh + 2p + 1) / 2 = 7 i + 4 + p
#pragma unroll
            for (int i = 0; i < 3; i++) bO[i] = frag_ld_lds(wbase + (7 * i + 4 + p) * WT);
          };
          const int v_base = wv * 32 + r;
          // Steps k = (slab row, M tile) in row order; the rows that feed no MFMA of a type (rows 0 and 13 for the odd kh) are not read.
          // The fragments run through a ring of four register sets, three reads in flight.
          constexpr int NE = 28, NO = 24;
          constexpr int ordE[NE] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
          constexpr int mtE[NE] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1};
          constexpr int ordO[NO] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12};
          constexpr int mtO[NO] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1};
          Frag8<T> fa[4];
          // the lane's fragment of type (odd, p) at slab row rho, M tile mt
          auto a_ptr = [&](int odd, int p, int rho, int mt) -> const T* {
            int pos = v_base + 2 * p + odd + (q >> 1), row = rho;
            // straddling type: the right-hand tap lives at the start of the next slab row (the last slab row only serves the lone
            // (6, 6) tap, whose right half meets zero weights: it keeps the in-row neighbour so that no lane reads past the slab)
            if (!odd && p == 3 && (q >> 1) && rho + 1 < ROWS) { pos = v_base; row = rho + 1; }
            const int hs = SWZ ? (hsel ^ ((pos >> 3) & 1) ^ ((row & 1) ? lp_par : 0)) : hsel;
            return slab + (pos + row * LP + mt * 16) * CK + hs * 8;
          };
          auto a_step = [&](int odd, int p, int k) -> const T* { return odd ? a_ptr(1, p, ordO[k], mtO[k]) : a_ptr(0, p, ordE[k], mtE[k]); };
          auto do_type = [&](int odd, int p, int nodd, int np, bool pre_next) {
            const int S = odd ? NO : NE;
#pragma unroll
            for (int k = 0; k < NE; k++) {
              if (k >= S) continue;
              const int nx = k + 3;
              if (nx < S) fa[nx % 4] = frag_ld_lds(a_step(odd, p, nx));
              else if (pre_next) fa[nx % 4] = frag_ld_lds(a_step(nodd, np, nx - S));
              __builtin_amdgcn_sched_barrier(0);
              const int rho = odd ? ordO[k] : ordE[k], mt = odd ? mtO[k] : mtE[k];
              if (odd) {
#pragma unroll
                for (int i = 2; i >= 0; i--) {
                  const int orow = rho - (2 * i + 1);
                  if (orow < 0 || orow >= RWO) continue;                 // compile-time after unrolling
                  acc[od][orow][mt] = mma16(fa[k % 4], bO[i], acc[od][orow][mt]);
                }
              } else {
#pragma unroll
                for (int i = 3; i >= 0; i--) {
                  const int orow = rho - 2 * i;
                  if (orow < 0 || orow >= RWO) continue;
                  acc[od][orow][mt] = mma16(fa[k % 4], bE[i], acc[od][orow][mt]);
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          static_assert(NE % 4 == 0 && NO % 4 == 0, "the fragment ring returns to slot 0 after every type");
          // type order E_0 O_0 E_1 O_1 E_2 O_2 E_3: the tiles of the next type are requested while the current one is swept
          load_bE(0);
          fa[0] = frag_ld_lds(a_step(0, 0, 0)); fa[1] = frag_ld_lds(a_step(0, 0, 1)); fa[2] = frag_ld_lds(a_step(0, 0, 2));
#pragma unroll 1
          for (int p = 0; p < 3; p++) {
            load_bO(p); __builtin_amdgcn_sched_barrier(0);
            do_type(0, p, 1, p, true);
            load_bE(p + 1); __builtin_amdgcn_sched_barrier(0);
            do_type(1, p, 0, p + 1, true);
          }
          do_type(0, 3, 0, 0, false);
        } else {
        const T* wbase = wq + ((int64_t)(kd * g.NCH + wch) * KWP) * KS * WT + lane_off;
        Frag8<T> b0[KS], b1[KS];
        auto load_b = [&](int kwp, Frag8<T>* bb) {
#pragma unroll
          for (int kh = 0; kh < KS; kh++) bb[kh] = frag_ld_lds(wbase + (kwp * KS + kh) * WT);
        };
        // A fragments pipelined through three register sets (two reads in flight while a row's MFMAs issue); step s = 2 rho + mt
        Frag8<T> fa[3];
        auto a_ptr = [&](int kwp, int s) -> const T* {
          const int rho = s >> 1, mt = s & 1;
          const int vk = v_lane + 2 * kwp;
          const int hs = SWZ ? (hsel ^ ((vk >> 3) & 1) ^ ((rho & 1) ? lp_par : 0)) : hsel;
          return slab + (vk + rho * LP + mt * 16) * CK + hs * 8;
        };
        auto do_pair = [&](int kwp, const Frag8<T>* bb, bool pre_next) {
          constexpr int S = 2 * ROWS;
#pragma unroll
          for (int s = 0; s < S; s++) {
            const int nx = s + 2;
            if (nx < S) fa[nx % 3] = frag_ld_lds(a_ptr(kwp, nx));
            else if (pre_next) fa[nx % 3] = frag_ld_lds(a_ptr(kwp + 1, nx - S));
            __builtin_amdgcn_sched_barrier(0);
            const int rho = s >> 1, mt = s & 1;
#pragma unroll
            for (int kh = 0; kh < KS; kh++) {
              const int orow = rho - kh;
              if (orow < 0 || orow >= RWO) continue;                 // compile-time after unrolling
              acc[od][orow][mt] = mma16(fa[s % 3], bb[kh], acc[od][orow][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (S % 3 != 0) { const Frag8<T> t0 = fa[S % 3], t1 = fa[(S + 1) % 3]; fa[0] = t0; fa[1] = t1; }
        };
        load_b(0, b0);
        fa[0] = frag_ld_lds(a_ptr(0, 0)); fa[1] = frag_ld_lds(a_ptr(0, 1));
        static_assert(KWP % 2 == 0, "tap pairs are swept two at a time");
#pragma unroll 1
        for (int kwp = 0; kwp < KWP; kwp += 2) {
          load_b(kwp + 1, b1); __builtin_amdgcn_sched_barrier(0);
          do_pair(kwp, b0, true);
          if (kwp + 2 < KWP) { load_b(kwp + 2, b0); __builtin_amdgcn_sched_barrier(0); }
          do_pair(kwp + 1, b1, kwp + 2 < KWP);
        }
        }
      }
      }
    }
  }

  // epilogue.  C/D layout of the 16x16 MFMA: column (output channel) = lane & 15, row (position) = 4 * (lane >> 4) + e.
  // Each wave transposes one output row (32 positions x 16 channels) through a private LDS patch and writes 16-byte chunks.
  constexpr int NC = 16;
  constexpr int EPC = 16 / (int)sizeof(TO), CPP = NC / EPC, PASSES = 32 * CPP / 64;
  if (g.dbg & 4) return;
  __syncthreads();                                                // every wave is done reading the slab
  TO* patch = (TO*)smem_raw + wv * (2 * 32 * NC);
  const float bv = (bias && r < g.Cout) ? bias[r] : 0.f;
  const int wbase_o = w0 + wv * 32;
  TO* y2 = (TO*)g.y2;
  if (!g.wide) {                                                  // unaligned output rows: plain 2/4-byte stores (rare shapes)
#pragma unroll
    for (int od = 0; od < DT; od++)
#pragma unroll
      for (int t = 0; t < RWO; t++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int d = d0 + od, oh = h0 + t, ow = wbase_o + mt * 16 + 4 * q + e;
            if (d < g.D && oh < g.H && ow < g.W && r < g.Cout) {
              const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
              TO* dst = (y2 && r >= g.osplit) ? y2 + vox * g.ldy2 + (r - g.osplit) : y + vox * g.ldy + r;
              st_f(dst, acc[od][t][mt][e] + bv);
            }
          }
    return;
  }
  float st1 = 0.f, st2 = 0.f;
#pragma unroll
  for (int od = 0; od < DT; od++) {
    const int d = d0 + od;
    if (d >= g.D) continue;
    if (g.stat_part) { st1 = 0.f; st2 = 0.f; }
#pragma unroll
    for (int t = 0; t < RWO; t++) {
      TO* pp = patch + (t & 1) * (32 * NC);
      const int oh = h0 + t;
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int m = mt * 16 + 4 * q + e;
          const float v = acc[od][t][mt][e] + bv;
          st_f(pp + m * NC + r, v);
          const float vs = (oh < g.H && wbase_o + m < g.W) ? as_stored<TO>(v) : 0.f;
          st1 += vs; st2 += vs * vs;
        }
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < PASSES; ps++) {
        const int qq = ps * 64 + lane, m = qq / CPP, cc = (qq % CPP) * EPC;
        const int ow = wbase_o + m;
        if (oh < g.H && ow < g.W && cc < g.Cout) {
          const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
          TO* dst = (y2 && cc >= g.osplit) ? y2 + vox * g.ldy2 + (cc - g.osplit) : y + vox * g.ldy + cc;
          if (cc + EPC <= g.Cout) *(v4u*)dst = *(const v4u*)(pp + m * NC + cc);
          else for (int k = 0; k < EPC; k++) if (cc + k < g.Cout) dst[k] = pp[m * NC + cc + k];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (g.stat_part) {          // block-uniform: per-channel (sum, sum of squares) of this block's voxels of depth slice d
      float a1 = st1 + __shfl_xor(st1, 16, 64), a2 = st2 + __shfl_xor(st2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      __syncthreads();
      float* sred = (float*)(smem_raw + 4 * 2 * 32 * NC * sizeof(TO));      // behind the four waves' patches
      if (lane < 16) { sred[(wv * 2) * 16 + lane] = a1; sred[(wv * 2 + 1) * 16 + lane] = a2; }
      __syncthreads();
      if (tid < 32) {
        const int which = tid >> 4, c = tid & 15;
        if (c < g.Cout) {
          const int blk = (d * g.tiles_h + th) * g.tiles_w + tw;
          g.stat_part[(((int64_t)n * g.stat_nblk + blk) * 2 + which) * g.Cout + c] =
              (sred[(0 + which) * 16 + c] + sred[(2 + which) * 16 + c]) + (sred[(4 + which) * 16 + c] + sred[(6 + which) * 16 + c]);
        }
      }
      __syncthreads();
    }
  }
}


// ================================================================================================ row lengths other than 128 (round 5)
// The kernel above gives each of its four waves 32 positions of a 128-position tile: at W = 96 (the segmentation network's own crop,
// OARSegmentation/config.py:24, and the cascade's sliding-window roi, train_light_linked_model.py:152-154) one wave in four multiplies
// padding, at W = 192 (BASELINE configs[4]) half of every second tile does: 15-20 % of the 7x7x7 rate (tools/bench_conv.py --widths).
// k_conv_cc16w tiles W in units of 2 x NMT M tiles (NMT = 3: 96 positions) and splits the block's 8 output rows instead: wave = (column half
// wcol, row half wrow) owns 4 rows x NMT M tiles of BOTH depth slices (2 x 4 x NMT x 4 accumulator registers: 96 for NMT = 3).  Every wave
// works on every staged slab, whichever of the two depth slices the slab serves (no idle waves at the ends of the kd range), and an A
// fragment now feeds the MFMAs of both depth slices (their kd differ by one: same slab, other weights): 1 ds_read_b128 per <= 4 MFMAs
// with a 4-row window where the 8-row window above needs one per <= 4 for ONE slice.  Same slab layout / swizzle / staging / packed
// weights / transposing epilogue / statistics as k_conv_cc16.
template <typename T, int KS, int NMT, int OCC, typename TO, int DT = 2>
__global__ void __launch_bounds__(256, OCC) k_conv_cc16w(const T* __restrict__ x, const T* __restrict__ wq, const float* __restrict__ bias,
                                                    TO* __restrict__ y, Cc16Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* slab = (T*)smem_raw;
  constexpr int PAD = KS / 2, KWP = (KS + 1) / 2, RWO = 8, RW = 4, ROWS = RWO + KS - 1, WROWS = RW + KS - 1, CK = 16, WPW = NMT * 16, TW = 2 * WPW;
  static_assert(DT == 1 || DT == 2, "one or two depth slices per block");
  constexpr int LP = (TW + KS - 1 + 7) & ~7, LR = ROWS;
  constexpr bool SWZ = sizeof(T) == 2;
  constexpr int lp_par = SWZ ? ((LP >> 3) & 1) : 0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wcol = wv & 1, wrow = wv >> 1;
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; b /= g.tiles_h; const int dt = b % g.dtiles; const int n = b / g.dtiles;
  const int h0 = th * RWO, w0 = tw * TW, d0 = dt * DT;
  v4f acc[DT][RW][NMT];
#pragma unroll
  for (int a = 0; a < DT; a++)
#pragma unroll
    for (int i = 0; i < RW; i++)
#pragma unroll
      for (int m = 0; m < NMT; m++) acc[a][i][m] = (v4f){0.f, 0.f, 0.f, 0.f};

  constexpr int pieces = LR * LP * 2;
  constexpr int SU = 8;
  const int cin_in = g.x3 ? 32 * g.x3 : g.Cin;
  const bool fast = SWZ && ((g.x3 ? 2 * g.x3 : g.NCH) * 16 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx)) && (g.ldx % 8 == 0) && (((uintptr_t)x & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 16 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0))) &&
                    (int64_t)g.H * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30) &&
                    (int64_t)LR * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30);
  const int st_half = tid & 1, st_lp0 = (tid >> 1) % LP, st_lr0 = (tid >> 1) / LP;
  const int hsel = q & 1;
  const int lane_off = r * 32 + q * 8;
  constexpr int WT = 512;
  const int row0 = wrow * RW;                       // first slab row of the wave's window

  for (int z = d0 - PAD; z < d0 + DT + PAD; z++) {
    if (z < 0 || z >= g.D) continue;
    const int nstage = g.x3 ? 2 * g.x3 : g.NCH;
    const int kd0 = z - d0 + PAD, kd1 = kd0 - 1;
    const bool has0 = kd0 >= 0 && kd0 < KS && d0 < g.D && !(g.dbg & 2), has1 = DT == 2 && kd1 >= 0 && kd1 < KS && d0 + 1 < g.D && !(g.dbg & 2);
    for (int ch = 0; ch < nstage; ch++) {
      lds_barrier();
      if (g.dbg & 1) {
      }
#if DP_CC16_BUF
      else if (fast) {
        // buffer loads with block-constant offsets, as in k_conv_cc16: the slab is 2 LP <= 256 pieces wide, so thread tid < 2 LP owns
        // column tid / 2 (half tid & 1) in every row
        const bool second = g.x2 && ch * CK >= g.csplit;
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + z) * g.H) * (int64_t)g.W * ldsrc + c0;
        const uint64_t pa = (uint64_t)(uintptr_t)xplane;
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa), phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const unsigned pbytes = __builtin_amdgcn_readfirstlane((unsigned)(g.H * g.W * ldsrc * 2 - c0 * 2));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)phi << 32) | plo), 0, (int)pbytes, 0x00020000);
        const unsigned rowb = (unsigned)(g.W * ldsrc * 2);
        static_assert(2 * LP <= 256 && ((LP >> 3) & 1) == 1, "one column per thread; row-parity form of the slab swizzle");
        if (tid < 2 * LP) {
          const int t2 = tid >> 1, iw = w0 - PAD + t2;
          const unsigned vb = ((unsigned)iw < (unsigned)g.W) ? (unsigned)((((h0 - PAD) * g.W + iw) * ldsrc + st_half * 8) * 2) : 0x80000000u;
          const int par0 = (t2 >> 3) & 1;
          T* const dA = slab + t2 * CK + ((st_half ^ par0) * 8);
          T* const dB = slab + t2 * CK + ((st_half ^ par0 ^ 1) * 8);
          v4u bufa[LR];
#pragma unroll
          for (int rr = 0; rr < LR; rr++) bufa[rr] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vb + (unsigned)rr * rowb, 0, 0));
#pragma unroll
          for (int rr = 0; rr < LR; rr++) *(v4u*)(((rr & 1) ? dB : dA) + rr * LP * CK) = bufa[rr];
        }
      }
#else
      else if (fast) {
        const bool second = g.x2 && ch * CK >= g.csplit;
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
        const T* xplane = xsrc + (((int64_t)n * g.D + z) * g.H) * (int64_t)g.W * ldsrc + c0 + st_half * 8;
        int lp = st_lp0, lr = st_lr0, v = tid >> 1;
        for (int p0 = 0; p0 < pieces; p0 += 256 * SU) {
          v4u buf[SU]; int vv[SU];
#pragma unroll
          for (int j = 0; j < SU; j++) {
            const int ih = h0 - PAD + lr, iw = w0 - PAD + lp;
            const bool ok = (p0 + j * 256 + tid < pieces) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            v4u t = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
            buf[j] = ok ? t : (v4u){0, 0, 0, 0};
            vv[j] = (p0 + j * 256 + tid < pieces) ? v : -1;
            v += 128; lp += 128;
#pragma unroll
            for (int c_ = 0; c_ < (128 + LP - 1) / LP; c_++) if (lp >= LP) { lp -= LP; lr++; }
          }
#pragma unroll
          for (int j = 0; j < SU; j++)
            if (vv[j] >= 0) *(v4u*)(slab + (int64_t)vv[j] * CK + (st_half ^ ((vv[j] >> 3) & 1)) * 8) = buf[j];
        }
      }
#endif
      else {
        for (int p = tid; p < pieces; p += 256) {
          int half = p & 1, v = p >> 1, lp = v % LP, lr = v / LP;
          int ih = h0 - PAD + lr, iw = w0 - PAD + lp, c = ch * CK + half * 8;
          int nv = cin_in - c; nv = nv > 8 ? 8 : nv;
          bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
          const bool second = g.x2 && c >= g.csplit;
          if (!second && g.x2 && c + nv > g.csplit) nv = g.csplit - c;
          const T* xsrc = second ? (const T*)g.x2 : x;
          const int ldsrc = second ? g.ldx2 : g.ldx, cc = c - (second ? g.csplit : 0);
          Frag8<T> f = ok ? frag_load(xsrc + ((((int64_t)n * g.D + z) * g.H + ih) * g.W + iw) * ldsrc + cc, nv) : frag_zero<T>();
          frag_st_lds(slab + (int64_t)v * CK + (half ^ (SWZ ? ((v >> 3) & 1) : 0)) * 8, f);
        }
      }
      lds_barrier();
      const int nrep = (g.x3 && ch < g.x3) ? 2 : 1;
#pragma unroll 1
      for (int rep = 0; rep < nrep; rep++) {
        const int wch = rep ? ch + 2 * g.x3 : ch;
        // one sweep of the staged slab for the depth slices it serves (H0 / H1: slice d0 / d0 + 1; block-uniform)
        auto sweep = [&]<bool H0, bool H1>() {
          if constexpr (KS == 7) {
            const T* wb0 = wq + ((int64_t)(kd0 * g.NCH + wch) * 25) * WT + lane_off;
            const T* wb1 = wq + ((int64_t)(kd1 * g.NCH + wch) * 25) * WT + lane_off;
            Frag8<T> bE[2][4], bO[2][3];
            auto load_bE = [&](int p) {
#pragma unroll
              for (int i = 0; i < 4; i++) { if constexpr (H0) bE[0][i] = frag_ld_lds(wb0 + (7 * i + p) * WT); if constexpr (H1) bE[1][i] = frag_ld_lds(wb1 + (7 * i + p) * WT); }
            };
            auto load_bO = [&](int p) {
#pragma unroll
              for (int i = 0; i < 3; i++) { if constexpr (H0) bO[0][i] = frag_ld_lds(wb0 + (7 * i + 4 + p) * WT); if constexpr (H1) bO[1][i] = frag_ld_lds(wb1 + (7 * i + 4 + p) * WT); }
            };
            const int v_base = wcol * WPW + r;
            // steps k = (window row, M tile): even kh use window rows 0 .. WROWS - 1, odd kh rows 1 .. WROWS - 2
            constexpr int NE = WROWS * NMT, NO = (WROWS - 2) * NMT, RING = (NMT % 2) ? 6 : 4;
            static_assert(NE % RING == 0 && NO % RING == 0, "the fragment ring returns to slot 0 after every type");
            Frag8<T> fa[RING];
            auto a_ptr = [&](int odd, int p, int rw_, int mt) -> const T* {
              int pos = v_base + 2 * p + odd + (q >> 1), row = row0 + rw_;
              if (!odd && p == 3 && (q >> 1) && row + 1 < ROWS) { pos = v_base; row = row + 1; }     // straddling slot: right tap = start of the next slab row
              const int hs = SWZ ? (hsel ^ ((pos >> 3) & 1) ^ ((row & 1) ? lp_par : 0)) : hsel;
              return slab + (pos + row * LP + mt * 16) * CK + hs * 8;
            };
            auto a_step = [&](int odd, int p, int k) -> const T* { return odd ? a_ptr(1, p, 1 + k / NMT, k % NMT) : a_ptr(0, p, k / NMT, k % NMT); };
            auto do_type = [&](int odd, int p, int nodd, int np, bool pre_next) {
              const int S = odd ? NO : NE;
#pragma unroll
              for (int k = 0; k < NE; k++) {
                if (k >= S) continue;
                const int nx = k + 3;
                if (nx < S) fa[nx % RING] = frag_ld_lds(a_step(odd, p, nx));
                else if (pre_next) fa[nx % RING] = frag_ld_lds(a_step(nodd, np, nx - S));
                __builtin_amdgcn_sched_barrier(0);
                const int rw_ = odd ? 1 + k / NMT : k / NMT, mt = k % NMT;
                if (odd) {
#pragma unroll
                  for (int i = 2; i >= 0; i--) {
                    const int orow = rw_ - (2 * i + 1);
                    if (orow < 0 || orow >= RW) continue;
                    if constexpr (H0) acc[0][orow][mt] = mma16(fa[k % RING], bO[0][i], acc[0][orow][mt]);
                    if constexpr (H1) acc[1][orow][mt] = mma16(fa[k % RING], bO[1][i], acc[1][orow][mt]);
                  }
                } else {
#pragma unroll
                  for (int i = 3; i >= 0; i--) {
                    const int orow = rw_ - 2 * i;
                    if (orow < 0 || orow >= RW) continue;
                    if constexpr (H0) acc[0][orow][mt] = mma16(fa[k % RING], bE[0][i], acc[0][orow][mt]);
                    if constexpr (H1) acc[1][orow][mt] = mma16(fa[k % RING], bE[1][i], acc[1][orow][mt]);
                  }
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            };
            load_bE(0);
            fa[0] = frag_ld_lds(a_step(0, 0, 0)); fa[1] = frag_ld_lds(a_step(0, 0, 1)); fa[2] = frag_ld_lds(a_step(0, 0, 2));
#pragma unroll 1
            for (int p = 0; p < 3; p++) {
              load_bO(p); __builtin_amdgcn_sched_barrier(0);
              do_type(0, p, 1, p, true);
              load_bE(p + 1); __builtin_amdgcn_sched_barrier(0);
              do_type(1, p, 0, p + 1, true);
            }
            do_type(0, 3, 0, 0, false);
          } else {
            const T* wb0 = wq + ((int64_t)(kd0 * g.NCH + wch) * KWP) * KS * WT + lane_off;
            const T* wb1 = wq + ((int64_t)(kd1 * g.NCH + wch) * KWP) * KS * WT + lane_off;
            Frag8<T> b0[2][KS], b1[2][KS];
            auto load_b = [&](int kwp, Frag8<T> (*bb)[KS]) {
#pragma unroll
              for (int kh = 0; kh < KS; kh++) { if constexpr (H0) bb[0][kh] = frag_ld_lds(wb0 + (kwp * KS + kh) * WT); if constexpr (H1) bb[1][kh] = frag_ld_lds(wb1 + (kwp * KS + kh) * WT); }
            };
            constexpr int S = WROWS * NMT;
            static_assert(S % 3 == 0, "three fragment registers sets");
            Frag8<T> fa[3];
            const int v_lane = wcol * WPW + r + (q >> 1);
            auto a_ptr = [&](int kwp, int s_) -> const T* {
              const int rw_ = s_ / NMT, mt = s_ % NMT, row = row0 + rw_;
              const int vk = v_lane + 2 * kwp;
              const int hs = SWZ ? (hsel ^ ((vk >> 3) & 1) ^ ((row & 1) ? lp_par : 0)) : hsel;
              return slab + (vk + row * LP + mt * 16) * CK + hs * 8;
            };
            auto do_pair = [&](int kwp, Frag8<T> (*bb)[KS], bool pre_next) {
#pragma unroll
              for (int s_ = 0; s_ < S; s_++) {
                const int nx = s_ + 2;
                if (nx < S) fa[nx % 3] = frag_ld_lds(a_ptr(kwp, nx));
                else if (pre_next) fa[nx % 3] = frag_ld_lds(a_ptr(kwp + 1, nx - S));
                __builtin_amdgcn_sched_barrier(0);
                const int rw_ = s_ / NMT, mt = s_ % NMT;
#pragma unroll
                for (int kh = 0; kh < KS; kh++) {
                  const int orow = rw_ - kh;
                  if (orow < 0 || orow >= RW) continue;
                  if constexpr (H0) acc[0][orow][mt] = mma16(fa[s_ % 3], bb[0][kh], acc[0][orow][mt]);
                  if constexpr (H1) acc[1][orow][mt] = mma16(fa[s_ % 3], bb[1][kh], acc[1][orow][mt]);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            };
            load_b(0, b0);
            fa[0] = frag_ld_lds(a_ptr(0, 0)); fa[1] = frag_ld_lds(a_ptr(0, 1));
            static_assert(KWP % 2 == 0, "tap pairs are swept two at a time");
#pragma unroll 1
            for (int kwp = 0; kwp < KWP; kwp += 2) {
              load_b(kwp + 1, b1); __builtin_amdgcn_sched_barrier(0);
              do_pair(kwp, b0, true);
              if (kwp + 2 < KWP) { load_b(kwp + 2, b0); __builtin_amdgcn_sched_barrier(0); }
              do_pair(kwp + 1, b1, kwp + 2 < KWP);
            }
          }
        };
        if constexpr (DT == 2) {
          if (has0 && has1) sweep.template operator()<true, true>();
          else if (has0) sweep.template operator()<true, false>();
          else if (has1) sweep.template operator()<false, true>();
        } else {
          if (has0) sweep.template operator()<true, false>();
        }
      }
    }
  }

  // epilogue: each wave transposes one of its output rows (WPW positions x 16 channels) through a private LDS patch, 16-byte stores
  constexpr int NC = 16;
  constexpr int EPC = 16 / (int)sizeof(TO), CPP = NC / EPC, ITEMS = WPW * CPP, PASSES = (ITEMS + 63) / 64;
  if (g.dbg & 4) return;
  __syncthreads();
  TO* patch = (TO*)smem_raw + wv * (2 * WPW * NC);
  const float bv = (bias && r < g.Cout) ? bias[r] : 0.f;
  const int wbase_o = w0 + wcol * WPW;
  TO* y2 = (TO*)g.y2;
  if (!g.wide) {
#pragma unroll
    for (int od = 0; od < DT; od++)
#pragma unroll
      for (int t = 0; t < RW; t++)
#pragma unroll
        for (int mt = 0; mt < NMT; mt++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int d = d0 + od, oh = h0 + row0 + t, ow = wbase_o + mt * 16 + 4 * q + e;
            if (d < g.D && oh < g.H && ow < g.W && r < g.Cout) {
              const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
              TO* dst = (y2 && r >= g.osplit) ? y2 + vox * g.ldy2 + (r - g.osplit) : y + vox * g.ldy + r;
              st_f(dst, acc[od][t][mt][e] + bv);
            }
          }
    return;
  }
  float st1 = 0.f, st2 = 0.f;
#pragma unroll
  for (int od = 0; od < DT; od++) {
    const int d = d0 + od;
    if (d >= g.D) continue;
    if (g.stat_part) { st1 = 0.f; st2 = 0.f; }
#pragma unroll
    for (int t = 0; t < RW; t++) {
      TO* pp = patch + (t & 1) * (WPW * NC);
      const int oh = h0 + row0 + t;
#pragma unroll
      for (int mt = 0; mt < NMT; mt++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int m = mt * 16 + 4 * q + e;
          const float v = acc[od][t][mt][e] + bv;
          st_f(pp + m * NC + r, v);
          const float vs = (oh < g.H && wbase_o + m < g.W) ? as_stored<TO>(v) : 0.f;
          st1 += vs; st2 += vs * vs;
        }
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < PASSES; ps++) {
        const int qq = ps * 64 + lane, m = qq / CPP, cc = (qq % CPP) * EPC;
        const int ow = wbase_o + m;
        if (qq < ITEMS && oh < g.H && ow < g.W && cc < g.Cout) {
          const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
          TO* dst = (y2 && cc >= g.osplit) ? y2 + vox * g.ldy2 + (cc - g.osplit) : y + vox * g.ldy + cc;
          if (cc + EPC <= g.Cout) *(v4u*)dst = *(const v4u*)(pp + m * NC + cc);
          else for (int k = 0; k < EPC; k++) if (cc + k < g.Cout) dst[k] = pp[m * NC + cc + k];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (g.stat_part) {
      float a1 = st1 + __shfl_xor(st1, 16, 64), a2 = st2 + __shfl_xor(st2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      __syncthreads();
      float* sred = (float*)(smem_raw + 4 * 2 * WPW * NC * sizeof(TO));
      if (lane < 16) { sred[(wv * 2) * 16 + lane] = a1; sred[(wv * 2 + 1) * 16 + lane] = a2; }
      __syncthreads();
      if (tid < 32) {
        const int which = tid >> 4, c = tid & 15;
        if (c < g.Cout) {
          const int blk = (d * g.tiles_h + th) * g.tiles_w + tw;
          g.stat_part[(((int64_t)n * g.stat_nblk + blk) * 2 + which) * g.Cout + c] =
              (sred[(0 + which) * 16 + c] + sred[(2 + which) * 16 + c]) + (sred[(4 + which) * 16 + c] + sred[(6 + which) * 16 + c]);
        }
      }
      __syncthreads();
    }
  }
}

// tile width of the launch: 96-position tiles (k_conv_cc16w) when they cover the row with less padding than 128-position tiles
// (W = 96, 192, 160, 288 ...; ties go to the 128-position kernel).  A function of (W, k, dtype) alone: dp_conv3d_tiled_stat_blocks must
// agree.  Exact fp32 always runs the 128-position kernel (its fragments are twice as wide): its grid and statistics rows are those of
// 128-position tiles too, not 96-position ones with every third tile empty.
// 7x7x7 only: the 3x3x3 launches are latency-bound and keep the three-blocks-per-CU kernel (measured at 4 x 96^3, 16 -> 16: 0.087 ms
// there, 0.109 ms on the 96-position tiles at two blocks per CU; 7x7x7 16 -> 16: 0.597 -> 0.508 ms, 32 -> 16: 1.014 -> 0.864).
static inline int cc16_tw(int W, int k, int dtype) {
  static const int off = [] { const char* e = getenv("DP_NO_CC16W"); return (e && atoi(e)) ? 1 : 0; }();
  static const int t3 = [] { const char* e = getenv("DP_CC16W3"); return e ? atoi(e) : 0; }();      // (experiment: the half-size 3x3x3 tile, 64 positions x 8 rows x 1 depth slice per block)
  if (!off && k == 3 && t3 && W % 64 == 0) return 64;
  if (off || k != 7 || dtype == DP_F32) return 128;
  static const int t64 = [] { const char* e = getenv("DP_CC16W_64"); return (e && atoi(e)) ? 1 : 0; }();      // (experiment: 64-position tiles of the same wave geometry for W % 64 == 0)
  if (t64 && W % 64 == 0) return 64;
  return cdiv(W, 96) * 96 < cdiv(W, 128) * 128 ? 96 : 128;
}

// ================================================================================================ 3x3x3 marching along depth (round 5)
// EXPERIMENT, off by default (env DP_CC16M=1): parity-green (every 3x3x3 case of tests/test_ops_gpu.py incl. statistics, virtual concat,
// fp32x3), but 0.138 ms against 0.098 ms for 16 -> 16 at 2 x 128^3 (0.29 against 0.195 for 32 -> 16).
// The 3x3x3 launches of the kernels above are bound by re-staging: a block owns one or two depth slices and stages every input slab it
// needs (3 per slice with DT = 1, 2 with DT = 2), so every input voxel travels L2 -> LDS 2-3 times (x 1.25 for the row halo).
// k_conv_cc16m turns the loop around: a block owns 8 rows x 64 positions and MARCHES over a segment of ML depth slices.  Every input slab
// is staged ONCE per block (ML + 2 slabs for ML slices: 1.4 voxel loads per output voxel instead of 3.8) and is swept input-stationary:
// an A fragment (16 positions of slab row rho, tap pair kwp) feeds the MFMAs of every (kd, kh) it contributes to -- up to nine, into
// the accumulators of the three output slices z - 1, z, z + 1 that are live at any time (3 x 8 rows x 4 registers = 96 per wave: one
// 16-position M tile per wave).  With one input chunk all 18 weight fragments of the 3x3x3 kernel stay in registers for the whole march.
// A slice is finished when the slab behind it has been swept; the wave then transposes it through its private LDS patch (no block
// barrier) and clears the accumulator slot for the slice three further on.  Statistics: one partial row per BLOCK.
// Why it loses: 96 accumulator + 72 weight + 12 fragment-ring + 24 prefetch registers and the addressing come to 304 per lane, i.e. ONE
// block of four waves per CU (at two blocks the compiler spills the weights into the sweep: 0.30 ms).  With one wave per SIMD nothing
// overlaps: the step time is the SUM of its phases -- knock-outs at 2 x 128^3, 34 steps per block: sweep 1.6 us (1.0 of MFMA issue),
// epilogue of the finished slice 1.6 us, exposed staging 0.6 us with the next slab prefetched into registers during the sweep (1.9
// without) -- where k_conv_cc16 overlaps the phases of three resident blocks.  The 4-row variant (DP_CC16M_ROWS=4: 48 accumulator
// registers, 255 in all, two blocks per CU, 1.6 voxel loads per output) closes most of the gap and no more: 0.116 ms (32 -> 16, whose two
// chunks reload the 18 weight fragments per slab: 0.30).
#define CC16M_TW 64
static inline bool cc16m_use(int k, int dtype) {
  static const bool on = [] { const char* e = getenv("DP_CC16M"); return e && atoi(e); }();      // measured slower than k_conv_cc16 (below): opt-in
  return on && k == 3 && (dtype == DP_BF16 || dtype == DP_F16 || dtype == DP_X3 || dtype == DP_X1);
}
static inline int cc16m_rows() {          // output rows per block: 8 (one block per CU), or 4 (48 accumulator registers: two blocks per CU)
  static const int e = [] { const char* v = getenv("DP_CC16M_ROWS"); return (v && atoi(v) == 4) ? 4 : 8; }();
  return e;
}
static inline int cc16m_ml(int D) {
  static const int e = [] { const char* v = getenv("DP_CC16M_ML"); return v ? atoi(v) : 16; }();
  const int ml = e < 1 ? 1 : e;
  return ml > D ? D : ml;
}

// ALL: every tap kd of this slab feeds a live output slice (the steady state of the march): straight-line code.  The first two and the
// last two steps of a segment take the guarded form (block-uniform branches around the MFMA groups).
template <int J, bool ALL, int RWO, typename T>
__device__ __forceinline__ void cc16m_sweep(v4f (&acc)[3][RWO], const Frag8<T> (&B)[3][2][3], const T* __restrict__ slab, int v_lane, int hsel,
                                            bool m0, bool m1, bool m2) {
  constexpr int LP = 72, CK = 16, ROWS = RWO + 2, S = 2 * ROWS;
  Frag8<T> fa[3];
  // four lane addresses (tap pair x row parity: LP / 8 = 9 is odd, so the swizzle bit flips with the row); the row is an immediate offset
  const T* abase[2][2];
#pragma unroll
  for (int kwp = 0; kwp < 2; kwp++)
#pragma unroll
    for (int par = 0; par < 2; par++) {
      const int vk = v_lane + 2 * kwp;
      abase[kwp][par] = slab + vk * CK + (hsel ^ ((vk >> 3) & 1) ^ par) * 8;
    }
  auto a_ptr = [&](int s) -> const T* { const int rho = s >> 1, kwp = s & 1; return abase[kwp][rho & 1] + rho * LP * CK; };
  fa[0] = frag_ld_lds(a_ptr(0)); fa[1] = frag_ld_lds(a_ptr(1));
#pragma unroll
  for (int s = 0; s < S; s++) {
    if (s + 2 < S) fa[(s + 2) % 3] = frag_ld_lds(a_ptr(s + 2));
    __builtin_amdgcn_sched_barrier(0);
    const int rho = s >> 1, kwp = s & 1;
    // input slab z, tap kd -> output slice z + 1 - kd, accumulator slot (J - kd) mod 3  (J = step index mod 3)
    if (ALL || m0) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++) { const int orow = rho - kh; if (orow < 0 || orow >= RWO) continue; acc[J][orow] = mma16(fa[s % 3], B[0][kwp][kh], acc[J][orow]); }
    }
    if (ALL || m1) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++) { const int orow = rho - kh; if (orow < 0 || orow >= RWO) continue; acc[(J + 2) % 3][orow] = mma16(fa[s % 3], B[1][kwp][kh], acc[(J + 2) % 3][orow]); }
    }
    if (ALL || m2) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++) { const int orow = rho - kh; if (orow < 0 || orow >= RWO) continue; acc[(J + 1) % 3][orow] = mma16(fa[s % 3], B[2][kwp][kh], acc[(J + 1) % 3][orow]); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// RES: one input chunk (and no DP_X3 operand split): the 18 weight fragments are loaded once and stay in registers for the whole march
template <typename T, typename TO, bool RES, int RWO = 8, int OCC = 1>
__global__ void __launch_bounds__(256, OCC) k_conv_cc16m(const T* __restrict__ x, const T* __restrict__ wq, const float* __restrict__ bias,
                                                       TO* __restrict__ y, Cc16Geom g, int ML) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int KS = 3, PAD = 1, KWP = 2, ROWS = RWO + KS - 1, CK = 16, TW = CC16M_TW, LP = 72, LR = ROWS, NC = 16, WT = 512;
  static_assert(LP == ((TW + KS - 1 + 7) & ~7), "slab pitch");
  constexpr int SLAB_BYTES = LR * LP * CK * (int)sizeof(T);
  T* slab = (T*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  TO* patch = (TO*)(smem_raw + SLAB_BYTES) + wv * (32 * NC);
  float* sred = (float*)(smem_raw + SLAB_BYTES + 4 * 32 * NC * sizeof(TO));
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);          // XCD-aware order (see k_conv_cc16)
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; b /= g.tiles_h; const int seg = b % g.dtiles; const int n = b / g.dtiles;
  const int h0 = th * RWO, w0 = tw * TW, d0 = seg * ML, MLe = min(ML, g.D - d0);
  v4f acc[3][RWO];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < RWO; i++) acc[a][i] = (v4f){0.f, 0.f, 0.f, 0.f};

  constexpr int pieces = LR * LP * 2;
  const int cin_in = g.x3 ? 32 * g.x3 : g.Cin;
  const int nstage = g.x3 ? 2 * g.x3 : g.NCH;
  const bool fast = (nstage * 16 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx)) && (g.ldx % 8 == 0) && (((uintptr_t)x & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 16 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0))) &&
                    (int64_t)g.H * g.W * max(g.ldx, g.x2 ? g.ldx2 : 0) < (1ll << 30);
  const int st_half = tid & 1, st_lp0 = (tid >> 1) % LP, st_lr0 = (tid >> 1) / LP;
  const int v_lane = wv * 16 + r + (q >> 1), hsel = q & 1;
  const int lane_off = r * 32 + q * 8;
  Frag8<T> B[3][2][3];
  auto load_B = [&](int wch) {
#pragma unroll
    for (int kd = 0; kd < 3; kd++) {
      const T* wbase = wq + ((int64_t)(kd * g.NCH + wch) * KWP) * KS * WT + lane_off;
#pragma unroll
      for (int kwp = 0; kwp < 2; kwp++)
#pragma unroll
        for (int kh = 0; kh < 3; kh++) B[kd][kwp][kh] = frag_ld_lds(wbase + (kwp * KS + kh) * WT);
    }
  };
  constexpr bool resident = RES;
  if (resident) {          // (complete BEFORE the march: a pending weight load would make every sweep wait for vmcnt(0), i.e. for the prefetch too)
    load_B(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float bv = (bias && r < g.Cout) ? bias[r] : 0.f;
  const int wbase_o = w0 + wv * 16;
  TO* y2 = (TO*)g.y2;
  constexpr int EPC = 16 / (int)sizeof(TO), CPP = NC / EPC, PASSES = 32 * CPP / 64;
  float st1 = 0.f, st2 = 0.f;

  // finished slice in accumulator slot SL -> memory (two output rows per trip through the wave's patch), then the slot is cleared
  auto finish = [&](auto slot_c, int d) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int t = 0; t < RWO; t += 2) {
#pragma unroll
      for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int m = tt * 16 + 4 * q + e;
          const float v = acc[SL][t + tt][e] + bv;
          st_f(patch + m * NC + r, v);
          const float vs = (h0 + t + tt < g.H && wbase_o + 4 * q + e < g.W) ? as_stored<TO>(v) : 0.f;
          st1 += vs; st2 += vs * vs;
        }
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < PASSES; ps++) {
        const int qq = ps * 64 + lane, m = qq / CPP, cc = (qq % CPP) * EPC;
        const int oh = h0 + t + (m >> 4), ow = wbase_o + (m & 15);
        if (oh < g.H && ow < g.W && cc < g.Cout) {
          const int64_t vox = (((int64_t)n * g.D + d) * g.H + oh) * g.W + ow;
          TO* dst = (y2 && cc >= g.osplit) ? y2 + vox * g.ldy2 + (cc - g.osplit) : y + vox * g.ldy + cc;
          if (cc + EPC <= g.Cout) *(v4u*)dst = *(const v4u*)(patch + m * NC + cc);
          else for (int k = 0; k < EPC; k++) if (cc + k < g.Cout) dst[k] = patch[m * NC + cc + k];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < RWO; i++) acc[SL][i] = (v4f){0.f, 0.f, 0.f, 0.f};
  };

  // Staging in two halves with the sweep in between: the global loads of the NEXT slab are issued (into registers) right after the
  // barrier that publishes the current one, travel while the current slab is swept, and are written to LDS after the next barrier --
  // with one block of four waves per CU (304 registers per lane) nothing else hides that latency.
  constexpr int SUP = (pieces + 255) / 256;          // 16-byte pieces per thread
  v4u pbuf[SUP];
  auto stage_load = [&](int z, int ch) {
    if (!fast) return;                               // (guarded shapes: staged directly in stage_store)
    const bool second = g.x2 && ch * CK >= g.csplit;
    const T* xsrc = second ? (const T*)g.x2 : x;
    const int ldsrc = second ? g.ldx2 : g.ldx, c0 = ch * CK - (second ? g.csplit : 0);
    const T* xplane = xsrc + (((int64_t)n * g.D + z) * g.H) * (int64_t)g.W * ldsrc + c0 + st_half * 8;
    int lp = st_lp0, lr = st_lr0;
#pragma unroll
    for (int j = 0; j < SUP; j++) {
      const int ih = h0 - PAD + lr, iw = w0 - PAD + lp;
      const bool ok = (j * 256 + tid < pieces) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      v4u t = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
      pbuf[j] = ok ? t : (v4u){0, 0, 0, 0};
      lp += 128;
#pragma unroll
      for (int c_ = 0; c_ < (128 + LP - 1) / LP; c_++) if (lp >= LP) { lp -= LP; lr++; }
    }
  };
  auto stage_store = [&](int z, int ch) {
    if (fast) {
#pragma unroll
      for (int j = 0; j < SUP; j++) {
        const int v = (tid >> 1) + 128 * j;
        if (j * 256 + tid < pieces) *(v4u*)(slab + (int64_t)v * CK + (st_half ^ ((v >> 3) & 1)) * 8) = pbuf[j];
      }
    } else {
      for (int p = tid; p < pieces; p += 256) {
        int half = p & 1, v = p >> 1, lp = v % LP, lr = v / LP;
        int ih = h0 - PAD + lr, iw = w0 - PAD + lp, c = ch * CK + half * 8;
        int nv = cin_in - c; nv = nv > 8 ? 8 : nv;
        bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
        const bool second = g.x2 && c >= g.csplit;
        if (!second && g.x2 && c + nv > g.csplit) nv = g.csplit - c;
        const T* xsrc = second ? (const T*)g.x2 : x;
        const int ldsrc = second ? g.ldx2 : g.ldx, cc = c - (second ? g.csplit : 0);
        Frag8<T> f = ok ? frag_load(xsrc + ((((int64_t)n * g.D + z) * g.H + ih) * g.W + iw) * ldsrc + cc, nv) : frag_zero<T>();
        frag_st_lds(slab + (int64_t)v * CK + (half ^ ((v >> 3) & 1)) * 8, f);
      }
    }
  };

  auto step = [&](auto j_c, int zi) {
    constexpr int J = decltype(j_c)::value;
    const int z = d0 - 1 + zi;
    // tap kd of slab z feeds output slice oi = zi - kd of the segment
    const bool m0 = zi < MLe, m1 = zi >= 1 && zi - 1 < MLe, m2 = zi >= 2 && zi - 2 < MLe;
    if (z >= 0 && z < g.D) {
      for (int ch = 0; ch < nstage; ch++) {
        lds_barrier();                                 // every wave is done with the previous slab
        stage_store(z, ch);
        lds_barrier();
        const int nrep = (g.x3 && ch < g.x3) ? 2 : 1;  // an x_hi slab meets w_hi (chunk ch) and w_lo (chunk 2 x3 + ch)
#pragma unroll 1
        for (int rep = 0; rep < nrep; rep++) {
          if (!resident) { load_B(rep ? ch + 2 * g.x3 : ch); __builtin_amdgcn_sched_barrier(0); }
          if (rep == 0) {                              // (after the weight loads: waiting for those must not wait for the prefetch)
            if (ch + 1 < nstage) stage_load(z, ch + 1);
            else if (zi + 1 < MLe + 2 && z + 1 < g.D) stage_load(z + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (m0 && m1 && m2) cc16m_sweep<J, true, RWO>(acc, B, slab, v_lane, hsel, true, true, true);
          else cc16m_sweep<J, false, RWO>(acc, B, slab, v_lane, hsel, m0, m1, m2);
        }
      }
    }
    if (m2) finish(std::integral_constant<int, (J + 1) % 3>{}, d0 + zi - 2);
  };
  stage_load(d0 == 0 ? 0 : d0 - 1, 0);               // the first slab inside the volume (slab -1 of the first segment is padding)
  for (int zi = 0; zi < MLe + 2; zi += 3) {
    step(std::integral_constant<int, 0>{}, zi);
    if (zi + 1 < MLe + 2) step(std::integral_constant<int, 1>{}, zi + 1);
    if (zi + 2 < MLe + 2) step(std::integral_constant<int, 2>{}, zi + 2);
  }

  if (g.stat_part) {          // block-uniform: per-channel (sum, sum of squares) over the voxels of ALL slices of this block
    float a1 = st1 + __shfl_xor(st1, 16, 64), a2 = st2 + __shfl_xor(st2, 16, 64);
    a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
    if (lane < 16) { sred[(wv * 2) * 16 + lane] = a1; sred[(wv * 2 + 1) * 16 + lane] = a2; }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, c = tid & 15;
      if (c < g.Cout) {
        const int blk = (seg * g.tiles_h + th) * g.tiles_w + tw;
        g.stat_part[(((int64_t)n * g.stat_nblk + blk) * 2 + which) * g.Cout + c] =
            (sred[(0 + which) * 16 + c] + sred[(2 + which) * 16 + c]) + (sred[(4 + which) * 16 + c] + sred[(6 + which) * 16 + c]);
      }
    }
  }
}

template <typename T, typename TO, int RWO, int OCC>
static int cc16m_go_rows(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s) {
  const int ML = cc16m_ml(g.D);
  const size_t smem = (size_t)(RWO + 2) * 72 * 16 * sizeof(T) + 4 * 32 * 16 * sizeof(TO) + 8 * 16 * sizeof(float);
  g.tiles_h = cdiv(g.H, RWO); g.tiles_w = cdiv(g.W, CC16M_TW); g.dtiles = cdiv(g.D, ML);
  const int64_t blocks = (int64_t)g.N * g.dtiles * g.tiles_h * g.tiles_w;
  if (blocks > 2000000000LL) { dp_set_error("conv_cc16m: grid too large"); return 1; }
  if (g.NCH == 1 && !g.x3) hipLaunchKernelGGL((k_conv_cc16m<T, TO, true, RWO, OCC>), dim3((unsigned)blocks), dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (TO*)y, g, ML);
  else hipLaunchKernelGGL((k_conv_cc16m<T, TO, false, RWO, OCC>), dim3((unsigned)blocks), dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (TO*)y, g, ML);
  return 0;
}
template <typename T, typename TO>
static int cc16m_go(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s) {
  return cc16m_rows() == 4 ? cc16m_go_rows<T, TO, 4, 2>(x, wq, bias, y, g, s) : cc16m_go_rows<T, TO, 8, 1>(x, wq, bias, y, g, s);
}

int cc16_stat_blocks(int D, int H, int W, int k, int dtype) {
  if (cc16m_use(k, dtype)) return cdiv(D, cc16m_ml(D)) * cdiv(H, cc16m_rows()) * cdiv(W, CC16M_TW);      // k_conv_cc16m: one partial row per block
  return D * cdiv(H, 8) * cdiv(W, cc16_tw(W, k, dtype));
}

template <typename T, int KS, int DT, int OCC, typename TO>
static int cc16_go_impl(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s);
template <typename T, int KS, typename TO = T>
static int cc16_go(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s) {
  // (fp32 fragments are twice as wide: one depth slice per block keeps the parity mode's register spills down)
  // 3x3x3 with one input chunk: one depth slice per block = 64 accumulator registers -> three blocks per CU, whose staging / sweep /
  // epilogue phases overlap (16->16 at 2 x 128^3: 112 -> 100 us; with two chunks the two variants tie)
  if constexpr (KS == 3 && sizeof(T) == 2 && sizeof(TO) == 2) { if (g.NCH == 1) return cc16_go_impl<T, KS, 1, 3, TO>(x, wq, bias, y, g, s); }
  if constexpr (KS == 3 && sizeof(T) == 2 && sizeof(TO) == 4) {
    // DP_X3 with ONE real input chunk (16 -> 16 at the 128^3 level: two stagings, three sweeps per slab): one depth slice per block,
    // three blocks per CU, like the bf16 single-chunk launch
    static const int x3dt1 = [] { const char* e = getenv("DP_X3_DT1"); return e ? atoi(e) : 1; }();
    if (x3dt1 && g.x3 == 1) return cc16_go_impl<T, KS, 1, 3, TO>(x, wq, bias, y, g, s);
  }
  return cc16_go_impl<T, KS, (sizeof(T) == 4 ? 1 : 2), 2, TO>(x, wq, bias, y, g, s);
}
template <typename T, int KS, int DT, int OCC, typename TO>
static int cc16_go_impl(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s) {
  constexpr int ROWS = 8 + KS - 1, LP = (128 + KS - 1 + 7) & ~7;
  size_t smem = (size_t)ROWS * LP * 16 * sizeof(T);
  const size_t need = 4 * 2 * 32 * 16 * sizeof(TO) + 8 * 16 * sizeof(float);      // epilogue patches + statistics scratch
  if (smem < need) smem = need;
  auto kern = k_conv_cc16<T, KS, DT, OCC, TO>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { dp_set_error("conv_cc16: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return 1; }
  }
  g.dtiles = cdiv(g.D, DT);
  const int64_t blocks = (int64_t)g.N * g.dtiles * g.tiles_h * g.tiles_w;
  if (blocks > 2000000000LL) { dp_set_error("conv_cc16: grid too large"); return 1; }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (TO*)y, g);
  return 0;
}

template <typename T, int KS, typename TO = T, int NMT = 3, int DT = 2, int OCC = 2>
static int cc16w_go(const void* x, const void* wq, const float* bias, void* y, Cc16Geom g, hipStream_t s) {
  constexpr int ROWS = 8 + KS - 1, LP = (2 * NMT * 16 + KS - 1 + 7) & ~7;
  size_t smem = (size_t)ROWS * LP * 16 * sizeof(T);
  const size_t need = 4 * 2 * (NMT * 16) * 16 * sizeof(TO) + 8 * 16 * sizeof(float);      // epilogue patches + statistics scratch
  if (smem < need) smem = need;
  auto kern = k_conv_cc16w<T, KS, NMT, OCC, TO, DT>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { dp_set_error("conv_cc16w: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return 1; }
  }
  g.dtiles = cdiv(g.D, DT);
  const int64_t blocks = (int64_t)g.N * g.dtiles * g.tiles_h * g.tiles_w;
  if (blocks > 2000000000LL) { dp_set_error("conv_cc16w: grid too large"); return 1; }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (TO*)y, g);
  return 0;
}

bool cc16_wide(const void* y, int ldy, const void* y2, int ldy2, int osplit, int dtype) {
  const int es = (dtype == DP_F32 || dtype == DP_X3 || dtype == DP_X1) ? 4 : 2, epc = 16 / es;      // element size of the OUTPUT
  return (ldy * es) % 16 == 0 && (((uintptr_t)y & 15) == 0) && (!y2 || ((ldy2 * es) % 16 == 0 && (((uintptr_t)y2 & 15) == 0) && osplit % epc == 0));
}
int cc16_launch(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias, void* y, int ldy,
                void* y2, int ldy2, int osplit, float* stat_part, int N, int D, int H, int W, int Cin, int Cout, int k, int dtype, hipStream_t s) {
  Cc16Geom g;
  g.x3 = 0;
  if (dtype == DP_X3) {
    if (x2 || Cin % 48 || ldx < 2 * (Cin / 3)) { dp_set_error("conv_cc16: a DP_X3 launch takes ONE [x_hi | x_lo] tensor of 2/3 Cin channels (Cin = 3 x a multiple of 16)"); return 1; }
    g.x3 = Cin / 48;
  }
  { const char* e = getenv("DP_DBG"); g.dbg = e ? atoi(e) : 0; }
  g.y2 = y2; g.ldy2 = ldy2; g.osplit = osplit; g.wide = cc16_wide(y, ldy, y2, ldy2, osplit, dtype) ? 1 : 0;
  if (stat_part && !g.wide) { dp_set_error("conv_cc16: statistics need 16-byte aligned output rows"); return 1; }
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ldx = ldx; g.ldy = ldy; g.NCH = (Cin + 15) / 16;
  const int tile_w = cc16_tw(W, k, dtype);
  g.tiles_h = cdiv(H, 8); g.tiles_w = cdiv(W, tile_w); g.dtiles = 0;
  g.x2 = x2; g.ldx2 = ldx2; g.csplit = csplit; g.stat_part = stat_part; g.stat_nblk = cc16_stat_blocks(D, H, W, k, dtype);
  int rc = 0;
  if (cc16m_use(k, dtype) && g.wide) {                 // 3x3x3, 16-bit operands: marching along depth
    if (dtype == DP_BF16) rc = cc16m_go<bf16_t, bf16_t>(x, wq, bias, y, g, s);
    else if (dtype == DP_F16) rc = cc16m_go<f16_t, f16_t>(x, wq, bias, y, g, s);
    else rc = cc16m_go<bf16_t, float>(x, wq, bias, y, g, s);
    if (rc) return rc;
    DP_CHECK_LAUNCH("conv_cc16m"); return 0;
  }
  if (tile_w == 64 && dtype == DP_BF16 && k == 3) {
    static const int occ = [] { const char* e = getenv("DP_CC16W3"); return e ? atoi(e) : 0; }();
    rc = occ >= 5 ? cc16w_go<bf16_t, 3, bf16_t, 2, 1, 5>(x, wq, bias, y, g, s) : occ == 4 ? cc16w_go<bf16_t, 3, bf16_t, 2, 1, 4>(x, wq, bias, y, g, s)
                  : occ == 3 ? cc16w_go<bf16_t, 3, bf16_t, 2, 1, 3>(x, wq, bias, y, g, s) : cc16w_go<bf16_t, 3, bf16_t, 2, 2, 2>(x, wq, bias, y, g, s);
    if (rc) return rc;
    DP_CHECK_LAUNCH("conv_cc16w"); return 0;
  }
  if (tile_w == 64 && dtype == DP_BF16 && k == 7) {
    rc = cc16w_go<bf16_t, 7, bf16_t, 2>(x, wq, bias, y, g, s);
    if (rc) return rc;
    DP_CHECK_LAUNCH("conv_cc16w"); return 0;
  }
  if (tile_w == 64) { dp_set_error("conv_cc16: DP_CC16W_64 is a bf16 7x7x7 experiment"); return 1; }
  if (tile_w == 96) {          // (never exact fp32: cc16_tw)
    if (dtype == DP_BF16) rc = k == 7 ? cc16w_go<bf16_t, 7>(x, wq, bias, y, g, s) : cc16w_go<bf16_t, 3>(x, wq, bias, y, g, s);
    else if (dtype == DP_F16) rc = k == 7 ? cc16w_go<f16_t, 7>(x, wq, bias, y, g, s) : cc16w_go<f16_t, 3>(x, wq, bias, y, g, s);
    else if (dtype == DP_X3 || dtype == DP_X1) rc = k == 7 ? cc16w_go<bf16_t, 7, float>(x, wq, bias, y, g, s) : cc16w_go<bf16_t, 3, float>(x, wq, bias, y, g, s);
    else { dp_set_error("conv_cc16: bad dtype"); return 1; }
    if (rc) return rc;
    DP_CHECK_LAUNCH("conv_cc16w"); return 0;
  }
  if (dtype == DP_BF16) rc = k == 7 ? cc16_go<bf16_t, 7>(x, wq, bias, y, g, s) : cc16_go<bf16_t, 3>(x, wq, bias, y, g, s);
  else if (dtype == DP_F16) rc = k == 7 ? cc16_go<f16_t, 7>(x, wq, bias, y, g, s) : cc16_go<f16_t, 3>(x, wq, bias, y, g, s);
  else if (dtype == DP_F32) rc = k == 7 ? cc16_go<float, 7>(x, wq, bias, y, g, s) : cc16_go<float, 3>(x, wq, bias, y, g, s);
  else if (dtype == DP_X3 || dtype == DP_X1) rc = k == 7 ? cc16_go<bf16_t, 7, float>(x, wq, bias, y, g, s) : cc16_go<bf16_t, 3, float>(x, wq, bias, y, g, s);
  else { dp_set_error("conv_cc16: bad dtype"); return 1; }
  if (rc) return rc;
  DP_CHECK_LAUNCH("conv_cc16"); return 0;
}
