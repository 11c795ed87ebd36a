// Weight gradient of the 7x7x7 convolutions on planes of at least 32 x 32 voxels, second generation: K along H.
// One block = one kd, one tile of 16 input channels and one tile of 16 output channels (wider layers multiply the grid).
//   dW[co][ci][kd][kh][kw] = sum_{n,d,h,w} gy[n,d,h,w,co] * x[n, d+kd-P, h+kh-P, w+kw-P, ci]        (stride 1, "same" padding)
// v_mfma_f32_16x16x32: M = 16 input channels, N = 16 output channels, ONE tap per MFMA, K = 32 voxels of ONE COLUMN (32 consecutive
// rows h of a fixed w).  k_wgrad_cc16 (conv_tiled.hip) lays K along W: its kw taps are register windows shifted by v_perm /
// v_mov (22 VALU per 14 MFMAs) and its loop exposes three LDS latencies per row; PMC showed its waves issuing 40 % of the time and
// the matrix pipe busy 35 %.  With K along H a kw (or kh) shift is just ANOTHER LDS ADDRESS of the same fragment shape:
//   * the x fragment (rows r0+kh .. +31, column c) meets the gy fragments of columns c-kw, kw = 0..KS-1: a wave keeps the last KS
//     gy fragments in a register ring and loads, per column step, ONE gy fragment and one x fragment per owned kh:
//     6 transpose reads per 13 MFMAs (KS = 7, two kh per wave; wave roles: RolePair / RoleLast below), no VALU at all;
//   * the column loop is fully unrolled (static ring indices, static immediate offsets, ramp-up / ramp-down MFMAs removed at
//     compile time) and every fragment is requested 6-13 MFMAs before its first use, into the registers its predecessor just left;
//   * k -> row map with bits 2 and 3 swapped and ODD row pitches (in voxels): the 32 lanes of a ds_read_b64_tr_b16 half touch 8
//     consecutive rows = all 64 banks once (the natural map puts rows r and r+8 on the same banks: 2-way conflict).
// Tiles: 32 rows x 32 columns of gy, (32+KS-1)^2 of x (halo factor 1.41 instead of 1.91 for the 8 x 64 tile), register-staged one
// tile ahead as in k_wgrad_tiled; grid decode, scratch layout ([tap][ci][co] fp32, atomics) and the unpack kernel are shared with it.
#include "common.h"
#include <utility>

namespace {

template <typename T, int KS>
struct HkCfg {
  static constexpr int PAD = KS / 2, TH = 32, TW = 32, XC = 16, GC = 16;
  static constexpr int LR = TH + KS - 1, LC = TW + KS - 1, LP = LC | 1, GP = TW + 1;      // odd pitches (voxels)
  static constexpr int CW = TW;                                 // gy columns a wave walks over (every wave: the whole tile width)
  static constexpr int STEPS = CW + KS - 1;                     // x columns a wave walks over
  static constexpr size_t SMEM = ((size_t)LR * LP * XC + (size_t)TH * GP * GC) * sizeof(T);
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Tap sets of the four waves (KS = 7: 49 taps of one kd).  Rows = kh, one x fragment per row and column step; a row's kw mask
// selects the taps the wave accumulates.  Two whole kh per wave leaves the fourth wave half idle (14 / 14 / 14 / 7 taps: 87.5 %);
// here waves 0-2 hand the kw = 6 tap of their even kh to wave 3: 13 / 13 / 13 / 10 (94 %).
struct RolePair {                         // waves 0..2: kh = 2w (kw 0..5) and 2w + 1 (all kw); rows relative to kh0 = 2w
  static constexpr int NR = 2, NACC = 13;
  static constexpr int row(int r) { return r; }
  static constexpr unsigned mask(int r) { return r == 0 ? 0x3Fu : 0x7Fu; }
  static constexpr bool dbuf(int) { return false; }
};
struct RoleLast {                         // wave 3: kh = 6 (all kw) and the kw = 6 taps of kh 0, 2, 4; absolute rows (kh0 = 0)
  static constexpr int NR = 4, NACC = 10;
  static constexpr int row(int r) { return r == 0 ? 6 : 2 * (r - 1); }
  static constexpr unsigned mask(int r) { return r == 0 ? 0x7Fu : 0x40u; }
  static constexpr bool dbuf(int r) { return r == 0; }          // (at most one double-buffered row per role)
};
template <typename R> constexpr int role_slot(int r, int kw) {     // accumulator index of tap (row r, kw)
  int s = 0;
  for (int i = 0; i < r; i++) s += __builtin_popcount(R::mask(i));
  return s + __builtin_popcount(R::mask(r) & ((1u << kw) - 1u));
}
// does row r meet any live gy column at x column step cc?  (gy column = cc - kw, 0 <= . < CW)
template <typename R> constexpr bool role_row_live(int r, int cc, int KS, int CW) {
  for (int kw = 0; kw < KS; kw++) if ((R::mask(r) >> kw & 1u) && cc - kw >= 0 && cc - kw < CW) return true;
  return false;
}

// BUF (round 6): the tile prefetch as BUFFER loads woven into the sweep.  Wave-cycle counters of the round-2 form (profiles/r06_a_wgrad_hk_stalls.md):
// the prefetch of a tile is ~20 VALU + ~10 SALU per 16-byte piece (a division by 38 for the piece's halo coordinates, four bounds
// compares, a 64-bit address, four v_cndmask to zero an out-of-volume piece) -- 4.2e7 VALU per launch next to 4.3e7 MFMAs -- issued as ONE
// block between two barriers, where the matrix pipe has nothing to do: without staging the kernel is 20 % faster (MFMA-busy 55 -> 69 %).
// Now (a) a thread owns a COLUMN of the halo tile (piece, column lc, row group r of 3) and walks rows r, r + 3, ...: every piece is the
// previous offset + 3 rows, every LDS write an immediate offset; (b) the loads go through a buffer descriptor of the PLANE
// (num_records = H * W * ld bytes), so rows above / below the volume are out of range by unsigned wrap-around and come back as zeros
// from the hardware -- no compare, no select; a column outside the volume (or a channel piece beyond the tensor) parks the thread's
// base offset at 2^31; (c) with 1-2 VALU per piece the 21 pieces of the NEXT tile are issued one per column step inside the sweep.
// Needs a block-uniform source tensor: virtual concats split on a multiple of 16 channels (else BUF = false, the round-2 form).
template <typename T, int KS, bool BUF>
__global__ void __launch_bounds__(256, 2) k_wgrad_hk(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ dwt, WgHkGeom g) {
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  using C = HkCfg<T, KS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = (T*)smem_raw;
  T* gs = xs + (size_t)C::LR * C::LP * C::XC;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i16 = lane & 15;
  int kd, mt, nt, yb;
  {
    const int L = blockIdx.x, inner = KS * g.MT * g.NTn;
    int c;
    if (g.ydim % 8 == 0) { const int xcd = L & 7, slot = L >> 3; c = slot % inner; yb = (slot / inner) * 8 + xcd; }
    else { c = L % inner; yb = L / inner; }
    kd = c % KS; const int r = c / KS; mt = r % g.MT; nt = r / g.MT;
  }
  static_assert(KS == 7, "wave roles are laid out for 7 x 7 x 7");
  const bool last_role = wv == 3;
  const int kh0 = last_role ? 0 : 2 * wv, g0 = 0;
  const int per_plane = g.tiles_h * g.tiles_w, units = g.N * g.D * per_plane, per = (units + g.ydim - 1) / g.ydim;
  const int u0 = yb * per, u1 = min(units, u0 + per);
  // lane part of the transposing reads: k = 8q + j (+4 for the second read of a pair) sits on row 16(q>>1) + 4(q&1) + j (+8)
  const int rowoff = 16 * (q >> 1) + 4 * (q & 1) + (i16 >> 2), c4 = 4 * (i16 & 3);
  const T* const xb = xs + (rowoff + kh0) * C::LP * C::XC + g0 * C::XC + c4;
  const T* const gb = gs + rowoff * C::GP * C::GC + g0 * C::GC + c4;

  v4f acc[RolePair::NACC];
#pragma unroll
  for (int a = 0; a < RolePair::NACC; a++) acc[a] = (v4f){0.f, 0.f, 0.f, 0.f};

  struct Tile { int n, d, id, h0, w0; };
  // Units are walked DEPTH-FASTEST inside a tile column (n, th, tw): the KS kd-blocks of a share (same XCD, same unit range) read
  // the same x tile in the same step and the gy tile (column, d) in KS CONSECUTIVE steps (d = id - kd + PAD), so their XCD's L2
  // serves six of the seven reads.  With plane-major units (round 2) a gy tile came back 16 tiles x kd later -- longer than a
  // share's whole walk: FETCH_SIZE 1.6 GB per launch for 270-400 MB of operands (profiles/r03_r_pmc_traffic.md).
  auto tile_ok = [&](int u) { const int id = u % g.D, d = id - kd + C::PAD; return d >= 0 && d < g.D; };
  auto tile_of = [&](int u) {
    Tile t; const int col = u / g.D, tw = col % g.tiles_w, th = (col / g.tiles_w) % g.tiles_h;
    t.id = u - col * g.D; t.n = col / per_plane; t.d = t.id - kd + C::PAD; t.h0 = th * C::TH; t.w0 = tw * C::TW;
    return t;
  };
  auto next_ok = [&](int u) { u++; while (u < u1 && !tile_ok(u)) u++; return u; };

  // register-staged tiles: 16-byte pieces (8 channels), two per voxel
  constexpr int PX = (C::LR * C::LC * 2 + 255) / 256, PG = C::TH * C::TW * 2 / 256;
  v4u rx[PX], rg[PG];
  const int piece = tid & 1, cpiece = mt * C::XC + piece * 8;
  const bool xsecond = g.x2 && cpiece >= g.csplit;
  const T* xsrc = (xsecond ? (const T*)g.x2 : x) + cpiece - (xsecond ? g.csplit : 0);
  const int ldsrc = xsecond ? g.ldx2 : g.ldx;
  const T* gsrc = gy + nt * C::GC + piece * 8;
  const bool x_exists = cpiece + 8 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx), g_exists = nt * C::GC + piece * 8 + 8 <= g.ldgy;
  // x piece j of this thread: voxel j * 128 + tid / 2 of the LR x LC halo tile (recomputed where needed: a coordinate table in
  // registers pushed the kernel over 256 VGPRs, and spilled coordinates serialise the global loads behind scratch reloads)
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };      // defeats loop-invariant hoisting of the coordinates
  auto xvox = [&](int t2, int j, int& lr, int& lc) { const int v = j * 128 + t2; lr = v / C::LC; lc = v - lr * C::LC; return v < C::LR * C::LC; };
  const int g_r = tid >> 6, g_c = (tid >> 1) & 31;       // gy piece j: row 4j + g_r, column g_c
  auto issue = [&](const Tile& t) {
    const T* xplane = xsrc + (((int64_t)t.n * g.D + t.id) * g.H) * (int64_t)g.W * ldsrc;
    const int ihb = t.h0 - C::PAD, iwb = t.w0 - C::PAD, t2 = opaque(tid >> 1);
#pragma unroll
    for (int j = 0; j < PX; j++) {
      int lr, lc; const bool in = xvox(t2, j, lr, lc);
      const int ih = ihb + lr, iw = iwb + lc;
      const bool ok = x_exists && in && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      v4u v = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
      rx[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
    const T* gplane = gsrc + (((int64_t)t.n * g.D + t.d) * g.H) * (int64_t)g.W * g.ldgy;
#pragma unroll
    for (int j = 0; j < PG; j++) {
      const int oh = t.h0 + 4 * j + g_r, ow = t.w0 + g_c;
      const bool ok = g_exists && oh < g.H && ow < g.W;
      v4u v = *(const v4u*)(gplane + (ok ? (oh * g.W + ow) * g.ldgy : 0));
      rg[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
  };
  auto commit = [&]() {
    const int t2 = opaque(tid >> 1);
#pragma unroll
    for (int j = 0; j < PX; j++) {
      int lr, lc;
      if (xvox(t2, j, lr, lc)) *(v4u*)(xs + (lr * C::LP + lc) * C::XC + piece * 8) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < PG; j++) *(v4u*)(gs + ((4 * j + g_r) * C::GP + g_c) * C::GC + piece * 8) = rg[j];
  };
  // ---- BUF staging (see the kernel's header comment)
  constexpr int RG3 = 3, PXB = (C::LR + RG3 - 1) / RG3;                    // 13 x pieces per thread (the last one only for row groups 0, 1)
  static_assert(!BUF || (2 * C::LC * RG3 <= 256 && PG * 4 == C::TH), "thread map of the buffer-load staging");
  v4u bx[PXB], bg[PG];
  const int b_t2 = tid >> 1, b_rg = b_t2 / C::LC, b_lc = b_t2 - b_rg * C::LC;
  const bool b_xsecond = g.x2 && mt * C::XC >= g.csplit;                   // block-uniform (csplit % 16 == 0)
  const T* const b_xsrc = b_xsecond ? (const T*)g.x2 : x;
  const int b_ldx = b_xsecond ? g.ldx2 : g.ldx, b_c0 = mt * C::XC - (b_xsecond ? g.csplit : 0) + piece * 8;
  const bool b_xthr = b_rg < RG3 && b_c0 + 8 <= b_ldx;
  const unsigned b_cx = (unsigned)(((b_rg * g.W + b_lc) * b_ldx + b_c0) * 2), b_xrow3 = (unsigned)(RG3 * g.W * b_ldx * 2);
  const unsigned b_cg = (unsigned)(((g_r * g.W + g_c) * g.ldgy + nt * C::GC + piece * 8) * 2), b_grow4 = (unsigned)(4 * g.W * g.ldgy * 2);
  const unsigned b_xbytes = (unsigned)(g.H * g.W * b_ldx * 2), b_gbytes = (unsigned)(g.H * g.W * g.ldgy * 2);
  // LDS home of the thread's x pieces: + k * RG3 * LP * XC (immediate).  The 28 threads without a column (row group 3) write their
  // zeros into the PAD column (lc = LC, never read: the row pitch is LC | 1) of rows 0, 3, ...: no branch in the commit -- a skipped
  // commit is a path on which the loads are never waited for, and the waitcnt pass then stalls the sweep on them
  static_assert(C::LP > C::LC, "the pad column takes the idle threads' writes");
  T* const b_xl = xs + ((b_rg < RG3 ? b_rg : 0) * C::LP + (b_rg < RG3 ? b_lc : C::LC)) * C::XC + piece * 8;
  T* const b_xl_last = (b_rg * 1 + (PXB - 1) * RG3 < C::LR || b_rg >= RG3) ? b_xl + (PXB - 1) * RG3 * C::LP * C::XC
                                                                          : xs + C::LC * C::XC + piece * 8;      // (row 38 does not exist: pad column of row 0)
  T* const b_gl = gs + (g_r * C::GP + g_c) * C::GC + piece * 8;            // + j * 4 * GP * GC
  auto rsrc_of = [](const T* base, unsigned bytes) {
    // (descriptor inputs made provably wave-uniform: a descriptor the compiler believes divergent is wrapped in a waterfall loop per load)
    const uint64_t a = (uint64_t)(uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  struct BufTile { __amdgpu_buffer_rsrc_t rx, rg; unsigned vx, vg; };
  auto buf_tile = [&](const Tile& t, bool live) {
    BufTile b;
    b.rx = rsrc_of(b_xsrc + (((int64_t)t.n * g.D + t.id) * g.H) * (int64_t)g.W * b_ldx, b_xbytes);
    b.rg = rsrc_of(gy + (((int64_t)t.n * g.D + t.d) * g.H) * (int64_t)g.W * g.ldgy, b_gbytes);
    const int ihb = t.h0 - C::PAD, iwb = t.w0 - C::PAD;
    const unsigned sx = (unsigned)((ihb * g.W + iwb) * b_ldx * 2), sg = (unsigned)((t.h0 * g.W + t.w0) * g.ldgy * 2);
    const bool xok = live && b_xthr && (unsigned)(iwb + b_lc) < (unsigned)g.W, gok = live && g_exists && t.w0 + g_c < g.W;
    b.vx = xok ? b_cx + sx : 0x80000000u;       // rows above the volume: negative -> wraps beyond num_records -> zeros from the hardware
    b.vg = gok ? b_cg + sg : 0x80000000u;
    return b;
  };
  auto buf_issue_x = [&](const BufTile& b, auto k_) {
    constexpr int k = decltype(k_)::value;
    unsigned v = b.vx + (unsigned)k * b_xrow3;
    if constexpr (k * RG3 + RG3 > C::LR) { if (k * RG3 + b_rg >= C::LR) v = 0x80000000u; }      // (row 38 of row group 2 does not exist)
    bx[k] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(b.rx, v, 0, 0));
  };
  auto buf_issue_g = [&](const BufTile& b, auto j_) {
    constexpr int j = decltype(j_)::value;
    bg[j] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(b.rg, b.vg + (unsigned)j * b_grow4, 0, 0));
  };
  auto buf_commit = [&]() {
    static_for<0, PXB>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if constexpr (k == PXB - 1) *(v4u*)b_xl_last = bx[k];
      else *(v4u*)(b_xl + k * RG3 * C::LP * C::XC) = bx[k];
    });
    static_for<0, PG>([&](auto j_) { constexpr int j = decltype(j_)::value; *(v4u*)(b_gl + j * 4 * C::GP * C::GC) = bg[j]; });
  };
  // one tile: the wave walks over the STEPS x columns of the tile's CW gy columns; everything below is static after unrolling.
  // Software pipeline with (almost) no second buffers -- the kernel sits at the 256-VGPR limit of two waves per SIMD, and a
  // spilled register costs an s_waitcnt vmcnt(0) in front of the tile prefetch: the gy fragment of column cc+1 is requested at
  // the start of step cc (ring of KS+1), an x row requests its next column right after its own last MFMA of the step, i.e. the
  // other rows' 6-9 MFMAs (100-150 cycles) before its first use; only RoleLast's 7-tap row, which has just three MFMAs of other
  // rows behind it, keeps two buffers (that role has three accumulators fewer).
  auto sweep = [&]<typename R, typename H>(R, H&& hook) {
    Frag8<T> X[R::NR], Xd[2], G[KS + 1];
    auto ldx = [&](auto r_, auto cc_) {
      constexpr int r = decltype(r_)::value, cc = decltype(cc_)::value;
      if constexpr (cc < C::STEPS && role_row_live<R>(r, cc, KS, C::CW)) {
        const Frag8<T> f = tr_pair<8 * C::LP * C::XC, T>(xb + (R::row(r) * C::LP + cc) * C::XC);
        if constexpr (R::dbuf(r)) Xd[cc & 1] = f; else X[r] = f;
      }
    };
    auto ldg = [&](auto gc_) {
      constexpr int gc = decltype(gc_)::value;
      if constexpr (gc < C::CW) G[gc % (KS + 1)] = tr_pair<8 * C::GP * C::GC, T>(gb + gc * C::GC);
    };
    static_for<0, R::NR>([&](auto r_) { ldx(r_, std::integral_constant<int, 0>{}); });
    ldg(std::integral_constant<int, 0>{});
    static_for<0, C::STEPS>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      constexpr std::integral_constant<int, cc + 1> nx{};
      ldg(nx);
      static_for<0, R::NR>([&](auto r_) { if constexpr (R::dbuf(decltype(r_)::value)) ldx(r_, nx); });
      hook(cc_);                           // (BUF: one piece of the next tile's prefetch per column step)
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, R::NR>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        static_for<0, KS>([&](auto kw_) {
          constexpr int kw = decltype(kw_)::value, gc = cc - kw;
          if constexpr (gc >= 0 && gc < C::CW && (R::mask(r) >> kw & 1u)) {
            constexpr int sl = role_slot<R>(r, kw);
            acc[sl] = mma16(R::dbuf(r) ? Xd[cc & 1] : X[r], G[gc % (KS + 1)], acc[sl]);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!R::dbuf(r)) ldx(r_, nx);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  const bool fast = (g.ldx % 8 == 0) && (g.ldgy % 8 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)gy & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 8 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0)));
  if (!fast) return;                                                // the launcher only selects this kernel for aligned operands
  int cu = u0;
  while (cu < u1 && !tile_ok(cu)) cu++;
  const bool nostage = g.dbg & 1, nosweep = g.dbg & 2;               // experiments only (env DP_DBG)
  auto nohook = [](auto) {};
  if constexpr (BUF) {
    // (DP_DBG bit 0, "no staging", parks the offsets out of range instead of skipping code: every load is still issued and every
    // commit still runs, so the compiler sees ONE path on which each load is waited for exactly once -- with the commit under a runtime
    // condition its waitcnt pass had to assume loads still in flight at the top of the sweep and stalled the column walk on vmcnt)
    if (cu < u1) {
      const BufTile b = buf_tile(tile_of(cu), !nostage);
      static_for<0, PXB>([&](auto k_) { buf_issue_x(b, k_); });
      static_for<0, PG>([&](auto j_) { buf_issue_g(b, j_); });
    }
    while (cu < u1) {
      lds_barrier();
      buf_commit();
      const int nu = next_ok(cu);
      // (no next tile: every offset parked out of range -- the loads come back as zeros without touching memory)
      const BufTile b = buf_tile(tile_of(nu < u1 ? nu : cu), nu < u1 && !nostage);
      // two pieces per column step from step 0 (x 0..12, then gy 0..7): the last request leaves at step 10 of 38, ~350 MFMAs (2.7 us)
      // before the commit asks for it -- with one piece per step up to step 25 the commit still waited on the tail (MFMA-busy 59.9 %)
      auto piece_at = [&](auto p_) {
        constexpr int pc = decltype(p_)::value;
        if constexpr (pc < PXB) buf_issue_x(b, std::integral_constant<int, pc>{});
        else if constexpr (pc < PXB + PG) buf_issue_g(b, std::integral_constant<int, pc - PXB>{});
      };
      auto hook = [&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        piece_at(std::integral_constant<int, 2 * cc>{});
        piece_at(std::integral_constant<int, 2 * cc + 1>{});
      };
      lds_barrier();
      // (DP_DBG bit 1, "no sweep": nothing is issued either -- a second code path with 21 loads in flight falls through, in the CFG, into
      // the sweep and makes the waitcnt pass stall the column walk on loads that are never pending there)
      if (!nosweep) { if (last_role) sweep(RoleLast{}, hook); else sweep(RolePair{}, hook); }
      cu = nu;
    }
  } else {
    if (cu < u1 && !nostage) issue(tile_of(cu));
    while (cu < u1) {
      lds_barrier();
      if (!nostage) commit();
      const int nu = next_ok(cu);
      if (nu < u1 && !nostage) issue(tile_of(nu));
      lds_barrier();
      if (!nosweep) { if (last_role) sweep(RoleLast{}, nohook); else sweep(RolePair{}, nohook); }
      cu = nu;
    }
  }
  if (g.dbg & 4) return;
  // C/D of the 16x16 MFMA: col (co) = lane&15, row (ci) = 4*(lane>>4) + e
  const int co = nt * 16 + (lane & 15);
  if (co >= g.Cout) return;
  dwt += (int64_t)yb * g.slab;            // (deterministic mode: one scratch slab per voxel share)
  auto flush = [&]<typename R>(R) {
    static_for<0, R::NR>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      static_for<0, KS>([&](auto kw_) {
        constexpr int kw = decltype(kw_)::value;
        if constexpr (R::mask(r) >> kw & 1u) {
          constexpr int sl = role_slot<R>(r, kw);
          const int kh = kh0 + R::row(r);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int ci = mt * 16 + 4 * q + e;
            if (ci < g.Cin) atomicAdd(dwt + ((int64_t)((kd * KS + kh) * KS + kw) * g.Cin + ci) * g.Cout + co, acc[sl][e]);
          }
        }
      });
    });
  };
  if (last_role) flush(RoleLast{}); else flush(RolePair{});
}

template <typename T, int KS, bool BUF>
int launch_hk_impl(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s);
template <typename T, int KS>
int launch_hk(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s) {
  // buffer-load staging (BUF) needs a block-uniform source tensor, 32-bit plane offsets and 16-byte aligned rows; DP_HK_BUF=0: the round-2 form
  static const int buf_env = [] { const char* e = getenv("DP_HK_BUF"); return e ? atoi(e) : 1; }();
  const int64_t xplane = (int64_t)g.H * g.W * (g.ldx > g.ldx2 ? g.ldx : g.ldx2) * 2, gplane = (int64_t)g.H * g.W * g.ldgy * 2;
  const bool buf = buf_env && (!g.x2 || g.csplit % 16 == 0) && xplane < (1LL << 30) && gplane < (1LL << 30);
  return buf ? launch_hk_impl<T, KS, true>(x, gy, ws, g, s) : launch_hk_impl<T, KS, false>(x, gy, ws, g, s);
}
template <typename T, int KS, bool BUF>
int launch_hk_impl(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s) {
  using C = HkCfg<T, KS>;
  auto kern = k_wgrad_hk<T, KS, BUF>;
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    if (e != hipSuccess) { dp_set_error("wgrad_hk: cannot raise dynamic LDS to %zu: %s", C::SMEM, hipGetErrorString(e)); return 1; }
    raised = true;
  }
  g.tiles_h = cdiv(g.H, C::TH); g.tiles_w = cdiv(g.W, C::TW);
  g.MT = cdiv(g.Cin, C::XC); g.NTn = cdiv(g.Cout, C::GC);
  const int units = g.N * g.D * g.tiles_h * g.tiles_w;
  static int occ = 0, ncu = 0;
  if (!occ) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, C::SMEM) != hipSuccess || occ < 1) occ = 1;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  static const int occ_env = [] { const char* e = getenv("DP_HK_OCC"); return e ? atoi(e) : 0; }();      // (experiment: blocks per CU of the persistent grid)
  int want = (ncu * (occ_env > 0 ? occ_env : occ)) / (KS * g.MT * g.NTn); if (want < 1) want = 1;
  int ydim = units < want ? units : want;
  // Shares in multiples of 8 whenever there are at least 8: only then does the block decode give every XCD whole shares (all kd /
  // channel-tile blocks of a unit range behind ONE L2).  Round 2 rounded down only when that cost < 5 % of the blocks; the layers
  // with several channel tiles (want = 36 or 9) then ran with their blocks dealt round-robin over the XCDs: FETCH_SIZE 5.1 GB per
  // launch for the 403 MB of a 32 -> 16 layer at 2 x 128^3, 2.9 GB for the 100 MB of 64 -> 32 at 64^3.  With whole shares (and the
  // depth-fastest walk above): 0.66 GB and 0.18 GB; the launches take the same time (they are matrix-bound either way) and leave
  // 12 GB per step of fabric traffic to the kernels on the other streams.  DP_HK_ROUND8=0: the old rule.
  static const int round8 = [] { const char* e = getenv("DP_HK_ROUND8"); return e ? atoi(e) : 1; }();
  if (g.max_slabs && ydim > g.max_slabs) ydim = g.max_slabs;
  if (ydim >= 8 && ((ydim & 7) * 20 <= ydim || round8)) ydim &= ~7;
  g.ydim = ydim;
  if (g.nslab_out) *g.nslab_out = g.max_slabs ? ydim : 1;
  hipLaunchKernelGGL(kern, dim3(KS * ydim * g.MT * g.NTn), dim3(256), C::SMEM, s, (const T*)x, (const T*)gy, ws, g);
  return 0;
}


// ================================================================================================ 3 x 3 x 3: all 27 taps per block, marching along depth
// The 3^3 weight gradients are fabric-bound (AI ~ 200 FLOP/B): what matters is that every x / gy plane is staged ONCE.  The kernels
// of conv_tiled.hip give each kd its own block (3 x the staging) and re-read the +-1 halo rows of 8-row tiles (1.25 x).  Here a block
// owns a 32 x 32 column of gy over a depth segment and marches through it: the three x planes a gy plane needs live in an LDS ring
// (148 KB with the gy tile: one block per CU), plane d+2 and gy plane d+1 are in flight in registers while plane d is swept, and
// the 27 tap accumulators (108 registers) stay in the wave for the whole segment.  Wave w walks gy columns 8w .. 8w+7 with K along H
// as above: per column step 9 x fragments ((kd, kh) rows) + 1 gy fragment for 27 MFMAs.
constexpr int HK3_MAX_BLOCKS = 512;          // slabs of 27 x 256 floats in the scratch (dp_conv3d_wgrad_tiled_ws_elems)

template <typename T>
struct Hk3Cfg {
  static constexpr int KS = 3, PAD = 1, TH = 32, TW = 32, XC = 16, GC = 16;
  static constexpr int LR = TH + 2, LC = TW + 2, LP = LC | 1, GP = TW + 1;
  static constexpr int CW = TW / 4, STEPS = CW + KS - 1, NROW = KS * KS;
  static constexpr int XPLANE = LR * LP * XC;                   // elements of one x plane tile
  static constexpr size_t SMEM = ((size_t)3 * XPLANE + (size_t)TH * GP * GC) * sizeof(T);
};

// BUF (round 6): the plane prefetch as buffer loads, as in k_wgrad_hk<7, BUF> -- a block owns ONE column of tiles for its whole life, so a thread's
// byte offsets are kernel constants (thread = (piece, column lc, row group of 3), rows walked 3 at a time); only the plane's buffer descriptor
// changes per step, and a plane outside the volume is a descriptor of zero bytes.  With one wave per SIMD (148 KB of LDS per block) the
// ~20 VALU per piece of the old form were time the matrix pipe simply waited.
template <typename T, bool BUF>
__global__ void __launch_bounds__(256, 1) k_wgrad_hk3(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ dwt, WgHkGeom g) {
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  using C = Hk3Cfg<T>;
  constexpr int KS = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = (T*)smem_raw;
  T* gs = xs + (size_t)3 * C::XPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i16 = lane & 15;
  // block -> (column (n, th, tw), depth segment, input-channel tile, output-channel tile); ydim = depth segments
  int b = blockIdx.x;
  const int seg = b % g.ydim; b /= g.ydim; const int mt = b % g.MT; b /= g.MT; const int nt = b % g.NTn; b /= g.NTn;
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; const int n = b / g.tiles_h;
  const int DS = (g.D + g.ydim - 1) / g.ydim, z0 = seg * DS, z1 = min(g.D, z0 + DS);       // (the launcher leaves no segment empty)
  const int h0 = th * C::TH, w0 = tw * C::TW, g0 = wv * C::CW;
  const int rowoff = 16 * (q >> 1) + 4 * (q & 1) + (i16 >> 2), c4 = 4 * (i16 & 3);
  const T* const xl = xs + rowoff * C::LP * C::XC + g0 * C::XC + c4;          // + slot * XPLANE + (kh * LP + column) * XC
  const T* const gb = gs + rowoff * C::GP * C::GC + g0 * C::GC + c4;

  v4f acc[27];
#pragma unroll
  for (int a = 0; a < 27; a++) acc[a] = (v4f){0.f, 0.f, 0.f, 0.f};

  constexpr int RG3 = 3, PXB = (C::LR + RG3 - 1) / RG3;
  constexpr int PX = BUF ? PXB : (C::LR * C::LC * 2 + 255) / 256, PG = C::TH * C::TW * 2 / 256;
  static_assert(!BUF || (2 * C::LC * RG3 <= 256 && C::LP > C::LC), "thread map of the buffer-load staging");
  v4u rxs[2][PX], rgs[2][PG];          // two register sets: the loads of plane d+3 / gy d+2 are issued while plane d is swept
  const int piece = tid & 1, cpiece = mt * C::XC + piece * 8;
  const bool xsecond = g.x2 && cpiece >= g.csplit;
  const T* xsrc = (xsecond ? (const T*)g.x2 : x) + cpiece - (xsecond ? g.csplit : 0);
  const int ldsrc = xsecond ? g.ldx2 : g.ldx;
  const T* gsrc = gy + nt * C::GC + piece * 8;
  const bool x_exists = cpiece + 8 <= (g.x2 ? g.csplit + g.ldx2 : g.ldx), g_exists = nt * C::GC + piece * 8 + 8 <= g.ldgy;
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto xvox = [&](int t2, int j, int& lr, int& lc) { const int v = j * 128 + t2; lr = v / C::LC; lc = v - lr * C::LC; return v < C::LR * C::LC; };
  const int g_r = tid >> 6, g_c = (tid >> 1) & 31;
  // ---- BUF staging: kernel-constant per-thread offsets
  const int b_t2 = tid >> 1, b_rg = b_t2 / C::LC, b_lc = b_t2 - b_rg * C::LC;
  const bool b_xsecond = g.x2 && mt * C::XC >= g.csplit;                   // block-uniform (csplit % 16 == 0)
  const T* const b_xsrc = b_xsecond ? (const T*)g.x2 : x;
  const int b_ldx = b_xsecond ? g.ldx2 : g.ldx, b_c0 = mt * C::XC - (b_xsecond ? g.csplit : 0) + piece * 8;
  const bool b_xok = b_rg < RG3 && b_c0 + 8 <= b_ldx && (unsigned)(w0 - C::PAD + b_lc) < (unsigned)g.W;
  const unsigned b_xrow3 = (unsigned)(RG3 * g.W * b_ldx * 2), b_grow4 = (unsigned)(4 * g.W * g.ldgy * 2);
  const unsigned b_vx = b_xok ? (unsigned)((((h0 - C::PAD + b_rg) * g.W + (w0 - C::PAD + b_lc)) * b_ldx + b_c0) * 2) : 0x80000000u;   // (rows above the volume wrap out of range)
  const unsigned b_vg = (g_exists && w0 + g_c < g.W) ? (unsigned)((((h0 + g_r) * g.W + w0 + g_c) * g.ldgy + nt * C::GC + piece * 8) * 2) : 0x80000000u;
  const unsigned b_xbytes = (unsigned)(g.H * g.W * b_ldx * 2), b_gbytes = (unsigned)(g.H * g.W * g.ldgy * 2);
  const int b_xl0 = ((b_rg < RG3 ? b_rg : 0) * C::LP + (b_rg < RG3 ? b_lc : C::LC)) * C::XC + piece * 8;       // (idle threads: the pad column)
  const int b_xl_last = (b_rg + (PXB - 1) * RG3 < C::LR || b_rg >= RG3) ? b_xl0 + (PXB - 1) * RG3 * C::LP * C::XC : C::LC * C::XC + piece * 8;
  auto rsrc_of = [](const T* base, unsigned bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  auto issue_x = [&](int p, v4u* rx, bool live = true) {                   // x plane p (zeros outside the volume) -> rx
    if constexpr (BUF) {
      const bool pin = live && p >= 0 && p < g.D;
      const auto rs = rsrc_of(b_xsrc + (((int64_t)n * g.D + (pin ? p : 0)) * g.H) * (int64_t)g.W * b_ldx, pin ? b_xbytes : 0u);
      static_for<0, PXB>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        unsigned v = b_vx + (unsigned)k * b_xrow3;
        if constexpr (k * RG3 + RG3 > C::LR) { if (k * RG3 + b_rg >= C::LR) v = 0x80000000u; }
        rx[k] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, v, 0, 0));
      });
      return;
    }
    const bool pin = p >= 0 && p < g.D;
    const T* xplane = xsrc + (((int64_t)n * g.D + (pin ? p : 0)) * g.H) * (int64_t)g.W * ldsrc;
    const int ihb = h0 - C::PAD, iwb = w0 - C::PAD, t2 = opaque(tid >> 1);
#pragma unroll
    for (int j = 0; j < PX; j++) {
      int lr, lc; const bool in = xvox(t2, j, lr, lc);
      const int ih = ihb + lr, iw = iwb + lc;
      const bool ok = pin && x_exists && in && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      v4u v = *(const v4u*)(xplane + (ok ? (ih * g.W + iw) * ldsrc : 0));
      rx[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
  };
  auto issue_g = [&](int d, v4u* rg, bool live = true) {
    if constexpr (BUF) {
      const bool pin = live && d >= 0 && d < g.D;
      const auto rs = rsrc_of(gy + (((int64_t)n * g.D + (pin ? d : 0)) * g.H) * (int64_t)g.W * g.ldgy, pin ? b_gbytes : 0u);
      static_for<0, PG>([&](auto j_) { constexpr int j = decltype(j_)::value; rg[j] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, b_vg + (unsigned)j * b_grow4, 0, 0)); });
      return;
    }
    const T* gplane = gsrc + (((int64_t)n * g.D + d) * g.H) * (int64_t)g.W * g.ldgy;
#pragma unroll
    for (int j = 0; j < PG; j++) {
      const int oh = h0 + 4 * j + g_r, ow = w0 + g_c;
      const bool ok = g_exists && oh < g.H && ow < g.W;
      v4u v = *(const v4u*)(gplane + (ok ? (oh * g.W + ow) * g.ldgy : 0));
      rg[j] = ok ? v : (v4u){0, 0, 0, 0};
    }
  };
  auto commit_x = [&](int slot, const v4u* rx) {
    if constexpr (BUF) {
      T* dst = xs + slot * C::XPLANE;
      static_for<0, PXB>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        if constexpr (k == PXB - 1) *(v4u*)(dst + b_xl_last) = rx[k];
        else *(v4u*)(dst + b_xl0 + k * RG3 * C::LP * C::XC) = rx[k];
      });
      return;
    }
    const int t2 = opaque(tid >> 1);
    T* dst = xs + slot * C::XPLANE;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      int lr, lc;
      if (xvox(t2, j, lr, lc)) *(v4u*)(dst + (lr * C::LP + lc) * C::XC + piece * 8) = rx[j];
    }
  };
  auto commit_g = [&](const v4u* rg) {
#pragma unroll
    for (int j = 0; j < PG; j++) *(v4u*)(gs + ((4 * j + g_r) * C::GP + g_c) * C::GC + piece * 8) = rg[j];
  };
  auto slot_of = [](int p) { return (p + 3) % 3; };     // p >= -1

  // gy plane d against x planes d-1, d, d+1 (kd = 0, 1, 2): rows r = 3 kd + kh; a row's next column is requested right after its own
  // three MFMAs (24 MFMAs of other rows before its first use), the gy fragment of column cc+1 at the start of step cc (ring of KS + 1)
  auto sweep = [&](int d) {
    const T* xb[3];
#pragma unroll
    for (int kd = 0; kd < 3; kd++) xb[kd] = xl + slot_of(d + kd - 1) * C::XPLANE;
    Frag8<T> X[9], G[KS + 1];
    auto ldx = [&](auto r_, auto cc_) {
      constexpr int r = decltype(r_)::value, cc = decltype(cc_)::value;
      if constexpr (cc < C::STEPS) X[r] = tr_pair<8 * C::LP * C::XC, T>(xb[r / 3] + ((r % 3) * C::LP + cc) * C::XC);
    };
    auto ldg = [&](auto gc_) {
      constexpr int gc = decltype(gc_)::value;
      if constexpr (gc < C::CW) G[gc % (KS + 1)] = tr_pair<8 * C::GP * C::GC, T>(gb + gc * C::GC);
    };
    static_for<0, 9>([&](auto r_) { ldx(r_, std::integral_constant<int, 0>{}); });
    ldg(std::integral_constant<int, 0>{});
    static_for<0, C::STEPS>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      constexpr std::integral_constant<int, cc + 1> nx{};
      ldg(nx);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 9>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        static_for<0, KS>([&](auto kw_) {
          constexpr int kw = decltype(kw_)::value, gc = cc - kw;
          if constexpr (gc >= 0 && gc < C::CW) acc[r * 3 + kw] = mma16(X[r], G[gc % (KS + 1)], acc[r * 3 + kw]);
        });
        __builtin_amdgcn_sched_barrier(0);
        ldx(r_, nx);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  const bool fast = (g.ldx % 8 == 0) && (g.ldgy % 8 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)gy & 15) == 0) &&
                    (!g.x2 || ((g.csplit % 8 == 0) && (g.ldx2 % 8 == 0) && (((uintptr_t)g.x2 & 15) == 0)));
  // (the launcher only selects this kernel for aligned operands; an unaligned call leaves zero slabs)
  // prologue: x planes z0-1 and z0 into the ring, then x plane z0+1 + gy plane z0 in flight
  // Two planes ahead (one block per CU = one wave per SIMD: nothing else hides the HBM latency, and one plane of loads in flight
  // per CU left the fabric at 2.5 TB/s): iteration d commits set d & 1 (x plane d+1, gy plane d; issued at d-2) and refills it
  // with x plane d+3 / gy plane d+2.  The depth loop is unrolled by two so that the set index is static.
  const bool nostage = g.dbg & 1, nosweep = g.dbg & 2;               // experiments only (env DP_DBG)
  if (fast && z0 < z1) {
    issue_x(z0 - 1, rxs[0]); commit_x(slot_of(z0 - 1), rxs[0]);
    issue_x(z0, rxs[0]); commit_x(slot_of(z0), rxs[0]);
    issue_x(z0 + 1, rxs[0]); issue_g(z0, rgs[0]);
    if constexpr (BUF) { issue_x(z0 + 2, rxs[1], z0 + 1 < z1); issue_g(z0 + 1, rgs[1], z0 + 1 < z1); }
    else if (z0 + 1 < z1) { issue_x(z0 + 2, rxs[1]); issue_g(z0 + 1, rgs[1]); }
  }
  auto step = [&](int d, v4u* rx, v4u* rg) {
    lds_barrier();                                                  // sweep d-1 is over: slot of plane d-2 and the gy tile are free
    if constexpr (BUF) {
      // every step commits and issues UNCONDITIONALLY (a prefetch beyond the segment, or DP_DBG's "no staging", is a descriptor of zero bytes):
      // with the refill under a runtime condition the compiler must assume that the OTHER register set's 20 loads may not be in flight and
      // waits for the newest load at every commit -- one plane of prefetch distance instead of two
      commit_x(slot_of(d + 1), rx); commit_g(rg);
      issue_x(d + 3, rx, d + 2 < z1 && !nostage); issue_g(d + 2, rg, d + 2 < z1 && !nostage);
    } else {
      if (!nostage) { commit_x(slot_of(d + 1), rx); commit_g(rg); }
      if (d + 2 < z1 && !nostage) { issue_x(d + 3, rx); issue_g(d + 2, rg); }
    }
    lds_barrier();
    if (!nosweep) sweep(d);
  };
  for (int d = z0; fast && d < z1; d += 2) {
    step(d, rxs[0], rgs[0]);
    if (d + 1 < z1) step(d + 1, rxs[1], rgs[1]);
  }
  // The four waves hold partial sums of the SAME 27 x 16 x 16 block (they walked different gy columns): add them up through LDS
  // (the ring is free now), then one non-atomic 27 KB slab per block -- k_wgrad_hk3_finish adds the slabs of all blocks.  (Atomics
  // into the 6 912-element tap-major scratch: 1 024 adds per address, measured 73 of the kernel's 110 us.)
  lds_barrier();
  float* red = (float*)smem_raw;                                   // [4 waves][27][16 ci][16 co]
#pragma unroll
  for (int t = 0; t < 27; t++)
#pragma unroll
    for (int e = 0; e < 4; e++) red[(wv * 27 + t) * 256 + (4 * q + e) * 16 + (lane & 15)] = acc[t][e];
  lds_barrier();
  float* slab = dwt + (int64_t)blockIdx.x * (27 * 256);
#pragma unroll
  for (int j = 0; j < 27; j++) {
    const int i = j * 256 + tid;
    slab[i] = (red[i] + red[27 * 256 + i]) + (red[2 * 27 * 256 + i] + red[3 * 27 * 256 + i]);
  }
}

// dW[co][ci][tap] = sum over the blocks of (mt, nt) of slab[tap][ci % 16][co % 16]; the slabs are handed back zeroed (scratch contract).
// Block = 64 consecutive slab elements x 4 groups of slabs (8 independent loads in flight per thread), LDS add of the four groups.
__global__ void __launch_bounds__(256) k_wgrad_hk3_finish(float* __restrict__ slabs, float* __restrict__ dw, int Cin, int Cout, int MT, int NTn, int nseg,
                                                          int nsp, int64_t s_co, int64_t s_ci, int64_t s_tap, int rezero) {
  __shared__ float red[256];
  int b = blockIdx.x;
  const int chunk = b % 108; b /= 108; const int mt = b % MT; const int nt = b / MT;        // 108 x 64 = 27 x 256 elements
  const int el = chunk * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int nslab = nsp * nseg, per = (nslab + 3) / 4, j0 = part * per, j1 = min(nslab, j0 + per);
  // slab j of this (mt, nt): j = sp * nseg + sg  ->  block index ((sp * NTn + nt) * MT + mt) * nseg + sg
  auto addr = [&](int j) { const int sp = j / nseg, sg = j - sp * nseg; return slabs + ((((int64_t)sp * NTn + nt) * MT + mt) * nseg + sg) * (27 * 256) + el; };
  float sacc = 0.f;
  int j = j0;
  for (; j + 8 <= j1; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = *addr(j + u);
#pragma unroll
    for (int u = 0; u < 8; u++) { sacc += v[u]; if (rezero) *addr(j + u) = 0.f; }
  }
  for (; j < j1; j++) { sacc += *addr(j); if (rezero) *addr(j) = 0.f; }
  red[threadIdx.x] = sacc;
  __syncthreads();
  if (part == 0) {
    const float t = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
    const int tap = el >> 8, ci = mt * 16 + ((el >> 4) & 15), co = nt * 16 + (el & 15);
    if (ci < Cin && co < Cout) dw[co * s_co + ci * s_ci + tap * s_tap] = t;
  }
}

template <typename T, bool BUF>
int launch_hk3_impl(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s);
template <typename T>
int launch_hk3(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s) {
  static const int buf_env = [] { const char* e = getenv("DP_HK_BUF"); return e ? atoi(e) : 1; }();
  const int64_t xplane = (int64_t)g.H * g.W * (g.ldx > g.ldx2 ? g.ldx : g.ldx2) * 2, gplane = (int64_t)g.H * g.W * g.ldgy * 2;
  const bool buf = buf_env && (!g.x2 || g.csplit % 16 == 0) && xplane < (1LL << 30) && gplane < (1LL << 30);
  return buf ? launch_hk3_impl<T, true>(x, gy, ws, g, s) : launch_hk3_impl<T, false>(x, gy, ws, g, s);
}
template <typename T, bool BUF>
int launch_hk3_impl(const void* x, const void* gy, float* ws, WgHkGeom g, hipStream_t s) {
  using C = Hk3Cfg<T>;
  auto kern = k_wgrad_hk3<T, BUF>;
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    if (e != hipSuccess) { dp_set_error("wgrad_hk3: cannot raise dynamic LDS to %zu: %s", C::SMEM, hipGetErrorString(e)); return 1; }
    raised = true;
  }
  g.tiles_h = cdiv(g.H, C::TH); g.tiles_w = cdiv(g.W, C::TW);
  g.MT = cdiv(g.Cin, C::XC); g.NTn = cdiv(g.Cout, C::GC);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  // depth segments: one resident wave of blocks (one block per CU); a segment re-reads two x planes, so no more of them than needed
  const int64_t cols = (int64_t)g.N * g.tiles_h * g.tiles_w * g.MT * g.NTn;
  int nseg = (int)(ncu / cols); if (nseg < 1) nseg = 1; if (nseg > g.D) nseg = g.D;
  const int DS = cdiv(g.D, nseg); nseg = cdiv(g.D, DS);
  g.ydim = nseg;
  if (cols * nseg > HK3_MAX_BLOCKS) return -1;                       // more slabs than the scratch holds: the caller takes the K-along-W kernel
  hipLaunchKernelGGL(kern, dim3((unsigned)(cols * nseg)), dim3(256), C::SMEM, s, (const T*)x, (const T*)gy, ws, g);
  hipLaunchKernelGGL(k_wgrad_hk3_finish, dim3(108 * g.MT * g.NTn), dim3(256), 0, s, ws, g.dw, g.Cin, g.Cout, g.MT, g.NTn, nseg,
                     g.N * g.tiles_h * g.tiles_w, g.s_co, g.s_ci, g.s_tap, g.rezero);
  return 0;
}

}  // namespace

bool wgrad_hk_applicable(int Cout, int k, int H, int W, int dtype) {
  static const bool off = getenv("DP_NO_HK") != nullptr;
  static const bool off3 = getenv("DP_NO_HK3") != nullptr;
  (void)Cout;
  // 7^3 on planes down to 24 x 24 (one zero-padded 32 x 32 tile; round 5, the 24^3 level of the 96^3 crop): 64 -> 64 at 4 x 24^3 0.241 -> 0.201 ms,
  // 128 -> 64 0.500 -> 0.396 (644 -> 771 / 621 -> 784 TFLOP/s); 16 x 16 and 12 x 12 planes lose on it (a quarter / a seventh of the tile is real)
  static const int minp = [] { const char* e = getenv("DP_HK_MINPLANE"); return e ? atoi(e) : 24; }();
  return !off && dtype != DP_F32 && (k == 7 || (k == 3 && !off3)) && H >= (k == 7 ? minp : 32) && W >= (k == 7 ? minp : 32);
}

int64_t wgrad_hk_ws_elems(int Cin, int Cout, int k) {
  (void)Cin; (void)Cout;
  return k == 3 ? (int64_t)HK3_MAX_BLOCKS * 27 * 256 : 0;
}

int wgrad_hk_launch(const void* x, const void* gy, float* ws, const WgHkGeom& g, int k, int dtype, hipStream_t s, int* finished) {
  *finished = 0;
  if (k == 3) {
    const int rc = dtype == DP_BF16 ? launch_hk3<bf16_t>(x, gy, ws, g, s) : launch_hk3<f16_t>(x, gy, ws, g, s);
    if (rc == 0) *finished = 1;
    return rc;
  }
  if (k != 7) { dp_set_error("wgrad_hk: kernel size %d not built", k); return 1; }
  return dtype == DP_BF16 ? launch_hk<bf16_t, 7>(x, gy, ws, g, s) : launch_hk<f16_t, 7>(x, gy, ws, g, s);
}
