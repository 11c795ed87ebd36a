// Multi-tensor weight re-packing: ONE launch rebuilds every kernel-layout copy (bf16/fp16/fp32 packed weights) of the
// parameters an optimizer step just changed.  The layouts are exactly those of the single-tensor pack entry points
// (dp_pack_conv_weight, dp_pack_conv_weight_tiled, dp_cast + dp_copy_rows for Linear / ConvTranspose matrices); the
// per-tensor launches cost ~600 launches per training step once the packs are (correctly) invalidated by every
// optimizer.step() (network_trainer.py:185-213 calls it once per iteration).
#include "common.h"

#define STREAM ((hipStream_t)stream)
#define PACK_CHUNK 8192          // destination elements per block (matrix-transpose tiles: 64 rows x 128 columns)

enum { PK_CAST = 0, PK_MAT = 1, PK_MAT_T = 2, PK_CONV = 3, PK_CONV_TILED = 4, PK_TCONV = 5, PK_CONV_CC16 = 6 };

struct PackDesc { const float* src; void* dst; int64_t kind, a, b, c, d, e; };
// x3 packs (fp32x3 mode, x3.hip): kind = base kind | pattern << 8 | cp << 16 with cp > 0.  The contraction axis of the packed copy
// (input channels of a convolution, columns of a Linear matrix) is then 3 blocks of cp "virtual" channels over the SAME cp (zero
// padded) real channels; block p holds bf16(w) when pattern bit p is 0 and bf16(w - bf16(w)) when it is 1 ([w_hi | w_hi | w_lo] = 0b100
// against activations split [x_hi | x_lo | x_hi] = 0b010).
__device__ __forceinline__ float x3_part(float v, int lo) { const float h = bf2f(f2bf(v)); return lo ? v - h : h; }

// Tiled convolution layouts (PK_CONV_TILED, PK_CONV_CC16; round 4): a chunk = FOUR destination columns (output channels of the
// packed copy) x ONE 16-channel destination chunk x ALL taps.  Its source is 4 (16 when transposed) contiguous runs of the fp32
// parameter -- 16 x taps (4 x taps) floats each -- which go through an LDS image [4 cols][taps][16 ch] and leave as whole 32- / 128-byte
// pieces of the destination tiles.  The element-per-thread walk it replaces fetched 924 MB from HBM to write 223 MB (every 28-byte
// run of seven kw taps pulled two sectors, and the (kd, kh) passes over one (co, ci) pair were a cache lifetime apart).
__host__ __device__ inline int pk_conv_cols(int kind, int Cout, int KS, int NPAIR) {      // destination columns incl. padding
  return kind == PK_CONV_CC16 ? 16 : ((Cout * NPAIR + 31) / 32) * 32 / NPAIR;
}
extern "C" int64_t dp_pack_chunks(int64_t kind_, int64_t a, int64_t b, int64_t c, int64_t d, int64_t e, int64_t dst_elems) {
  const int kind = (int)(kind_ & 0xff), x3cp = (int)(kind_ >> 16);
  (void)e;
  if (kind == PK_MAT_T) return ((a + 63) / 64) * ((c + 127) / 128);
  if (kind == PK_CONV_TILED || kind == PK_CONV_CC16) {
    const int nch = x3cp ? 3 * x3cp / 16 : (int)((b + 15) / 16);
    return (int64_t)(pk_conv_cols(kind, (int)a, (int)c, (int)d) / 4) * nch;
  }
  return (dst_elems + PACK_CHUNK - 1) / PACK_CHUNK;
}

template <typename T>
__device__ __forceinline__ void pack_conv_lds(const PackDesc& P, unsigned ck, T* img) {
  const float* __restrict__ w = P.src;
  T* __restrict__ dst = (T*)P.dst;
  const int kind = (int)(P.kind & 0xff), x3pat = (int)((P.kind >> 8) & 0xff), x3cp = (int)(P.kind >> 16);
  const int Cout = (int)P.a, Cin = (int)P.b, KS = (int)P.c, NPAIR = kind == PK_CONV_TILED ? (int)P.d : 1, tf = (int)P.e;
  const int taps = KS * KS * KS, NCH = x3cp ? 3 * x3cp / 16 : (Cin + 15) / 16;
  const int ch = (int)(ck % (unsigned)NCH), co0 = (int)(ck / (unsigned)NCH) * 4;
  int ci0 = ch * 16, lo = 0;
  if (x3cp) { const int part = ci0 / x3cp; ci0 -= part * x3cp; lo = (x3pat >> part) & 1; }
  const int tid = threadIdx.x;
  // ---- source -> img[cc][tap][c] (element (cc * taps + tap) * 16 + c); the runs are read four floats at a time (16-byte requests at
  // 4-byte alignment; 16 x taps and 4 x taps are multiples of 4 for taps = 27, 343)
  // All 16 x 4 x taps floats of the chunk are (16 or 4) x taps / 4 four-float units; a thread requests its units in batches of 11
  // BEFORE it touches LDS (one memory round trip per batch: a loop per run was 4-16 dependent round trips per block, 19 us each).
  // The LDS image keeps the SOURCE order -- run-major, tap fastest: consecutive lanes write consecutive 8 bytes (an image in
  // destination order put the lanes of a wave 128 bytes apart: two banks) -- and the transposition happens on the read side.
  typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
  const int rl = (tf ? 4 : 16) * taps;                         // floats per run
  const int upr = rl / 4;                                      // units per run
  const int nrun = tf ? 16 : 4, nunits = nrun * upr;
  constexpr int UB = 11;
  for (int u0 = 0; u0 < nunits; u0 += 256 * UB) {
    v4f_u q4[UB];
#pragma unroll
    for (int b = 0; b < UB; b++) {
      const int u = u0 + b * 256 + tid;
      const int rn = u / upr, i = (u - rn * upr) * 4;
      q4[b] = (v4f_u){0.f, 0.f, 0.f, 0.f};
      if (u < nunits) {
        // run rn: tf == 0: column co0 + rn, floats [ci0 .. ci0 + 15][tap]; tf == 1: channel ci0 + rn, floats [co0 .. co0 + 3][tap]
        const int64_t base = tf ? ((int64_t)(ci0 + rn) * Cout + co0) * taps : ((int64_t)(co0 + rn) * Cin + ci0) * taps;
        const int nv = tf ? ((ci0 + rn < Cin) ? min(4, Cout - co0) * taps : 0) : ((co0 + rn < Cout) ? min(16, Cin - ci0) * taps : 0);
        if (i + 4 <= nv) q4[b] = *(const v4f_u*)(w + base + i);
        else for (int j = 0; j < 4; j++) if (i + j < nv) q4[b][j] = w[base + i + j];
      }
    }
#pragma unroll
    for (int b = 0; b < UB; b++) {
      const int u = u0 + b * 256 + tid;
      if (u >= nunits) continue;
      T h[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { float v = q4[b][j]; if (x3cp) v = x3_part(v, lo); st_f(&h[j], v); }
      *(uint2*)(img + (int64_t)u * 4) = *(const uint2*)h;       // image element (run * rl + i): 4 two-byte elements, 8-byte aligned
    }
  }
  __syncthreads();
  // element (column cc, tap, channel c) of the image
  auto at = [&](int cc, int tap, int c) -> unsigned {
    const int idx = tf ? (c * rl + cc * taps + (taps - 1 - tap)) : (cc * rl + c * taps + tap);
    return (unsigned)*(const unsigned short*)(img + idx);
  };
  auto piece = [&](int cc, int tap, int hv) -> v4u {           // channels hv * 8 .. + 7 of (cc, tap)
    v4u v;
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = at(cc, tap, hv * 8 + 2 * k) | (at(cc, tap, hv * 8 + 2 * k + 1) << 16);
    return v;
  };
  const v4u zero = (v4u){0, 0, 0, 0};
  if (kind == PK_CONV_TILED) {
    // dst[kd][jh][kw][chunk][ntile][col 32][ci 16]; NPAIR == 2: col = (kh & 1) * 16 + co % 16, ntile = co / 16 (kh = KS: zero padding)
    const int JH = NPAIR == 2 ? (KS + 1) / 2 : KS, NTT = (Cout * NPAIR + 31) / 32, KHS = NPAIR == 2 ? 2 * JH : KS;
    const int pieces = KS * KHS * KS * 8;                      // per (kd, kh slot, kw): 4 columns x 2 halves of 16 bytes
    for (int i = tid; i < pieces; i += 256) {
      const int hv = i & 1, cc = (i >> 1) & 3; int t = i >> 3;
      const int kw = t % KS; t /= KS; const int khs = t % KHS, kd = t / KHS;
      const int co = co0 + cc;
      int jh, col, nt;
      if (NPAIR == 2) { jh = khs >> 1; col = (khs & 1) * 16 + (co & 15); nt = co >> 4; } else { jh = khs; col = co & 31; nt = co >> 5; }
      const int64_t di = ((((int64_t)((kd * JH + jh) * KS + kw) * NCH + ch) * NTT + nt) << 9) + (col << 4) + hv * 8;
      *(v4u*)(dst + di) = khs < KS ? piece(cc, (kd * KS + khs) * KS + kw, hv) : zero;
    }
  } else if (KS == 7) {
    // dst[kd][chunk][slot 25][co 16][k 32]: k < 16 tap t = 2 slot, k >= 16 tap 2 slot + 1 (t = 7 kh + kw; t = 49: zero)
    const int pieces = 7 * 50 * 8;
    for (int i = tid; i < pieces; i += 256) {
      const int hv = i & 1, cc = (i >> 1) & 3; int t = i >> 3;
      const int tt = t % 50, kd = t / 50;
      const int64_t di = ((((int64_t)(kd * NCH + ch) * 25 + (tt >> 1))) << 9) + ((co0 + cc) << 5) + ((tt & 1) << 4) + hv * 8;
      *(v4u*)(dst + di) = tt < 49 ? piece(cc, kd * 49 + tt, hv) : zero;
    }
  } else {
    // dst[kd][chunk][kwp][kh][co 16][k 32]: k < 16 tap kw = 2 kwp, k >= 16 kw = 2 kwp + 1 (kw = KS: zero)
    const int KWP = (KS + 1) / 2, pieces = KS * KWP * 2 * KS * 8;
    for (int i = tid; i < pieces; i += 256) {
      const int hv = i & 1, cc = (i >> 1) & 3; int t = i >> 3;
      const int kh = t % KS; t /= KS; const int kw = t % (2 * KWP), kd = t / (2 * KWP);
      const int64_t di = ((((int64_t)(kd * NCH + ch) * KWP + (kw >> 1)) * KS + kh) << 9) + ((co0 + cc) << 5) + ((kw & 1) << 4) + hv * 8;
      *(v4u*)(dst + di) = kw < KS ? piece(cc, (kd * KS + kh) * KS + kw, hv) : zero;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_pack_multi(const PackDesc* __restrict__ tab, const int* __restrict__ chunk_t,
                                                    const int* __restrict__ chunk_i) {
  const PackDesc P = tab[chunk_t[blockIdx.x]];
  const unsigned ck = (unsigned)chunk_i[blockIdx.x];
  const float* __restrict__ w = P.src;
  T* __restrict__ dst = (T*)P.dst;
  const int kind = (int)(P.kind & 0xff), x3pat = (int)((P.kind >> 8) & 0xff), x3cp = (int)(P.kind >> 16);
  extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
  if ((kind == PK_CONV_TILED || kind == PK_CONV_CC16) && sizeof(T) == 2) {
    if constexpr (sizeof(T) == 2) pack_conv_lds<T>(P, ck, (T*)pk_smem);
    return;
  }
  if (kind == PK_MAT_T) {
    // dst[r][c] (rows a, valid columns b, pitch c) = src[c][r] (src is [b][a]); tile = 64 dst rows x 128 dst columns
    // the tile is transposed on its way INTO LDS (as T, row pitch 130 elements = 65 words: conflict-free 2-byte writes), so that a
    // destination row leaves as 16-byte pieces (the 2-byte global stores of round 2-3 ran at 1.7 TB/s)
    constexpr int TP = 130;
    T* tile = (T*)pk_smem;                               // [64 dst rows][TP]
    const unsigned rows = (unsigned)P.a, cols = (unsigned)P.b, pitch = (unsigned)P.c;
    const unsigned tc = (pitch + 127) / 128;
    const unsigned r0 = (ck / tc) * 64, c0 = (ck % tc) * 128;
    // 128 source rows x 64 floats = 2048 four-float units, 8 per thread, all requested before the first LDS write
    typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
    v4f_u q4[8]; unsigned lo8 = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned u = b * 256 + threadIdx.x, sr = u >> 4, sc = (u & 15) * 4;     // src row = dst col, src cols sc .. sc + 3 = dst rows
      unsigned cv = c0 + sr, part = 0;
      if (x3cp) { part = cv / (unsigned)x3cp; cv -= part * (unsigned)x3cp; }
      q4[b] = (v4f_u){0.f, 0.f, 0.f, 0.f};
      if (cv < cols && part < 3) {
        const float* src = w + (int64_t)cv * rows + r0 + sc;
        if (r0 + sc + 4 <= rows) q4[b] = *(const v4f_u*)src;
        else for (int j = 0; j < 4; j++) if (r0 + sc + j < rows) q4[b][j] = src[j];
      }
      if (x3cp && ((x3pat >> part) & 1)) lo8 |= 1u << b;
    }
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned u = b * 256 + threadIdx.x, sr = u >> 4, sc = (u & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float v = q4[b][j];
        if (x3cp) v = x3_part(v, (lo8 >> b) & 1);
        st_f(tile + (sc + j) * TP + sr, v);
      }
    }
    __syncthreads();
    const bool vec = sizeof(T) == 2 && (pitch & 7) == 0 && (((uintptr_t)dst) & 15) == 0;
    if (vec) {
      for (unsigned i = threadIdx.x; i < 64 * 16; i += 256) {
        const unsigned dr = i >> 4, dc = (i & 15) * 8;
        if (r0 + dr < rows && c0 + dc < pitch) {
          union { v4u raw; unsigned short e[8]; unsigned u[4]; } pk;
          const unsigned* src32 = (const unsigned*)(tile + dr * TP + dc);        // (TP and dc even: 4-byte aligned)
          pk.u[0] = src32[0]; pk.u[1] = src32[1]; pk.u[2] = src32[2]; pk.u[3] = src32[3];
          *(v4u*)(dst + (int64_t)(r0 + dr) * pitch + c0 + dc) = pk.raw;
        }
      }
    } else {
      for (unsigned i = threadIdx.x; i < 64 * 128; i += 256) {
        const unsigned dr = i >> 7, dc = i & 127;
        if (r0 + dr < rows && c0 + dc < pitch) dst[(int64_t)(r0 + dr) * pitch + c0 + dc] = tile[dr * TP + dc];
      }
    }
    return;
  }
  const unsigned base = ck * PACK_CHUNK;
  if (kind == PK_CAST) {
    const unsigned n = (unsigned)P.a;
    if (base + PACK_CHUNK <= n && (((uintptr_t)w | (uintptr_t)dst) & 15) == 0) {
#pragma unroll
      for (int j = 0; j < PACK_CHUNK / 1024; j++) {
        const unsigned i = base + j * 1024 + threadIdx.x * 4;
        const v4f v = *(const v4f*)(w + i);
        st_f(dst + i, v[0]); st_f(dst + i + 1, v[1]); st_f(dst + i + 2, v[2]); st_f(dst + i + 3, v[3]);
      }
    } else {
      for (unsigned i = base + threadIdx.x; i < min(n, base + PACK_CHUNK); i += 256) st_f(dst + i, w[i]);
    }
    return;
  }
  unsigned total;
  if (kind == PK_MAT) total = (unsigned)(P.a * P.c);
  else if (kind == PK_CONV) total = (unsigned)((P.d == 0 ? P.a : P.b) * P.c * ((( P.d == 0 ? P.b : P.a) + 7) & ~7));
  else if (kind == PK_CONV_TILED) {
    const int KS = (int)P.c, NP = (int)P.d;
    total = (unsigned)(KS * (NP == 2 ? (KS + 1) / 2 : KS) * KS * ((x3cp ? 3 * x3cp : (P.b + 15)) / 16) * ((P.a * NP + 31) / 32) * 512);
  } else if (kind == PK_CONV_CC16) {
    const int KS = (int)P.c;
    total = (unsigned)(KS * ((x3cp ? 3 * x3cp : (P.b + 15)) / 16) * (KS == 7 ? 25 : ((KS + 1) / 2) * KS) * 512);
  } else total = (unsigned)(P.d ? P.a * P.c : 8 * P.b * P.c);
  const unsigned end = min(total, base + PACK_CHUNK);
  for (unsigned i = base + threadIdx.x; i < end; i += 256) {
    float v = 0.f;
    int part = 0;
    unsigned di = i;                 // destination element (the tiled convolution layouts permute the thread order)
    if (kind == PK_MAT) {
      const unsigned cols = (unsigned)P.b, pitch = (unsigned)P.c, r = i / pitch; unsigned c = i - r * pitch;
      if (x3cp) { part = (int)(c / (unsigned)x3cp); c -= (unsigned)part * (unsigned)x3cp; }
      if (c < cols) v = w[(int64_t)r * cols + c];
    } else if (kind == PK_CONV) {
      // mode 0: dst[co][t][ciP] = w[co][ci][t]; mode 1: dst[ci][t][coP]; mode 2: dst[ci][T-1-t][coP]   (k_pack_w)
      const unsigned Cout = (unsigned)P.a, Cin = (unsigned)P.b, taps = (unsigned)P.c; const int mode = (int)P.d;
      const unsigned inner = mode == 0 ? Cin : Cout, innerP = (inner + 7) & ~7u;
      const unsigned c = i % innerP, rt = i / innerP, t = rt % taps, row = rt / taps;
      if (c < inner) {
        const unsigned co = mode == 0 ? row : c, ci = mode == 0 ? c : row, ts = mode == 2 ? taps - 1 - t : t;
        v = w[((int64_t)co * Cin + ci) * taps + ts];
      }
    } else if (kind == PK_CONV_TILED) {
      // dst[kd][jh][kw][chunk][ntile][col 32][ci 16]   (k_pack_w_tiled)
      const int Cout = (int)P.a, Cin = (int)P.b, KS = (int)P.c, NPAIR = (int)P.d, tf = (int)P.e;
      const int JH = NPAIR == 2 ? (KS + 1) / 2 : KS, NCH = x3cp ? 3 * x3cp / 16 : (Cin + 15) / 16, NTT = (Cout * NPAIR + 31) / 32, taps = KS * KS * KS;
      // thread order (kd, jh, chunk, ntile | col, ci | kw): consecutive lanes walk the KS kw taps of one (co, ci) pair, which are
      // consecutive floats of the source (7 lanes per 28-byte run instead of 64 lanes on 64 different sectors: the 4-byte gathers
      // 1372 bytes apart made this the slowest HBM kernel of the step, 0.75 TB/s); a bijection of the destination indices
      unsigned t = i;
      const int kw = (int)(t % KS); t /= KS; const int c = (int)(t & 15), col = (int)((t >> 4) & 31); t >>= 9;
      const int nt = (int)(t % NTT); t /= NTT; const int ch = (int)(t % NCH); t /= NCH;
      const int jh = (int)(t % JH), kd = (int)(t / JH);
      di = ((((unsigned)((kd * JH + jh) * KS + kw) * NCH + ch) * NTT + nt) << 9) + (col << 4) + c;
      int kh, co;
      if (NPAIR == 2) { const int s = col >> 4; kh = 2 * jh + s; co = nt * 16 + (col & 15); }
      else { kh = jh; co = nt * 32 + col; }
      int ci = ch * 16 + c;
      if (x3cp) { part = ci / x3cp; ci -= part * x3cp; }
      if (kh < KS && co < Cout && ci < Cin) {
        const int tap = (kd * KS + kh) * KS + kw;
        v = tf ? w[((int64_t)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * Cin + ci) * taps + tap];
      }
    } else if (kind == PK_CONV_CC16) {
      // KS = 3: dst[kd][chunk][kwp][kh][co 16][k 32]; KS = 7: dst[kd][chunk][slot 25][co 16][k 32], slot j = taps 2j | 2j + 1 in the
      // linear order t = 7 kh + kw   (k_pack_w_cc16)
      const int Cout = (int)P.a, Cin = (int)P.b, KS = (int)P.c, tf = (int)P.e;
      const int KWP = (KS + 1) / 2, NCH = x3cp ? 3 * x3cp / 16 : (Cin + 15) / 16, taps = KS * KS * KS;
      unsigned t = i;
      int kh, kw, c16, co, ch, kd;
      if (KS == 7) {
        // thread order (kd, chunk | co, ci | t 0..49): consecutive lanes walk the 49 (kh, kw) taps of one (co, ci, kd), 196 consecutive
        // bytes of the source
        const int tt = (int)(t % 50); t /= 50; c16 = (int)(t & 15); co = (int)((t >> 4) & 15); t >>= 8;
        ch = (int)(t % NCH); kd = (int)(t / NCH);
        kh = tt / 7; kw = tt >= 49 ? KS : tt % 7;
        di = ((((unsigned)(kd * NCH + ch) * 25 + (tt >> 1))) << 9) + (co << 5) + ((tt & 1) << 4) + c16;
      } else {
        // thread order (kd, chunk, kh | co, ci | kw): see PK_CONV_TILED
        kw = (int)(t % (2 * KWP)); t /= 2 * KWP; c16 = (int)(t & 15); co = (int)((t >> 4) & 15); t >>= 8;
        kh = (int)(t % KS); t /= KS; ch = (int)(t % NCH); kd = (int)(t / NCH);
        const int kwp = kw >> 1, k = ((kw & 1) << 4) + c16;
        di = ((((unsigned)(kd * NCH + ch) * KWP + kwp) * KS + kh) << 9) + (co << 5) + k;
      }
      int ci = ch * 16 + c16;
      if (x3cp) { part = ci / x3cp; ci -= part * x3cp; }
      if (kw < KS && co < Cout && ci < Cin) {
        const int tap = (kd * KS + kh) * KS + kw;
        v = tf ? w[((int64_t)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * Cin + ci) * taps + tap];
      }
    } else {   // PK_TCONV: w[ci][co][abc]; d == 0: dst[(abc, co)][ciP]; d != 0: dst[ci][(abc, co)P]
      const unsigned cin = (unsigned)P.a, cout = (unsigned)P.b, pitch = (unsigned)P.c;
      const unsigned r = i / pitch, c = i - r * pitch;
      unsigned ci, q;
      bool ok;
      if (P.d == 0) { q = r; ci = c; ok = c < cin; } else { ci = r; q = c; ok = c < 8 * cout; }
      if (ok) { const unsigned abc = q / cout, co = q - abc * cout; v = w[((int64_t)ci * cout + co) * 8 + abc]; }
    }
    if (x3cp) v = x3_part(v, (x3pat >> part) & 1);
    st_f(dst + di, v);
  }
}

extern "C" int dp_pack_chunk(void) { return PACK_CHUNK; }
extern "C" int dp_pack_multi(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, int dtype, void* stream) {
  if (nchunks <= 0) return 0;
  // dynamic LDS: the transposing matrix tile (128 x 65 floats) or the convolution image (4 x 343 x 16 two-byte elements)
  constexpr size_t smem = 4 * 343 * 16 * 2 > 128 * 65 * 4 ? 4 * 343 * 16 * 2 : 128 * 65 * 4;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_multi<T>, dim3(nchunks), dim3(256), smem, STREAM, (const PackDesc*)table,
                                        (const int*)chunk_t, (const int*)chunk_i));
  DP_CHECK_LAUNCH("pack_multi"); return 0;
}
