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

template <typename T>
__global__ void __launch_bounds__(256) k_pack_multi(const PackDesc* __restrict__ tab, const int* __restrict__ chunk_t,
                                                    const int* __restrict__ chunk_i) {
  const PackDesc P = tab[chunk_t[blockIdx.x]];
  const unsigned ck = (unsigned)chunk_i[blockIdx.x];
  const float* __restrict__ w = P.src;
  T* __restrict__ dst = (T*)P.dst;
  const int kind = (int)(P.kind & 0xff), x3pat = (int)((P.kind >> 8) & 0xff), x3cp = (int)(P.kind >> 16);
  if (kind == PK_MAT_T) {
    // dst[r][c] (rows a, valid columns b, pitch c) = src[c][r] (src is [b][a]); tile = 64 dst rows x 128 dst columns
    __shared__ float tile[128][65];
    const unsigned rows = (unsigned)P.a, cols = (unsigned)P.b, pitch = (unsigned)P.c;
    const unsigned tc = (pitch + 127) / 128;
    const unsigned r0 = (ck / tc) * 64, c0 = (ck % tc) * 128;
    for (unsigned i = threadIdx.x; i < 128 * 64; i += 256) {
      const unsigned sr = i >> 6, sc = i & 63;          // src row = dst col, src col = dst row
      float v = 0.f;
      unsigned cv = c0 + sr, part = 0;
      if (x3cp) { part = cv / (unsigned)x3cp; cv -= part * (unsigned)x3cp; }
      if (cv < cols && part < 3 && r0 + sc < rows) v = w[(int64_t)cv * rows + r0 + sc];
      if (x3cp) v = x3_part(v, (x3pat >> part) & 1);
      tile[sr][sc] = v;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 64 * 128; i += 256) {
      const unsigned dr = i >> 7, dc = i & 127;
      if (r0 + dr < rows && c0 + dc < pitch) st_f(dst + (int64_t)(r0 + dr) * pitch + c0 + dc, tile[dc][dr]);
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
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_multi<T>, dim3(nchunks), dim3(256), 0, STREAM, (const PackDesc*)table,
                                        (const int*)chunk_t, (const int*)chunk_i));
  DP_CHECK_LAUNCH("pack_multi"); return 0;
}
