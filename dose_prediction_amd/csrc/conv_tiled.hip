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

#define STREAM ((hipStream_t)stream)
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mma32(const Frag8<bf16_t>& a, const Frag8<bf16_t>& b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a.u), __builtin_bit_cast(v8bf, b.u), c, 0, 0, 0);
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
  return (k == 3 || k == 7) && stride == 1 && dil == 1 && pad == k / 2 && W >= 16 && Cin >= 8 && Cout >= 8;
}

extern "C" int dp_conv3d_tiled_weight_elems(int Cin, int Cout, int k, int stride, int pad, int dil, int W) {
  if (!tiled_applicable(Cin, Cout, k, stride, pad, dil, W)) return 0;
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  int JH = np == 2 ? (k + 1) / 2 : k;
  return k * JH * k * ((Cin + 15) / 16) * ((Cout * np + 31) / 32) * 512;
}
extern "C" int dp_pack_conv_weight_tiled(const float* w, void* dst, int Cout, int Cin, int k, int transposed_flipped, int dtype, void* stream) {
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  int JH = np == 2 ? (k + 1) / 2 : k;
  int64_t total = (int64_t)k * JH * k * ((Cin + 15) / 16) * ((Cout * np + 31) / 32) * 512;
  int g = (int)((total + 255) / 256); if (g > 8192) g = 8192;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_w_tiled<T>, dim3(g), dim3(256), 0, STREAM, w, (T*)dst, Cout, Cin, k, np, transposed_flipped));
  DP_CHECK_LAUNCH("pack_conv_weight_tiled"); return 0;
}

// ---------------------------------------------------------------------------------------------- the kernel
template <typename T, int KS, int NPAIR, int RW, int NT>
__global__ void __launch_bounds__(256) k_conv_tiled(const T* __restrict__ x, const T* __restrict__ wq, const float* __restrict__ bias,
                                                    T* __restrict__ y, TiledGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* slab = (T*)smem_raw;
  constexpr int PAD = KS / 2, JH = NPAIR == 2 ? (KS + 1) / 2 : KS, RWO = RW - (NPAIR - 1), NTAP = JH * KS, CK = 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int wc = wv % g.TWC, rg = wv / g.TWC;
  int b = blockIdx.x;
  const int tw = b % g.tiles_w; b /= g.tiles_w; const int th = b % g.tiles_h; b /= g.tiles_h; const int d = b % g.D; const int n = b / g.D;
  const int h0 = th * (g.TRG * RWO), w0 = tw * (g.TWC * 32);
  const int nt0 = blockIdx.y * NT;                 // first N tile of this block

  v16f acc[RW][NT];
#pragma unroll
  for (int i = 0; i < RW; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int pieces = g.LR * g.LP * 2;              // 8-channel pieces of the slab
  const int64_t wtap_stride = (int64_t)g.NCH * g.NTT * 512;      // elements between consecutive (kd,jh,kw) taps
  // lane-constant part of the A address: position (wc*32 + r), channel half hh, first row of this wave's row group
  const int a_lane = ((rg * RWO) * g.LP + wc * 32 + r) * CK + hh * 8;

  for (int kd = 0; kd < KS; kd++) {
    const int id = d + kd - PAD;
    if (id < 0 || id >= g.D) continue;             // block-uniform: the whole depth slice is zero padding
    for (int ch = 0; ch < g.NCH; ch++) {
      __syncthreads();
      for (int p = tid; p < pieces; p += 256) {
        int half = p & 1, v = p >> 1, lp = v % g.LP, lr = v / g.LP;
        int ih = h0 - PAD + lr, iw = w0 - PAD + lp, c = ch * CK + half * 8;
        int nv = g.Cin - c; nv = nv > 8 ? 8 : nv;
        bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && nv > 0;
        Frag8<T> f = ok ? frag_load(x + ((((int64_t)n * g.D + id) * g.H + ih) * g.W + iw) * g.ldx + c, nv) : frag_zero<T>();
        frag_st_lds(slab + (int64_t)v * CK + half * 8, f);
      }
      __syncthreads();
      const T* wbase = wq + ((int64_t)kd * NTAP * g.NCH + ch) * g.NTT * 512 + (int64_t)nt0 * 512 + r * 16 + hh * 8;
      Frag8<T> bcur[NT], bnext[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) bcur[j] = frag_load(wbase + j * 512, 8);
#pragma unroll 1
      for (int tt = 0; tt < NTAP; tt++) {
        const int jh = tt / KS, kw = tt - jh * KS;
        if (tt + 1 < NTAP) {
#pragma unroll
          for (int j = 0; j < NT; j++) bnext[j] = frag_load(wbase + (int64_t)(tt + 1) * wtap_stride + j * 512, 8);
        }
        const int a_tap = a_lane + ((NPAIR == 2 ? 2 * jh : jh) * g.LP + kw) * CK;
#pragma unroll
        for (int i = 0; i < RW; i++) {
          Frag8<T> fa = frag_ld_lds(slab + a_tap + i * g.LP * CK);
#pragma unroll
          for (int j = 0; j < NT; j++) acc[i][j] = mma32(fa, bcur[j], acc[i][j]);
        }
#pragma unroll
        for (int j = 0; j < NT; j++) bcur[j] = bnext[j];
      }
    }
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int hw0 = h0 + rg * RWO, wbase_o = w0 + wc * 32;
  if (NPAIR == 2) {
    const int co = lane & 15;
    const bool writer = (lane & 16) == 0 && co < g.Cout;
    const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < RWO; t++) {
      const int oh = hw0 + t;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        float z1 = __shfl_down(acc[t + 1 < RW ? t + 1 : t][0][e], 16, 64);    // odd tap of the pair: accumulated one row below
        float v = acc[t][0][e] + z1 + bv;
        int ow = wbase_o + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (writer && oh < g.H && ow < g.W) st_f(y + ((((int64_t)n * g.D + d) * g.H + oh) * g.W + ow) * g.ldy + co, v);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int co = (nt0 + j) * 32 + r;
      const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
#pragma unroll
      for (int i = 0; i < RW; i++) {
        const int oh = hw0 + i;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          int ow = wbase_o + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (co < g.Cout && oh < g.H && ow < g.W) st_f(y + ((((int64_t)n * g.D + d) * g.H + oh) * g.W + ow) * g.ldy + co, acc[i][j][e] + bv);
        }
      }
    }
  }
}

template <typename T, int KS, int NPAIR, int RW, int NT>
static int launch_tiled(const void* x, const void* wq, const float* bias, void* y, TiledGeom g, int ygrid, hipStream_t s) {
  size_t smem = (size_t)g.LR * g.LP * 16 * sizeof(T);
  auto kern = k_conv_tiled<T, KS, NPAIR, RW, NT>;
  if (smem > 160 * 1024) { dp_set_error("conv3d_tiled: slab %zu B exceeds LDS", smem); return 1; }
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { dp_set_error("conv3d_tiled: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e)); return 1; }
  }
  dim3 grid(g.N * g.D * g.tiles_h * g.tiles_w, ygrid);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const T*)x, (const T*)wq, bias, (T*)y, g);
  return 0;
}

// Tiled convolution with weights packed by dp_pack_conv_weight_tiled.  Same-size output ("same" padding).
extern "C" int dp_conv3d_tiled(const void* x, int ldx, const void* wq, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                               int Cin, int Cout, int k, int dtype, void* stream) {
  if (!tiled_applicable(Cin, Cout, k, 1, k / 2, 1, W)) DP_FAIL("conv3d_tiled: shape not supported");
  int rw, nt; int np = tiled_config(Cout, &rw, &nt);
  int rwo = rw - (np - 1), JH = np == 2 ? (k + 1) / 2 : k;
  TiledGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ldx = ldx; g.ldy = ldy;
  if (W > 64) { g.TWC = 4; g.TRG = 1; } else if (W > 32) { g.TWC = 2; g.TRG = 2; } else { g.TWC = 1; g.TRG = 4; }
  g.LR = g.TRG * rwo + (np - 1) + (np == 2 ? 2 * (JH - 1) : k - 1);
  g.LP = g.TWC * 32 + k - 1;
  g.NCH = (Cin + 15) / 16; g.NTT = (Cout * np + 31) / 32;
  g.tiles_h = cdiv(H, g.TRG * rwo); g.tiles_w = cdiv(W, g.TWC * 32);
  int ygrid = cdiv(g.NTT, nt);
  if ((int64_t)N * D * g.tiles_h * g.tiles_w > 2000000000LL) DP_FAIL("conv3d_tiled: grid too large");
  int rc = 0;
  hipStream_t s = STREAM;
#define GO(TT, KS_, NP, RW_, NT_) rc = launch_tiled<TT, KS_, NP, RW_, NT_>(x, wq, bias, y, g, ygrid, s)
#define BYCFG(TT, KS_) do { if (np == 2) GO(TT, KS_, 2, 9, 1); else if (nt == 1) GO(TT, KS_, 1, 8, 1); else GO(TT, KS_, 1, 4, 2); } while (0)
  if (dtype == DP_BF16) { if (k == 7) BYCFG(bf16_t, 7); else BYCFG(bf16_t, 3); }
  else if (dtype == DP_F32) { if (k == 7) BYCFG(float, 7); else BYCFG(float, 3); }
  else DP_FAIL("conv3d_tiled: bad dtype");
#undef BYCFG
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
