// Generic implicit-GEMM 3-D convolution on MFMA (any kernel size / stride / padding / dilation, forward form and
// strided data-gradient form) and the generic weight-gradient kernel.  These are the always-correct paths used for
// the rare shapes (stride 2, dilated, tiny channel counts); the LDS-tiled kernels in conv_tiled.hip take the
// hot stride-1 3x3x3 / 7x7x7 layers.
//
// GEMM view (forward): M = output voxels, N = Cout, K = taps x CinP with k = tap*CinP + ci (CinP = roundup8(Cin)).
// A K-step of 32 is four 8-channel groups; lane (r = l&15, q = l>>4) owns group q of voxel row r, so its A fragment
// is ONE 16-byte load of 8 consecutive channels of one (shifted) input voxel, and its B fragment one 16-byte load of
// the packed weight row [co][tap][ci..ci+7].
#include "common.h"

#define STREAM ((hipStream_t)stream)

// ------------------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void k_pack_w(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int taps, int mode) {
  // mode 0: dst[co][t][ciP]   = w[co][ci][t]
  // mode 1: dst[ci][t][coP]   = w[co][ci][t]
  // mode 2: dst[ci][T-1-t][coP] = w[co][ci][t]
  int rows = mode == 0 ? Cout : Cin, inner = mode == 0 ? Cin : Cout, innerP = (inner + 7) & ~7;
  int64_t total = (int64_t)rows * taps * innerP;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % innerP); int64_t rt = i / innerP; int t = (int)(rt % taps), row = (int)(rt / taps);
    float v = 0.f;
    if (c < inner) {
      int co = mode == 0 ? row : c, ci = mode == 0 ? c : row, ts = mode == 2 ? taps - 1 - t : t;
      v = w[((int64_t)co * Cin + ci) * taps + ts];
    }
    st_f(dst + i, v);
  }
}
extern "C" int dp_pack_conv_weight(const float* w, void* dst, int Cout, int Cin, int taps, int mode, int dtype, void* stream) {
  if (mode < 0 || mode > 2) DP_FAIL("pack_conv_weight: bad mode");
  int64_t total = (int64_t)(mode == 0 ? Cout : Cin) * taps * roundup8(mode == 0 ? Cin : Cout);
  int g = (int)((total + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_pack_w<T>, dim3(g), dim3(256), 0, STREAM, w, (T*)dst, Cout, Cin, taps, mode));
  DP_CHECK_LAUNCH("pack_conv_weight"); return 0;
}

// ------------------------------------------------------------------------------------------------ generic conv
struct ConvGeom {
  int N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, CinP, k, stride, pad, dil, mode, ldx, ldy;
};

template <typename T, int NTL>
__global__ void __launch_bounds__(256) k_conv_generic(const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ bias,
                                                      T* __restrict__ y, ConvGeom g) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int64_t Mtot = (int64_t)g.N * g.Do * g.Ho * g.Wo;
  const int64_t mbase = ((int64_t)blockIdx.x * 4 + wv) * 64;          // wave tile: 64 voxels x (NTL*16) channels
  const int co0 = blockIdx.y * (NTL * 16);
  const int taps = g.k * g.k * g.k, c8n = g.CinP >> 3, Ktot8 = taps * c8n, k2 = g.k * g.k;
  if (mbase >= Mtot) return;
  // decode this lane's 4 output voxels (row r of M-tile i)
  int vn[4], vd[4], vh[4], vw[4]; bool vok[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int64_t m = mbase + i * 16 + r; vok[i] = m < Mtot; if (!vok[i]) m = 0;
    vw[i] = (int)(m % g.Wo); int64_t t = m / g.Wo; vh[i] = (int)(t % g.Ho); t /= g.Ho; vd[i] = (int)(t % g.Do); vn[i] = (int)(t / g.Do);
  }
  v4f acc[4][NTL];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NTL; j++) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int64_t wrow = (int64_t)taps * g.CinP;
  for (int ks = 0; ks * 4 < Ktot8; ks++) {
    int k8 = ks * 4 + q; bool kok = k8 < Ktot8;
    int tap = kok ? k8 / c8n : 0, c8 = kok ? k8 - tap * c8n : 0;
    int kd = tap / k2, kh = (tap - kd * k2) / g.k, kw = tap - kd * k2 - kh * g.k;
    int nvc = min(8, g.Cin - c8 * 8);
    Frag8<T> fb[NTL];
#pragma unroll
    for (int j = 0; j < NTL; j++) {
      int co = co0 + j * 16 + r;
      fb[j] = (kok && co < g.Cout) ? frag_load(wp + (int64_t)co * wrow + (int64_t)k8 * 8, 8) : frag_zero<T>();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int id, ih, iw; bool ok = kok && vok[i];
      if (g.mode == 0) {
        id = vd[i] * g.stride - g.pad + kd * g.dil; ih = vh[i] * g.stride - g.pad + kh * g.dil; iw = vw[i] * g.stride - g.pad + kw * g.dil;
      } else {
        int td = vd[i] + g.pad - kd * g.dil, th = vh[i] + g.pad - kh * g.dil, tw = vw[i] + g.pad - kw * g.dil;
        ok = ok && td >= 0 && th >= 0 && tw >= 0 && (td % g.stride == 0) && (th % g.stride == 0) && (tw % g.stride == 0);
        id = td / g.stride; ih = th / g.stride; iw = tw / g.stride;
      }
      ok = ok && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
      Frag8<T> fa = ok ? frag_load(x + ((((int64_t)vn[i] * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx + c8 * 8, nvc) : frag_zero<T>();
#pragma unroll
      for (int j = 0; j < NTL; j++) acc[i][j] = mma16(fa, fb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NTL; j++) {
    int co = co0 + j * 16 + r;
    if (co >= g.Cout) continue;
    float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int64_t m = mbase + i * 16 + q * 4 + e;
        if (m < Mtot) st_f(y + m * g.ldy + co, acc[i][j][e] + bv);
      }
  }
}

int dp_conv3d_tiled_try(const void* x, int ldx, const void* wp, const float* bias, void* y, int ldy, int N, int Di, int Hi, int Wi,
                        int Do, int Ho, int Wo, int Cin, int Cout, int k, int stride, int pad, int dil, int mode, int dtype, void* stream);

extern "C" int dp_conv3d(const void* x, int ldx, const void* wp, const float* bias, void* y, int ldy, int N, int Di, int Hi, int Wi,
                         int Do, int Ho, int Wo, int Cin, int Cout, int k, int stride, int pad, int dil, int mode, int dtype, void* stream) {
  if (mode != 0 && mode != 1) DP_FAIL("conv3d: bad mode");
  if (k < 1 || stride < 1 || dil < 1) DP_FAIL("conv3d: bad geometry");
  int rc = dp_conv3d_tiled_try(x, ldx, wp, bias, y, ldy, N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, k, stride, pad, dil, mode, dtype, stream);
  if (rc >= 0) return rc;       // handled (0) or failed (>0) by the LDS-tiled kernel; -1 = not applicable
  ConvGeom g = {N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, roundup8(Cin), k, stride, pad, dil, mode, ldx, ldy};
  int64_t Mtot = (int64_t)N * Do * Ho * Wo;
  int ntl = Cout > 32 ? 4 : (Cout > 16 ? 2 : 1);
  dim3 grid(cdiv(Mtot, 256), cdiv(Cout, ntl * 16));
  if (grid.y > 65535) DP_FAIL("conv3d: Cout too large");
#define LAUNCH(NTL) DP_DISPATCH(dtype, hipLaunchKernelGGL((k_conv_generic<T, NTL>), grid, dim3(256), 0, STREAM, (const T*)x, (const T*)wp, bias, (T*)y, g))
  if (ntl == 4) LAUNCH(4); else if (ntl == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  DP_CHECK_LAUNCH("conv3d_generic"); return 0;
}

// ------------------------------------------------------------------------------------------------ generic weight gradient
// dw[co][ci][tap] += sum_v gy[v][co (+ tap*choff)] * x[in(v,tap)][ci].  GEMM view: M = co, N = ci, K = voxels.
// Both operands are k-STRIDED in memory (channels are the contiguous axis), so each wave stages its 32-voxel K-step
// of gy and (gathered) x into a private LDS tile [voxel][channel] and reads k-major fragments with
// ds_read_b64_tr_b16 (bf16; map verified by tools/probes/mfma_probe.hip) or plain ds_read_b32 (f32).
// Block = 4 waves that split the block's voxel range; each wave accumulates up to 4x4 (co x ci) tiles and
// atomically adds them to the fp32 gradient at the end.
#define WG_VOX 4096        // voxels per block
template <typename T> struct WgCfg { static constexpr int CB = sizeof(T) == 2 ? 64 : 32;   // channels per block in each of co / ci
                                     static constexpr int ROW = CB + 8; };                 // padded LDS row (elements)

template <typename T> __device__ __forceinline__ Frag8<T> ld_kmajor(const T* tile, int ch0, int lane);
template <> __device__ __forceinline__ Frag8<bf16_t> ld_kmajor<bf16_t>(const bf16_t* tile, int ch0, int lane) {
  constexpr int WG_ROW = WgCfg<bf16_t>::ROW;
  int q = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
  const bf16_t* a = tile + (8 * q + qq) * WG_ROW + ch0 + 4 * p;
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + 4 * WG_ROW));
  Frag8<bf16_t> f;
  f.u[0] = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  f.u[1] = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  f.u[2] = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  f.u[3] = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return f;
}
template <> __device__ __forceinline__ Frag8<f16_t> ld_kmajor<f16_t>(const f16_t* tile, int ch0, int lane) {
  constexpr int WG_ROW = WgCfg<f16_t>::ROW;
  int q = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
  const f16_t* a = tile + (8 * q + qq) * WG_ROW + ch0 + 4 * p;
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + 4 * WG_ROW));
  Frag8<f16_t> f;
  f.u[0] = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  f.u[1] = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  f.u[2] = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  f.u[3] = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return f;
}
template <> __device__ __forceinline__ Frag8<float> ld_kmajor<float>(const float* tile, int ch0, int lane) {
  constexpr int WG_ROW = WgCfg<float>::ROW;
  int q = lane >> 4, r = lane & 15; Frag8<float> f;
#pragma unroll
  for (int j = 0; j < 8; j++) f.v[j] = tile[(8 * q + j) * WG_ROW + ch0 + r];
  return f;
}

struct WgGeom {
  int N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, k, stride, pad, dil, shift, choff, ldx, ldgy;
  int64_t s_co, s_ci, s_tap;
  int64_t vpb;         // output voxels per block (WG_VOX; deterministic mode: all of them, swept by ONE wave per (tap, channel block))
};

template <typename T>
__global__ void __launch_bounds__(256) k_wgrad_generic(const T* __restrict__ x, const T* __restrict__ gy, float* __restrict__ dw, WgGeom g) {
  constexpr int WG_CB = WgCfg<T>::CB, WG_ROW = WgCfg<T>::ROW, CPR = WG_CB / 8;   // CPR = 16-byte chunks per staged row
  __shared__ __attribute__((aligned(16))) T sx[4][32 * WG_ROW];
  __shared__ __attribute__((aligned(16))) T sg[4][32 * WG_ROW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int tap = blockIdx.y, k2 = g.k * g.k;
  const int kd = tap / k2, kh = (tap - kd * k2) / g.k, kw = tap - kd * k2 - kh * g.k;
  const int ncob = (g.Cout + WG_CB - 1) / WG_CB;
  const int cob = blockIdx.z % ncob, cib = blockIdx.z / ncob;
  const int co0 = cob * WG_CB, ci0 = cib * WG_CB;
  const int nco = min(WG_CB, g.Cout - co0), nci = min(WG_CB, g.Cin - ci0);
  const int tco = (nco + 15) >> 4, tci = (nci + 15) >> 4;
  const int64_t Vtot = (int64_t)g.N * g.Do * g.Ho * g.Wo;
  const int64_t vbeg = (int64_t)blockIdx.x * g.vpb, vend = min(Vtot, vbeg + g.vpb);
  const int vstep = 32 * (blockDim.x >> 6);
  T* mx = sx[wv]; T* mg = sg[wv];
  v4f acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int64_t v0 = vbeg + wv * 32; v0 < vend; v0 += vstep) {
    // stage 32 voxels x WG_CB channels of gy and x: 32*CPR chunks each, CPR/2 per lane
#pragma unroll
    for (int u = 0; u < CPR / 2; u++) {
      int c = lane + u * 64, vr = c / CPR, ch = (c % CPR) * 8;
      int64_t v = v0 + vr; bool vok = v < vend;
      int64_t vv = vok ? v : 0;
      int ow = (int)(vv % g.Wo); int64_t t = vv / g.Wo; int oh = (int)(t % g.Ho); t /= g.Ho; int od = (int)(t % g.Do); int n = (int)(t / g.Do);
      int nvg = min(8, nco - ch); nvg = nvg < 0 ? 0 : nvg;
      Frag8<T> fg = (vok && nvg > 0) ? frag_load(gy + vv * g.ldgy + (int64_t)tap * g.choff + co0 + ch, nvg) : frag_zero<T>();
      frag_st_lds(mg + vr * WG_ROW + ch, fg);
      int id = od, ih = oh, iw = ow; bool xok = vok;
      if (g.shift) {
        id = od * g.stride - g.pad + kd * g.dil; ih = oh * g.stride - g.pad + kh * g.dil; iw = ow * g.stride - g.pad + kw * g.dil;
        xok = xok && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
      }
      int nvx = min(8, nci - ch); nvx = nvx < 0 ? 0 : nvx;
      Frag8<T> fx = (xok && nvx > 0) ? frag_load(x + ((((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx + ci0 + ch, nvx) : frag_zero<T>();
      frag_st_lds(mx + vr * WG_ROW + ch, fx);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done (tile is wave-private)
    __builtin_amdgcn_wave_barrier();
    Frag8<T> fa[4], fb[4];
#pragma unroll
    for (int a = 0; a < 4; a++) if (a < tco) fa[a] = ld_kmajor<T>(mg, a * 16, lane);
#pragma unroll
    for (int b = 0; b < 4; b++) if (b < tci) fb[b] = ld_kmajor<T>(mx, b * 16, lane);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) if (a < tco && b < tci) acc[a][b] = mma16(fa[a], fb[b], acc[a][b]);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      if (a >= tco || b >= tci) continue;
      int ci = ci0 + b * 16 + r;
      if (ci >= g.Cin) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int co = co0 + a * 16 + q * 4 + e;
        if (co < g.Cout) atomicAdd(dw + co * g.s_co + ci * g.s_ci + tap * g.s_tap, acc[a][b][e]);
      }
    }
}

int dp_wgrad_tiled_try(const void* x, int ldx, const void* gy, int ldgy, float* dw, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                       int Cin, int Cout, int k, int stride, int pad, int dil, int shift, int gy_tap_choff, int64_t s_co, int64_t s_ci,
                       int64_t s_tap, int dtype, void* stream);

extern "C" int dp_conv3d_wgrad(const void* x, int ldx, const void* gy, int ldgy, float* dw, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                               int Cin, int Cout, int k, int stride, int pad, int dil, int shift, int gy_tap_choff, int64_t s_co, int64_t s_ci,
                               int64_t s_tap, int dtype, void* stream) {
  int rc = dp_wgrad_tiled_try(x, ldx, gy, ldgy, dw, N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, k, stride, pad, dil, shift, gy_tap_choff, s_co, s_ci, s_tap, dtype, stream);
  if (rc >= 0) return rc;
  int64_t Vtot = (int64_t)N * Do * Ho * Wo;
  // deterministic mode: every dW element is one wave's accumulator, added ONCE to the (zeroed) destination -- no atomics meet
  const bool det = dp_det(DET_WGRAD_GENERIC) != 0;
  WgGeom g = {N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, k, stride, pad, dil, shift, gy_tap_choff, ldx, ldgy, s_co, s_ci, s_tap, det ? Vtot : (int64_t)WG_VOX};
  int taps = k * k * k;
  int cb = dtype != DP_F32 ? WgCfg<bf16_t>::CB : WgCfg<float>::CB;
  dim3 grid(det ? 1 : cdiv(Vtot, WG_VOX), taps, cdiv(Cout, cb) * cdiv(Cin, cb));
  if (grid.y > 65535 || grid.z > 65535) DP_FAIL("wgrad: grid too large (taps %d)", taps);
  DP_DISPATCH(dtype, hipLaunchKernelGGL(k_wgrad_generic<T>, grid, dim3(det ? 64 : 256), 0, STREAM, (const T*)x, (const T*)gy, dw, g));
  DP_CHECK_LAUNCH("wgrad_generic"); return 0;
}
