// Masked-L1 loss and dose metrics on device.
//   reference: DosePrediction/Train/loss.py:13-28, 69-107 (Loss / GenLoss: mean |pred - gt| over possible_dose_mask > 0),
//              train_light_pyfer.py:166-172 (post-processing: zero where mask < 1 or pred < 0) and
//              Evaluate/evaluate_openKBP.py:42-48 (get_3D_Dose_dif, x70 Gy).
// The reference indexes with a boolean mask (dynamic shape + host synchronisation); here the mean is sum(|p-g|*m)/sum(m)
// as ONE HBM row stream with a deterministic two-stage reduction, and the backward is one more stream.  fp32 tensors
// (the networks hand NCDHW fp32 back at the module boundary).
#include "common.h"

#define STREAM ((hipStream_t)stream)
#define L1_BLOCK 256
#define L1_ELEMS 8192       // elements per block

// flags: bit 0 = post-process the prediction first (p = (mask < 1 || p < 0) ? 0 : p)
__global__ void __launch_bounds__(L1_BLOCK) k_masked_l1_partial(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
                                                                int64_t n, float* __restrict__ part, int flags, float delta) {
  __shared__ float red[2][L1_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * L1_ELEMS, end = min(n, base + L1_ELEMS);
  float s = 0.f, c = 0.f;
  const bool vec = (((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)mask) & 15) == 0;
  auto one = [&](float p, float g, float m) {
    if (flags & 1) p = (m < 1.f || p < 0.f) ? 0.f : p;
    if (m > 0.f) {      // delta > 0: nn.HuberLoss(delta) element (loss.py:53), else |p - g|
      const float a = fabsf(p - g);
      s += delta > 0.f ? (a < delta ? 0.5f * a * a : delta * (a - 0.5f * delta)) : a;
      c += 1.f;
    }
  };
  if (vec) {
    const int64_t end4 = base + ((end - base) & ~(int64_t)3);       // base is a multiple of 4: whole float4 groups, then <= 3 tail elements
    for (int64_t i = base + threadIdx.x * 4; i < end4; i += L1_BLOCK * 4) {
      v4f p = *(const v4f*)(pred + i), g = *(const v4f*)(gt + i), m = *(const v4f*)(mask + i);
#pragma unroll
      for (int k = 0; k < 4; k++) one(p[k], g[k], m[k]);
    }
    if (threadIdx.x == 0) for (int64_t i = end4; i < end; i++) one(pred[i], gt[i], mask[i]);
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += L1_BLOCK) one(pred[i], gt[i], mask[i]);
  }
  s = wave_sum(s); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
// out[0] = sum |p-g| over the mask, out[1] = number of masked elements, out[2] = their ratio (the loss; 0 when the mask is empty)
__global__ void __launch_bounds__(64) k_masked_l1_finish(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  double s = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) { s += part[2 * b]; c += part[2 * b + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  if (threadIdx.x == 0) { out[0] = (float)s; out[1] = (float)c; out[2] = c > 0.0 ? (float)(s / c) : 0.f; }
}
// gpred = gup[0] * sign(p - g) * (mask > 0) / max(count, 1)
__global__ void __launch_bounds__(256) k_masked_l1_bwd(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
                                                       const float* __restrict__ stats, const float* __restrict__ gup, float* __restrict__ gpred, int64_t n,
                                                       float delta) {
  const float sc = gup[0] / fmaxf(stats[1], 1.f);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - gt[i];
    float gsign = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (delta > 0.f) gsign = fabsf(d) < delta ? d : delta * gsign;          // Huber: d inside the quadratic zone, +-delta outside
    gpred[i] = mask[i] > 0.f ? sc * gsign : 0.f;
  }
}
__global__ void __launch_bounds__(256) k_dose_postprocess(const float* __restrict__ pred, const float* __restrict__ mask, float* __restrict__ out, int64_t n, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float p = pred[i];
    out[i] = (mask[i] < 1.f || p < 0.f) ? 0.f : scale * p;
  }
}

extern "C" int64_t dp_masked_l1_ws_elems(int64_t n) { return 2 * ((n + L1_ELEMS - 1) / L1_ELEMS); }
static int masked_fwd(const float* pred, const float* gt, const float* mask, int64_t n, float* ws, float* out3, int postprocess, float delta, void* stream) {
  if (n <= 0) DP_FAIL("masked_l1: empty tensor");
  int64_t nblk = (n + L1_ELEMS - 1) / L1_ELEMS;
  if (nblk > 2000000000LL) DP_FAIL("masked_l1: tensor too large");
  hipLaunchKernelGGL(k_masked_l1_partial, dim3((unsigned)nblk), dim3(L1_BLOCK), 0, STREAM, pred, gt, mask, n, ws, postprocess ? 1 : 0, delta);
  hipLaunchKernelGGL(k_masked_l1_finish, dim3(1), dim3(64), 0, STREAM, (const float*)ws, (int)nblk, out3);
  DP_CHECK_LAUNCH("masked_l1_fwd"); return 0;
}
static int masked_bwd(const float* pred, const float* gt, const float* mask, const float* out3, const float* gup, float* gpred, int64_t n, float delta,
                      void* stream) {
  int64_t g = (n + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_masked_l1_bwd, dim3((unsigned)g), dim3(256), 0, STREAM, pred, gt, mask, out3, gup, gpred, n, delta);
  DP_CHECK_LAUNCH("masked_l1_bwd"); return 0;
}
extern "C" int dp_masked_l1_fwd(const float* pred, const float* gt, const float* mask, int64_t n, float* ws, float* out3, int postprocess, void* stream) {
  return masked_fwd(pred, gt, mask, n, ws, out3, postprocess, 0.f, stream);
}
extern "C" int dp_masked_l1_bwd(const float* pred, const float* gt, const float* mask, const float* out3, const float* gup, float* gpred, int64_t n,
                                void* stream) {
  return masked_bwd(pred, gt, mask, out3, gup, gpred, n, 0.f, stream);
}
extern "C" int dp_masked_huber_fwd(const float* pred, const float* gt, const float* mask, int64_t n, float delta, float* ws, float* out3, void* stream) {
  if (!(delta > 0.f)) DP_FAIL("masked_huber: delta must be > 0");
  return masked_fwd(pred, gt, mask, n, ws, out3, 0, delta, stream);
}
extern "C" int dp_masked_huber_bwd(const float* pred, const float* gt, const float* mask, const float* out3, const float* gup, float* gpred, int64_t n,
                                   float delta, void* stream) {
  if (!(delta > 0.f)) DP_FAIL("masked_huber: delta must be > 0");
  return masked_bwd(pred, gt, mask, out3, gup, gpred, n, delta, stream);
}
extern "C" int dp_dose_postprocess(const float* pred, const float* mask, float* out, int64_t n, float scale, void* stream) {
  int64_t g = (n + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_dose_postprocess, dim3((unsigned)g), dim3(256), 0, STREAM, pred, mask, out, n, scale);
  DP_CHECK_LAUNCH("dose_postprocess"); return 0;
}

// Ground-truth pyramid of GenLoss (loss.py:56-66, 88-97): dose by F.interpolate(mode="trilinear", align_corners=True), the
// possible-dose mask by mode="nearest-exact", both to (Do, Ho, Wo); fp32 single-channel volumes [N][D][H][W].
__global__ void __launch_bounds__(256) k_resample_gt(const float* __restrict__ dose, const float* __restrict__ mask, float* __restrict__ odose,
                                                     float* __restrict__ omask, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Do * Ho * Wo;
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const float nd = (float)Di / (float)Do, nh = (float)Hi / (float)Ho, nw = (float)Wi / (float)Wo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i; const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); t /= Ho; const int od = (int)(t % Do); const int n = (int)(t / Do);
    const float* src = dose + (int64_t)n * Di * Hi * Wi;
    const float fd = od * sd, fh = oh * sh, fw = ow * sw;
    const int d0 = min((int)fd, Di - 1), h0 = min((int)fh, Hi - 1), w0 = min((int)fw, Wi - 1);
    const int d1 = min(d0 + 1, Di - 1), h1 = min(h0 + 1, Hi - 1), w1 = min(w0 + 1, Wi - 1);
    const float td = fd - d0, th = fh - h0, tw = fw - w0;
    auto at = [&](int d, int h, int w) { return src[((int64_t)d * Hi + h) * Wi + w]; };
    const float c00 = at(d0, h0, w0) * (1.f - tw) + at(d0, h0, w1) * tw, c01 = at(d0, h1, w0) * (1.f - tw) + at(d0, h1, w1) * tw;
    const float c10 = at(d1, h0, w0) * (1.f - tw) + at(d1, h0, w1) * tw, c11 = at(d1, h1, w0) * (1.f - tw) + at(d1, h1, w1) * tw;
    odose[i] = (c00 * (1.f - th) + c01 * th) * (1.f - td) + (c10 * (1.f - th) + c11 * th) * td;
    const int md = min((int)floorf((od + 0.5f) * nd), Di - 1), mh = min((int)floorf((oh + 0.5f) * nh), Hi - 1), mw = min((int)floorf((ow + 0.5f) * nw), Wi - 1);
    omask[i] = mask[(int64_t)n * Di * Hi * Wi + ((int64_t)md * Hi + mh) * Wi + mw];
  }
}
extern "C" int dp_resample_gt(const float* dose, const float* mask, float* out_dose, float* out_mask, int N, int Di, int Hi, int Wi, int Do, int Ho,
                              int Wo, void* stream) {
  if (Do < 1 || Ho < 1 || Wo < 1) DP_FAIL("resample_gt: empty output");
  int64_t total = (int64_t)N * Do * Ho * Wo, g = (total + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_resample_gt, dim3((unsigned)g), dim3(256), 0, STREAM, dose, mask, out_dose, out_mask, N, Di, Hi, Wi, Do, Ho, Wo);
  DP_CHECK_LAUNCH("resample_gt"); return 0;
}

// ================================================================================================ DiceCE (OAR-TRANSEG's training loss)
// MONAI 0.7.0 DiceCELoss(to_onehot_y=True, softmax=True) as the reference builds it (OARSegmentation/train_light_transeg.py:148, called
// at :196 / :212 on the network's logits [B][C][D][H][W] and the label volume [B][1][D][H][W]):
//   p = softmax(logits, channel);  onehot = one_hot(label);
//   dice = mean over (b, c) of 1 - (2 I_bc + smooth_nr) / (G_bc + P_bc + smooth_dr),  I = sum_v onehot p, G = sum_v onehot, P = sum_v p
//   ce   = nn.CrossEntropyLoss(reduction="mean")(logits, label) = mean over (b, v) of -log p[label]
//   loss = lambda_dice dice + lambda_ce ce                                   (include_background, no squared_pred / jaccard / batch)
// Forward: ONE pass over the logits (softmax, CE term and the three Dice sums per voxel; 1 + 3 C partial sums per block, combined in a
// fixed order by the finish kernel, which also leaves the per-(b, c) coefficients of the backward pass in `stats`).  Backward: ONE more
// pass that recomputes the softmax and writes d loss / d logits.  The ATen composition this replaces in the benchmark's C3 step
// (cross_entropy alone: log_softmax + nll_loss forward / backward, 4 launches, 0.53 ms at 2 x 128^3) had no Dice term at all.
#define DCE_BLOCK 256
#define DCE_VOX 4096        // voxels per block
enum { DCE_LBL_F32 = 0, DCE_LBL_I64 = 1, DCE_LBL_I32 = 2, DCE_LBL_U8 = 3 };
__device__ __forceinline__ int dce_label(const void* lab, int kind, int64_t i) {
  switch (kind) {
    case DCE_LBL_F32: return (int)((const float*)lab)[i];
    case DCE_LBL_I64: return (int)((const int64_t*)lab)[i];
    case DCE_LBL_I32: return ((const int*)lab)[i];
    default: return (int)((const unsigned char*)lab)[i];
  }
}
// stats layout: [0] loss, [1] dice, [2] ce, [3] lambda_ce / (B V), then per (b, c): [4 + 2 (b C + c)] = u_bc, [+1] = w_bc with
//   d(lambda_dice dice) / d p_c(b, v) = u_bc - w_bc [label(b, v) == c]
template <int CM>
__global__ void __launch_bounds__(DCE_BLOCK) k_dice_ce_partial(const float* __restrict__ z, const void* __restrict__ lab, int lkind, int C, int64_t V, int nblk,
                                                               float* __restrict__ part) {
  __shared__ float red[DCE_BLOCK / 64][1 + 3 * CM];
  const int b = blockIdx.y, blk = blockIdx.x;
  const float* zb = z + (int64_t)b * C * V;
  const int64_t v0 = (int64_t)blk * DCE_VOX, v1 = min(V, v0 + DCE_VOX);
  float ce = 0.f, I[CM], G[CM], P[CM];
#pragma unroll
  for (int c = 0; c < CM; c++) I[c] = G[c] = P[c] = 0.f;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += DCE_BLOCK) {
    const int t = dce_label(lab, lkind, (int64_t)b * V + v);
    float l[CM], m = -INFINITY, zt = 0.f;
#pragma unroll
    for (int c = 0; c < CM; c++) { l[c] = c < C ? zb[(int64_t)c * V + v] : -INFINITY; m = fmaxf(m, l[c]); if (c == t) zt = l[c]; }
    float S = 0.f;
#pragma unroll
    for (int c = 0; c < CM; c++) { l[c] = c < C ? expf(l[c] - m) : 0.f; S += l[c]; }
    const float rS = 1.f / S;
    if (t >= 0 && t < C) ce += (m + logf(S)) - zt;        // -log softmax[t] in its log-sum-exp form
#pragma unroll
    for (int c = 0; c < CM; c++) {
      const float p = l[c] * rS;
      P[c] += p;
      if (c == t) { I[c] += p; G[c] += 1.f; }
    }
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  ce = wave_sum(ce);
  if (lane == 0) red[wv][0] = ce;
#pragma unroll
  for (int c = 0; c < CM; c++) {
    const float a = wave_sum(I[c]), g = wave_sum(G[c]), p = wave_sum(P[c]);
    if (lane == 0) { red[wv][1 + c] = a; red[wv][1 + CM + c] = g; red[wv][1 + 2 * CM + c] = p; }
  }
  __syncthreads();
  float* out = part + ((int64_t)b * nblk + blk) * (1 + 3 * CM);
  for (int j = threadIdx.x; j < 1 + 3 * CM; j += DCE_BLOCK) out[j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}
template <int CM>
__global__ void __launch_bounds__(1024) k_dice_ce_finish(const float* __restrict__ part, int B, int C, int64_t V, int nblk, float snr, float sdr, float ld, float lc,
                                                         float* __restrict__ stats) {
  // column j < 1 + 3 CM of sample b: 16 threads take every 16th block row each (eight loads in flight), then a fixed-order tree over the 16
  // (round 5: one thread per column walked all the rows one dependent load at a time -- 0.18 ms for 216 rows x 4 samples)
  constexpr int NCOL = 1 + 3 * CM, G = 16;
  __shared__ double tot[NCOL];
  __shared__ double red[NCOL][G];
  __shared__ double acc[2];
  const int col = threadIdx.x / G, sub = threadIdx.x % G;
  if (threadIdx.x == 0) { acc[0] = 0.0; acc[1] = 0.0; }
  for (int b = 0; b < B; b++) {
    __syncthreads();
    if (col < NCOL) {
      double s = 0.0;
      const float* p = part + (int64_t)b * nblk * NCOL + col;
      int k = sub;
      for (; k + 7 * G < nblk; k += 8 * G) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(int64_t)(k + u * G) * NCOL];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
      }
      for (; k < nblk; k += G) s += p[(int64_t)k * NCOL];
      red[col][sub] = s;
    }
    __syncthreads();
    if (col < NCOL && sub == 0) {
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < G; u++) s += red[col][u];
      tot[col] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      acc[1] += tot[0];
      for (int c = 0; c < C; c++) {
        const double I = tot[1 + c], den = tot[1 + CM + c] + tot[1 + 2 * CM + c] + (double)sdr, num = 2.0 * I + (double)snr;
        acc[0] += 1.0 - num / den;
        stats[4 + 2 * (b * C + c)] = (float)((double)ld / ((double)B * C) * num / (den * den));
        stats[4 + 2 * (b * C + c) + 1] = (float)((double)ld / ((double)B * C) * 2.0 / den);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dice = acc[0] / ((double)B * C), ce = acc[1] / ((double)B * (double)V);
    stats[0] = (float)((double)ld * dice + (double)lc * ce); stats[1] = (float)dice; stats[2] = (float)ce;
    stats[3] = (float)((double)lc / ((double)B * (double)V));
  }
}
template <int CM>
__global__ void __launch_bounds__(256) k_dice_ce_bwd(const float* __restrict__ z, const void* __restrict__ lab, int lkind, int C, int64_t V,
                                                     const float* __restrict__ stats, const float* __restrict__ gup, float* __restrict__ gz) {
  const int b = blockIdx.y;
  const float* zb = z + (int64_t)b * C * V; float* gb = gz + (int64_t)b * C * V;
  float u[CM], w[CM];
#pragma unroll
  for (int c = 0; c < CM; c++) { u[c] = c < C ? stats[4 + 2 * (b * C + c)] : 0.f; w[c] = c < C ? stats[4 + 2 * (b * C + c) + 1] : 0.f; }
  const float g0 = gup[0], cs = stats[3];
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    float l[CM], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CM; c++) { l[c] = c < C ? zb[(int64_t)c * V + v] : -INFINITY; m = fmaxf(m, l[c]); }
    float S = 0.f;
#pragma unroll
    for (int c = 0; c < CM; c++) { l[c] = c < C ? expf(l[c] - m) : 0.f; S += l[c]; }
    const float rS = 1.f / S;
    const int t = dce_label(lab, lkind, (int64_t)b * V + v);
    // a label outside [0, C) contributes no cross-entropy term in k_dice_ce_partial (torch / MONAI raise on such labels; here the voxel is
    // ignored by the CE half and matches no class in the Dice half): its CE gradient is zero as well, so backward == d(forward)
    const float csv = (t >= 0 && t < C) ? cs : 0.f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CM; c++) { l[c] *= rS; s += (u[c] - (c == t ? w[c] : 0.f)) * l[c]; }
#pragma unroll
    for (int c = 0; c < CM; c++)
      if (c < C) gb[(int64_t)c * V + v] = g0 * (l[c] * ((u[c] - (c == t ? w[c] : 0.f)) - s) + csv * (l[c] - (c == t ? 1.f : 0.f)));
  }
}
extern "C" int64_t dp_dice_ce_ws_elems(int B, int C, int64_t V) {
  const int cm = C <= 8 ? 8 : 16;
  return (int64_t)B * ((V + DCE_VOX - 1) / DCE_VOX) * (1 + 3 * cm);
}
extern "C" int64_t dp_dice_ce_stats_elems(int B, int C) { return 4 + 2 * (int64_t)B * C; }
extern "C" int dp_dice_ce_fwd(const float* logits, const void* labels, int label_kind, int B, int C, int64_t V, float smooth_nr, float smooth_dr,
                              float lambda_dice, float lambda_ce, float* ws, float* stats, void* stream) {
  if (B < 1 || C < 2 || C > 16 || V < 1) DP_FAIL("dice_ce: need B >= 1, 2 <= C <= 16, V >= 1 (got %d, %d, %lld)", B, C, (long long)V);
  if (label_kind < 0 || label_kind > 3) DP_FAIL("dice_ce: label kind %d (0 float32, 1 int64, 2 int32, 3 uint8)", label_kind);
  const int64_t nblk = (V + DCE_VOX - 1) / DCE_VOX;
  if (nblk > 2000000000LL || B > 65535) DP_FAIL("dice_ce: tensor too large");
  if (C <= 8) {
    hipLaunchKernelGGL(k_dice_ce_partial<8>, dim3((unsigned)nblk, B), dim3(DCE_BLOCK), 0, STREAM, logits, labels, label_kind, C, V, (int)nblk, ws);
    hipLaunchKernelGGL(k_dice_ce_finish<8>, dim3(1), dim3(16 * 25), 0, STREAM, (const float*)ws, B, C, V, (int)nblk, smooth_nr, smooth_dr, lambda_dice, lambda_ce, stats);
  } else {
    hipLaunchKernelGGL(k_dice_ce_partial<16>, dim3((unsigned)nblk, B), dim3(DCE_BLOCK), 0, STREAM, logits, labels, label_kind, C, V, (int)nblk, ws);
    hipLaunchKernelGGL(k_dice_ce_finish<16>, dim3(1), dim3(16 * 49), 0, STREAM, (const float*)ws, B, C, V, (int)nblk, smooth_nr, smooth_dr, lambda_dice, lambda_ce, stats);
  }
  DP_CHECK_LAUNCH("dice_ce_fwd"); return 0;
}
extern "C" int dp_dice_ce_bwd(const float* logits, const void* labels, int label_kind, int B, int C, int64_t V, const float* stats, const float* gup,
                              float* glogits, void* stream) {
  if (B < 1 || C < 2 || C > 16 || V < 1 || B > 65535) DP_FAIL("dice_ce_bwd: bad shape");
  int64_t g = (V + 255) / 256; if (g > 8192) g = 8192;
  if (C <= 8) hipLaunchKernelGGL(k_dice_ce_bwd<8>, dim3((unsigned)g, B), dim3(256), 0, STREAM, logits, labels, label_kind, C, V, stats, gup, glogits);
  else hipLaunchKernelGGL(k_dice_ce_bwd<16>, dim3((unsigned)g, B), dim3(256), 0, STREAM, logits, labels, label_kind, C, V, stats, gup, glogits);
  DP_CHECK_LAUNCH("dice_ce_bwd"); return 0;
}
