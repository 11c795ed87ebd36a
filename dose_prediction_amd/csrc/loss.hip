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
