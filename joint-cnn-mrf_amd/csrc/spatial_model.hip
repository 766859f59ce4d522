// MRF spatial model (main.py:77-125) in log space, fp32.
//
//   E_j = log(sp(h_j)+d) + sum_{c != j} log( R( sp(e_{j|c}) (*) sp(h_c) ) + sp(b_{j|c}) + d )
//
// sp(x) = softplus(5x)/5 (main.py:106-108), d = 1e-6 (:110), (*) = VALID true convolution of
// the 120x180 prior with the 60x90 likelihood -> 61x91 (:83-87), R = TF-1.x bilinear
// 61x91 -> 60x90 (:89), h = bn_sm(heat_map) (:112-113).
//
// Batch-independent operands (sp(e), sp(b)) are tabulated once at jcm_finalize; sp(h_c) is
// computed once per (image, channel) (10x, not 81x as the graph does).  The pairwise
// convolution here is a direct LDS-resident sliding-window kernel on the VALU:
//   - one workgroup per (image, pair) keeps the whole prior (120 x 200-float pitch = 96 KB) in LDS;
//   - a thread owns 8 consecutive outputs of one row and slides an 8+8 register window along x;
//   - the likelihood row is wave-uniform, stored reversed and zero-padded ([60][96]) so it is
//     fetched through the scalar cache and costs no LDS bandwidth.
#include "kernels.h"

namespace jcm {

constexpr int SM_H = 60, SM_W = 90, SM_HW = SM_H * SM_W;     // heat map (data.py:12)
constexpr int SM_PH = 120, SM_PW = 180;                       // prior (prepare_pairwise_distribution.py:37)
constexpr int SM_CH = 61, SM_CW = 91, SM_CHW = SM_CH * SM_CW; // VALID output (main.py:87)
constexpr int SM_LP = 96;                                     // reversed likelihood row pitch (90 -> 96, zero padded)
constexpr int SM_PP = 200;                                    // prior row pitch in LDS (180 -> 200, zero padded)
constexpr float SM_DELTA_F = 1e-6f;

// tf.nn.softplus shortcuts at +-(log(eps_f32)+2)
__device__ __forceinline__ float softplus5(float x) {
  const float z = 5.0f * x;
  const float thr = 13.942385f;
  float s;
  if (z > thr) s = z;
  else if (z < -thr) s = expf(z);
  else s = log1pf(expf(z));
  return 0.2f * s;
}

__global__ void sm_softplus5_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = softplus5(in[i]);
}
hipError_t sm_softplus5(const float* in, float* out, int64_t n, hipStream_t st) {
  int64_t g = (n + 255) / 256;
  hipLaunchKernelGGL(sm_softplus5_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}

__global__ void sm_softplus5_multi_kernel(const float* const* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int p = blockIdx.y;
  const float* src = in[p];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[(size_t)p * n + i] = softplus5(src[i]);
}
hipError_t sm_softplus5_multi(const float* const* in, float* out, int P, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(sm_softplus5_multi_kernel, dim3((int)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), P), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}

// lik[b][c] : [60][96] rows reversed along x:  lik[u][t] = sp(bn(h[b,u,89-t,c])) for t<90, 0 beyond;
// the unary term reads it back as lik[u][89-x].
// channels [0,Ca) come from hm [B,5400,Ca], the rest from extra [B,5400,C-Ca] (the concat of main.py:528 read in place)
__global__ void sm_likelihood_kernel(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld, const float* __restrict__ sc,
                                     const float* __restrict__ sh, float* __restrict__ lik, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = i % SM_LP;
    int64_t r = i / SM_LP;
    const int u = r % SM_H; r /= SM_H;
    const int c = r % C;
    const int64_t b = r / C;
    float v = 0.f;
    if (t < SM_W) {
      const int64_t pix = (b * SM_H + u) * SM_W + (SM_W - 1 - t);
      const float hv = c < Ca ? hm[pix * Ca + c] : extra[pix * extra_ld + (c - Ca)];
      v = sc ? softplus5(hv * sc[c] + sh[c]) : hv;   // sc == nullptr: raw reversed copy (jcm_conv_mrf)
    }
    lik[i] = v;
  }
}
hipError_t sm_likelihood(const float* hm, int Ca, const float* extra, const float* bn_scale, const float* bn_shift, float* lik, int B, int C,
                         hipStream_t st, int extra_ld) {
  if (extra_ld <= 0) extra_ld = C - Ca;
  const int64_t total = (int64_t)B * C * SM_H * SM_LP;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_likelihood_kernel, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, st, hm, Ca, extra, extra_ld, bn_scale, bn_shift, lik, C, total);
  return hipGetLastError();
}

// cpre[b][p][y][x] = sum_{u,v} prior_p[y+59-u][x+89-v] * L[b][cond[p]][u][v]
//                  = sum_u sum_t prior_p[y+59-u][x+t] * likr[u][t]            (t = 89-v)
constexpr int PC_XT = 8;
constexpr int PC_STRIPS = SM_LP / PC_XT;          // 12 strips of 8 cover x = 0..95
constexpr int PC_THREADS = 768;                   // 61 rows x 12 strips = 732 active
__global__ __launch_bounds__(PC_THREADS) void sm_pair_conv_kernel(const float* __restrict__ priors, const float* __restrict__ lik,
                                                                 const int* __restrict__ cond, float* __restrict__ cpre, int P, int C) {
  extern __shared__ __attribute__((aligned(16))) float pl[];   // [120][200]
  const int p = blockIdx.x % P;
  const int b = blockIdx.x / P;
  const int tid = threadIdx.x;
  const float* pr = priors + (size_t)p * SM_PH * SM_PW;
  for (int i = tid; i < SM_PH * SM_PP; i += PC_THREADS) {
    const int r = i / SM_PP, c = i - r * SM_PP;
    pl[i] = c < SM_PW ? pr[r * SM_PW + c] : 0.f;
  }
  __syncthreads();
  const int y = tid / PC_STRIPS, x0 = (tid - y * PC_STRIPS) * PC_XT;
  if (y >= SM_CH) return;
  const float* lk = lik + ((size_t)b * C + cond[p]) * SM_H * SM_LP;   // workgroup-uniform
  float acc[PC_XT];
#pragma unroll
  for (int j = 0; j < PC_XT; ++j) acc[j] = 0.f;
  for (int u = 0; u < SM_H; ++u) {
    // Two-level summation: a likelihood row (90 terms) accumulates into a fresh partial, the 60
    // partials into the total.  A single 5400-term fp32 chain of near-constant positive terms
    // drifts by ~3e-5 relative (correlated roundings); this keeps it at the 1e-6 level.
    float racc[PC_XT];
#pragma unroll
    for (int j = 0; j < PC_XT; ++j) racc[j] = 0.f;
    const float* arow = pl + (y + SM_H - 1 - u) * SM_PP + x0;
    const float* hrow = lk + u * SM_LP;
    float win[2 * PC_XT];
    *reinterpret_cast<float4*>(win) = *reinterpret_cast<const float4*>(arow);
    *reinterpret_cast<float4*>(win + 4) = *reinterpret_cast<const float4*>(arow + 4);
#pragma unroll
    for (int tb = 0; tb < SM_LP / PC_XT; ++tb) {
      *reinterpret_cast<float4*>(win + 8) = *reinterpret_cast<const float4*>(arow + tb * 8 + 8);
      *reinterpret_cast<float4*>(win + 12) = *reinterpret_cast<const float4*>(arow + tb * 8 + 12);
#pragma unroll
      for (int i = 0; i < PC_XT; ++i) {
        const float hv = hrow[tb * 8 + i];   // uniform -> scalar load
#pragma unroll
        for (int j = 0; j < PC_XT; ++j) racc[j] = fmaf(hv, win[i + j], racc[j]);
      }
#pragma unroll
      for (int j = 0; j < PC_XT; ++j) win[j] = win[j + 8];
    }
#pragma unroll
    for (int j = 0; j < PC_XT; ++j) acc[j] += racc[j];
  }
  float* o = cpre + ((size_t)b * P + p) * SM_CHW + y * SM_CW + x0;
#pragma unroll
  for (int j = 0; j < PC_XT; ++j)
    if (x0 + j < SM_CW) o[j] = acc[j];
}

hipError_t sm_pair_conv(const float* priors, const float* maps, const int* cond, float* cpre, int B, int P, int C, hipStream_t st) {
  const int lds = SM_PH * SM_PP * sizeof(float);
  static LdsAttr attr;   // per device, not per process: a second Engine on another GPU needs its own call
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(sm_pair_conv_kernel), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(sm_pair_conv_kernel, dim3(B * P), dim3(PC_THREADS), lds, st, priors, maps, cond, cpre, P, C);
  return hipGetLastError();
}

// TF-1.x bilinear 61x91 -> 60x90 at one output pixel (main.py:89).
__device__ __forceinline__ float resize_61x91(const float* __restrict__ c, int oy, int ox) {
  const float sy = (float)SM_CH / (float)SM_H, sx = (float)SM_CW / (float)SM_W;
  const float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);   // rounded, never fused into the lerp-weight subtract
  const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  const int yhi = min(ylo + 1, SM_CH - 1), xhi = min(xlo + 1, SM_CW - 1);
  const float ty = fy - (float)ylo, tx = fx - (float)xlo;
  const float tl = c[ylo * SM_CW + xlo], tr = c[ylo * SM_CW + xhi];
  const float bl = c[yhi * SM_CW + xlo], br = c[yhi * SM_CW + xhi];
  const float top = tl + (tr - tl) * tx;
  const float bot = bl + (br - bl) * tx;
  return top + (bot - top) * ty;
}

// One thread per (b, j, pixel); the C-1 pairs of joint j are p = j*(C-1) .. in graph order.
__global__ void sm_finish_kernel(const float* __restrict__ lik, const float* __restrict__ cpre, const float* __restrict__ spb,
                                 float* __restrict__ logits, int K, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % SM_HW;
    int64_t r = i / SM_HW;
    const int j = r % K;
    const int64_t b = r / K;
    const int oy = pix / SM_W, ox = pix - oy * SM_W;
    float e = logf(lik[((b * C + j) * SM_H + oy) * SM_LP + (SM_W - 1 - ox)] + SM_DELTA_F);   // main.py:117
    const int PJ = C - 1, P = K * PJ;
    for (int q = 0; q < PJ; ++q) {                                                        // main.py:118-123
      const int p = j * PJ + q;
      const float cv = resize_61x91(cpre + ((size_t)b * P + p) * SM_CHW, oy, ox);
      e += logf((cv + spb[(size_t)p * SM_HW + pix]) + SM_DELTA_F);
    }
    logits[((size_t)b * SM_HW + pix) * K + j] = e;
  }
}
hipError_t sm_finish(const float* lik, const float* cpre, const float* spbias, float* logits, int B, int K, int C, hipStream_t st) {
  const int64_t total = (int64_t)B * K * SM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_finish_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, lik, cpre, spbias, logits, K, C, total);
  return hipGetLastError();
}

__global__ void sm_resize_only_kernel(const float* __restrict__ cpre, float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % SM_HW;
    const int64_t b = i / SM_HW;
    out[i] = resize_61x91(cpre + (size_t)b * SM_CHW, pix / SM_W, pix % SM_W);
  }
}
hipError_t sm_resize_only(const float* cpre, float* out, int B, hipStream_t st) {
  const int64_t total = (int64_t)B * SM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_resize_only_kernel, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, st, cpre, out, total);
  return hipGetLastError();
}

}  // namespace jcm
