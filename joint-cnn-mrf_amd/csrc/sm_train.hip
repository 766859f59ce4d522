// Backward of the spatial model (main.py:94-125) for the joint training step.
//
// Forward (per image, pair p = (j, c)):  s_c = sp(h_c),  Cpre_p = A_p (*) s_c  [61x91],
//   T_p = R(Cpre_p) + sp(b_p) + d,   E_j = log(s_j + d) + sum_p log T_p,   h = BN_train(hm10).
// With G_j = dL/dE_j and q_p = G_j / T_p:
//   d sp(b_p) = sum_b q_p                         d s_j  += G_j / (s_j + d)
//   D_p = R^T q_p placed on the window [59..119] x [89..179] of a zero 120x180 frame
//   dA_p  = sum_b  D_p (star) s_c   = IFFT( sum_b  FFT(D_p) conj(FFT(s_c)) ) / N
//   d s_c += sum_{p: cond p = c}  IFFT( FFT(D_p) conj(FFT(A_p)) ) / N   on [0..59] x [0..89]
// The circular 120x180 frame is alias-free for both correlations for the same reason the forward
// is: the loss only sees the window, on which circular == linear convolution.
// The sum over the batch (dA) and over the pairs of a conditioning joint (ds) are taken in the
// frequency domain, so the backward costs 81 forward transforms per image + 10 inverse per image
// + 81 inverse per step -- all of them in LDS (sm_lds.hip; spectra transposed [91][120], which the elementwise kernels here do not care about).
#include "kernels.h"

namespace jcm {

namespace {
constexpr int F_H = 120, F_W = 180, F_WC = F_W / 2 + 1;
constexpr int F_HW = F_H * F_W, F_HWC = F_H * F_WC;
constexpr int FM_H = 60, FM_W = 90, FM_HW = FM_H * FM_W;
constexpr float kDelta = 1e-6f;

__device__ __forceinline__ float sigmoid5(float x) { return 1.0f / (1.0f + expf(-5.0f * x)); }   // d/dx [softplus(5x)/5]

}  // namespace

// scale = gamma * rstd, shift = beta - mean * scale    (training-mode BN as one multiply-add)
__global__ void bn_fold_stats_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ g,
                                     const float* __restrict__ b, float* __restrict__ sc, float* __restrict__ sh, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = g[i] * rstd[i];
  sc[i] = s;
  sh[i] = b[i] - mean[i] * s;
}
hipError_t bn_fold_stats(const float* mean, const float* rstd, const float* gamma, const float* beta, float* sc, float* sh, int n,
                         hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mean, rstd, gamma, beta, sc, sh, n);
  return hipGetLastError();
}

// hm10[n][c] = c < K ? prob[n][c] : y[n][c]      (tf.concat([hm_pred_pd, hm_target[..., K:]], 3), main.py:528)
__global__ void sm_concat_target_kernel(const float* __restrict__ prob, const float* __restrict__ y, float* __restrict__ out, int K, int C,
                                        size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const size_t n = i / C;
    out[i] = c < K ? prob[n * K + c] : y[i];
  }
}
hipError_t sm_concat_target(const float* prob, const float* y, float* out, size_t N, int K, int C, hipStream_t st) {
  const size_t total = N * C;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_concat_target_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, prob, y, out, K, C, total);
  return hipGetLastError();
}

// dspb[p][pix] (+)= sum_b G[b,pix,j(p)] / T[b][p][pix]
__global__ void sm_bwd_dbias_kernel(const float* __restrict__ G, const float* __restrict__ T, float* __restrict__ dspb, int nb, int K,
                                    int P, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * FM_HW) return;
  const int pix = i % FM_HW, p = i / FM_HW;
  const int j = p / (P / K);
  float s = accumulate ? dspb[i] : 0.f;
  for (int b = 0; b < nb; ++b) s += G[((size_t)b * FM_HW + pix) * K + j] / T[((size_t)b * P + p) * FM_HW + pix];
  dspb[i] = s;
}
hipError_t sm_bwd_dbias(const float* G, const float* T, float* dspb, int nb, int K, int P, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(sm_bwd_dbias_kernel, dim3((P * FM_HW + 255) / 256), dim3(256), 0, st, G, T, dspb, nb, K, P, accumulate);
  return hipGetLastError();
}

// dAhat[p][k] (+)= sum_b Dhat[b][p][k] * conj(Lhat[b][cond[p]][k])
__global__ void sm_bwd_spec_da_kernel(const float2* __restrict__ Dhat, const float2* __restrict__ Lhat, const int* __restrict__ cond,
                                      float2* __restrict__ dA, int nb, int C, int P, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * F_HWC) return;
  const int k = i % F_HWC, p = i / F_HWC;
  const int c = cond[p];
  float2 s = accumulate ? dA[i] : make_float2(0.f, 0.f);
  for (int b = 0; b < nb; ++b) {
    const float2 d = Dhat[((size_t)b * P + p) * F_HWC + k];
    const float2 l = Lhat[((size_t)b * C + c) * F_HWC + k];
    s.x += d.x * l.x + d.y * l.y;
    s.y += d.y * l.x - d.x * l.y;
  }
  dA[i] = s;
}
hipError_t sm_bwd_spec_da(const float2* Dhat, const float2* Lhat, const int* cond, float2* dA, int nb, int C, int P, int accumulate,
                          hipStream_t st) {
  hipLaunchKernelGGL(sm_bwd_spec_da_kernel, dim3((P * F_HWC + 255) / 256), dim3(256), 0, st, Dhat, Lhat, cond, dA, nb, C, P, accumulate);
  return hipGetLastError();
}

// dLhat[b][c][k] = sum_{j != c, j < K} Dhat[b][p(j,c)][k] * conj(Ahat[p][k]),  p(j,c) = j*(C-1) + (c < j ? c : c-1)
__global__ void sm_bwd_spec_dl_kernel(const float2* __restrict__ Dhat, const float2* __restrict__ Ahat, float2* __restrict__ dL, int K,
                                      int C, int64_t total) {
  const int P = K * (C - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = i % F_HWC;
    int64_t r = i / F_HWC;
    const int c = r % C;
    const int64_t b = r / C;
    float2 s = make_float2(0.f, 0.f);
    for (int j = 0; j < K; ++j) {
      if (j == c) continue;
      const int p = j * (C - 1) + (c < j ? c : c - 1);
      const float2 d = Dhat[((size_t)b * P + p) * F_HWC + k];
      const float2 a = Ahat[(size_t)p * F_HWC + k];
      s.x += d.x * a.x + d.y * a.y;
      s.y += d.y * a.x - d.x * a.y;
    }
    dL[i] = s;
  }
}
hipError_t sm_bwd_spec_dl(const float2* Dhat, const float2* Ahat, float2* dL, int nb, int K, int C, hipStream_t st) {
  const int64_t total = (int64_t)nb * C * F_HWC;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_bwd_spec_dl_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, Dhat, Ahat, dL, K, C, total);
  return hipGetLastError();
}

// dh[b,pix,c] = ( dLframe[b][c][oy][ox] / N  +  [c < K] G[b,pix,c] / (s_c + d) ) * sigmoid(5 h),  h = hm*sc + sh, s_c = frame value
__global__ void sm_bwd_dh_kernel(const float* __restrict__ dLframe, const float* __restrict__ G, const float* __restrict__ frame,
                                 const float* __restrict__ hm, const float* __restrict__ sc, const float* __restrict__ sh,
                                 float* __restrict__ dh, int K, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    int64_t r = i / C;
    const int pix = r % FM_HW;
    const int64_t b = r / FM_HW;
    const int oy = pix / FM_W, ox = pix - oy * FM_W;
    const size_t fi = ((size_t)(b * C + c) * F_H + oy) * F_W + ox;
    float ds = dLframe[fi] * (1.0f / (float)F_HW);
    if (c < K) ds += G[((size_t)b * FM_HW + pix) * K + c] / (frame[fi] + kDelta);
    dh[i] = ds * sigmoid5(hm[i] * sc[c] + sh[c]);
  }
}
hipError_t sm_bwd_dh(const float* dLframe, const float* G, const float* frame, const float* hm, const float* sc, const float* sh, float* dh,
                     int nb, int K, int C, hipStream_t st) {
  const int64_t total = (int64_t)nb * FM_HW * C;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_bwd_dh_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, dLframe, G, frame, hm, sc, sh, dh, K, C, total);
  return hipGetLastError();
}

// parameter gradients of pair p: d energy = dAframe / N * sigmoid(5 e),  d bias = dspb * sigmoid(5 b); written at the pair's
// offsets in the flat gradient buffer
__global__ void sm_bwd_params_kernel(const float* __restrict__ dAframe, const float* __restrict__ dspb, const float* const* __restrict__ e_ptr,
                                     const float* const* __restrict__ b_ptr, const int64_t* __restrict__ e_off,
                                     const int64_t* __restrict__ b_off, float* __restrict__ grads, int P) {
  const int p = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F_HW) grads[e_off[p] + i] = dAframe[(size_t)p * F_HW + i] * (1.0f / (float)F_HW) * sigmoid5(e_ptr[p][i]);
  if (i < FM_HW) grads[b_off[p] + i] = dspb[(size_t)p * FM_HW + i] * sigmoid5(b_ptr[p][i]);
}
hipError_t sm_bwd_params(const float* dAframe, const float* dspb, const float* const* e_ptr, const float* const* b_ptr, const int64_t* e_off,
                         const int64_t* b_off, float* grads, int P, hipStream_t st) {
  hipLaunchKernelGGL(sm_bwd_params_kernel, dim3((F_HW + 255) / 256, P), dim3(256), 0, st, dAframe, dspb, e_ptr, b_ptr, e_off, b_off, grads, P);
  return hipGetLastError();
}

}  // namespace jcm
