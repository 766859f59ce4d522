// Plain kernels around the spatial model's 120 x 180 frames (main.py:77-91,94-125): the zero-padded likelihood frame, the elementwise
// spectrum product and the resize of the VALID window.  The transforms themselves are in sm_fused.hip / sm_lds.hip (in LDS, hand-written);
// rounds 1-4 had rocFFT (hipFFT API) calls in this file, removed in round 5.
#include "kernels.h"

namespace jcm {

constexpr int F_H = 120, F_W = 180, F_WC = F_W / 2 + 1;      // real frame, complex row length
constexpr int F_HW = F_H * F_W, F_HWC = F_H * F_WC;
constexpr int FM_H = 60, FM_W = 90, FM_HW = FM_H * FM_W;

__device__ __forceinline__ float softplus5f(float x) {
  const float z = 5.0f * x;
  const float thr = 13.942385f;
  float s;
  if (z > thr) s = z;
  else if (z < -thr) s = expf(z);
  else s = log1pf(expf(z));
  return 0.2f * s;
}

// frame[b][c][120][180]: sp(bn(h[b,y,x,c])) on the top-left 60x90, zero elsewhere (sc == nullptr: raw copy).
// The C-channel map is read from two tensors -- channels [0,Ca) from hm [B,5400,Ca], the rest from extra
// [B,5400,C-Ca] -- which is tf.concat([hm, torso], axis=3) of main.py:528 without materialising it.
__global__ void sm_pad_frame_kernel(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld, const float* __restrict__ sc,
                                    const float* __restrict__ sh, float* __restrict__ frame, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = i % F_W;
    int64_t r = i / F_W;
    const int y = r % F_H; r /= F_H;
    const int c = r % C;
    const int64_t b = r / C;
    float v = 0.f;
    if (y < FM_H && x < FM_W) {
      const int64_t pix = (b * FM_H + y) * FM_W + x;
      const float hv = c < Ca ? hm[pix * Ca + c] : extra[pix * extra_ld + (c - Ca)];
      v = sc ? softplus5f(hv * sc[c] + sh[c]) : hv;
    }
    frame[i] = v;
  }
}
hipError_t sm_pad_frame(const float* hm, int Ca, const float* extra, const float* sc, const float* sh, float* frame, int B, int C,
                        hipStream_t st, int extra_ld) {
  if (extra_ld <= 0) extra_ld = C - Ca;
  const int64_t total = (int64_t)B * C * F_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_pad_frame_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, hm, Ca, extra, extra_ld, sc, sh, frame, C, total);
  return hipGetLastError();
}

// spec[b][p] = lhat[b][cond[p]] * phat[p] * scale       (scale = 1/(120*180): the transforms are unnormalised)
// two complex values (16 bytes) per thread; F_HWC = 120*91 is even
__global__ void sm_spec_mul_kernel(const float4* __restrict__ lhat, const float4* __restrict__ phat, const int* __restrict__ cond,
                                   float4* __restrict__ spec, int C, int P, float scale, int64_t total) {
  constexpr int HALF = F_HWC / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = i % HALF;
    int64_t r = i / HALF;
    const int p = r % P;
    const int64_t b = r / P;
    const float4 l = lhat[(b * C + cond[p]) * HALF + k];
    const float4 q = phat[(int64_t)p * HALF + k];
    float4 o;
    o.x = (l.x * q.x - l.y * q.y) * scale;
    o.y = (l.x * q.y + l.y * q.x) * scale;
    o.z = (l.z * q.z - l.w * q.w) * scale;
    o.w = (l.z * q.w + l.w * q.z) * scale;
    spec[i] = o;
  }
}
hipError_t sm_spec_mul(const float2* lhat, const float2* phat, const int* cond, float2* spec, int B, int C, int P, hipStream_t st) {
  static_assert(F_HWC % 2 == 0, "two complex values per thread");
  const int64_t total = (int64_t)B * P * (F_HWC / 2);
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_spec_mul_kernel, dim3((int)(g > 32768 ? 32768 : g)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(lhat), reinterpret_cast<const float4*>(phat), cond,
                     reinterpret_cast<float4*>(spec), C, P, 1.0f / (float)F_HW, total);
  return hipGetLastError();
}

// TF-1.x bilinear 61x91 -> 60x90 (main.py:89) sampled from the VALID window of a circular
// convolution frame: Cpre[y][x] = frame[59+y][89+x].
__device__ __forceinline__ float resize_from_frame(const float* __restrict__ fr, int oy, int ox) {
  const float sy = 61.0f / 60.0f, sx = 91.0f / 90.0f;
  const float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);
  const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  const int yhi = min(ylo + 1, 60), xhi = min(xlo + 1, 90);
  const float ty = fy - (float)ylo, tx = fx - (float)xlo;
  const float* w = fr + 59 * F_W + 89;
  const float tl = w[ylo * F_W + xlo], tr = w[ylo * F_W + xhi];
  const float bl = w[yhi * F_W + xlo], br = w[yhi * F_W + xhi];
  const float top = tl + (tr - tl) * tx;
  const float bot = bl + (br - bl) * tx;
  return top + (bot - top) * ty;
}

__global__ void sm_resize_frame_kernel(const float* __restrict__ cfull, float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % FM_HW;
    const int64_t b = i / FM_HW;
    out[i] = resize_from_frame(cfull + b * F_HW, pix / FM_W, pix % FM_W);
  }
}
hipError_t sm_resize_frame(const float* cfull, float* out, int B, hipStream_t st) {
  const int64_t total = (int64_t)B * FM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_resize_frame_kernel, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, st, cfull, out, total);
  return hipGetLastError();
}

}  // namespace jcm
