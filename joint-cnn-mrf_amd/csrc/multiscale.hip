// Multi-scale test-time evaluation helpers (main.py:326-379): the
// pad-or-crop window + skimage.transform.resize (0.13.x defaults: bilinear, half-pixel
// centres, zeros outside, clip to the input's [min,max]) that the reference applies to every
// image (8 rescaled copies) and to every heat map on the way back, and the mean over the 8
// copies.  HBM-bound; coordinates and weights in double so that they match the float64
// arithmetic of the host library the reference calls.
#include "kernels.h"

namespace jcm {

// windows[w] = (src index, y0, x0, h, w): rows/cols outside the source image read as 0
// (np.lib.pad(..., 'constant', 0), main.py:331,367); inside, it is a crop (main.py:339,359).
__device__ __forceinline__ float win_px(const float* __restrict__ s, int H, int W, int C, int y0, int x0, int wh, int ww,
                                        int r, int cc, int ch) {
  if (r < 0 || r > wh - 1 || cc < 0 || cc > ww - 1) return 0.f;      // outside the window: cval
  const int gy = y0 + r, gx = x0 + cc;
  if (gy < 0 || gy >= H || gx < 0 || gx >= W) return 0.f;            // the padded zeros
  return s[((size_t)gy * W + gx) * C + ch];
}

__global__ __launch_bounds__(256) void window_minmax_kernel(const float* __restrict__ src, int H, int W, int C,
                                                            const int* __restrict__ windows, float2* __restrict__ mm) {
  __shared__ float smn[4], smx[4];
  const int* wv = windows + blockIdx.x * 5;
  const int si = wv[0], y0 = wv[1], x0 = wv[2], wh = wv[3], ww = wv[4];
  const float* s = src + (size_t)si * H * W * C;
  float mn = INFINITY, mx = -INFINITY;
  const int total = wh * ww * C;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int ch = i % C, p = i / C;
    const float v = win_px(s, H, W, C, y0, x0, wh, ww, p / ww, p % ww, ch);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0)
    mm[blockIdx.x] = make_float2(fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3])), fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3])));
}

__global__ void window_resize_kernel(const float* __restrict__ src, int H, int W, int C, const int* __restrict__ windows,
                                     const float2* __restrict__ mm, int OH, int OW, float* __restrict__ out, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % C;
    size_t r = i / C;
    const int ox = r % OW; r /= OW;
    const int oy = r % OH;
    const int wi = (int)(r / OH);
    const int* wv = windows + wi * 5;
    const int si = wv[0], y0 = wv[1], x0 = wv[2], wh = wv[3], ww = wv[4];
    const float* s = src + (size_t)si * H * W * C;
    const double fr = ((double)wh / (double)OH) * ((double)oy + 0.5) - 0.5;
    const double fc = ((double)ww / (double)OW) * ((double)ox + 0.5) - 0.5;
    const int r0 = (int)floor(fr), r1 = (int)ceil(fr), c0 = (int)floor(fc), c1 = (int)ceil(fc);
    const double dr = fr - (double)r0, dc = fc - (double)c0;
    const double top = (1.0 - dc) * (double)win_px(s, H, W, C, y0, x0, wh, ww, r0, c0, ch) + dc * (double)win_px(s, H, W, C, y0, x0, wh, ww, r0, c1, ch);
    const double bot = (1.0 - dc) * (double)win_px(s, H, W, C, y0, x0, wh, ww, r1, c0, ch) + dc * (double)win_px(s, H, W, C, y0, x0, wh, ww, r1, c1, ch);
    double v = (1.0 - dr) * top + dr * bot;
    const float2 m = mm[wi];
    const bool preserve = !(m.x <= 0.f && 0.f <= m.y);       // cval = 0 outside [min, max]: exact zeros survive the clip
    if (!(preserve && v == 0.0)) v = fmin(fmax(v, (double)m.x), (double)m.y);
    out[i] = (float)v;
  }
}

hipError_t window_resize(const float* src, int H, int W, int C, const int* windows_dev, int NW, float2* mm_scratch,
                         int OH, int OW, float* out, hipStream_t st) {
  hipLaunchKernelGGL(window_minmax_kernel, dim3(NW), dim3(256), 0, st, src, H, W, C, windows_dev, mm_scratch);
  const size_t total = (size_t)NW * OH * OW * C;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(window_resize_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, src, H, W, C, windows_dev,
                     mm_scratch, OH, OW, out, total);
  return hipGetLastError();
}

// out[i][m] = mean over g < G of in[i*G + g][m]      (np.average(..., axis=0), main.py:413-414)
__global__ void group_mean_kernel(const float* __restrict__ in, float* __restrict__ out, int G, size_t M, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i % M, n = i / M;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += (double)in[(n * G + g) * M + m];
    out[i] = (float)(s / (double)G);
  }
}
hipError_t group_mean(const float* in, float* out, int n, int G, size_t M, hipStream_t st) {
  const size_t total = (size_t)n * M;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(group_mean_kernel, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, st, in, out, G, M, total);
  return hipGetLastError();
}

}  // namespace jcm
