// Stride-1 SAME convolution (9x9, 5x5) in the FREQUENCY domain, fp32 arithmetic: every such layer of an fp32 handle, the wide 9x9
// layers (conv4_*, conv5: 256/512 -> 512 channels) of a bf16 handle.
//
// A 9x9 layer with 512 x 512 channels is 229 GFLOP per image as a direct convolution.  With the maps transformed once
// (NY x NX >= (H+8) x (W+8): a linear convolution, nothing wraps) the layer is, for every frequency, one complex matrix product
// over the channels:  Y[f][b][co] = sum_ci X[f][b][ci] * Wf[f][ci][co]  -- 8*Cin*Cout*NY*(NX/2+1) = 7.3 GFLOP per image for conv5
// (70 x 98 transform of the 60 x 90 maps), 31x fewer.  The transforms add 0.4 GFLOP.  In fp32 this route is MORE accurate than the
// fp32 MFMA accumulation chain it replaces (4e-7 of the output scale against 1.3e-6 for a sequential fp32 sum of the 41 472
// products: DESIGN.md 4.1c), so it runs behind the same parity tests.
//
//   rows_fwd   (image, row, 64 channels)  : NHWC fp32, two adjacent channels = one complex number z = x_c + i x_{c+1}; complex FFT
//                                           along x in LDS; X_c, X_{c+1} recovered through the Hermitian symmetry -> T[b][kx][y][ci]
//   cols_fwd   (image, kx, 64 channels)   : FFT along y -> Xf[ky][kx][b][ci]   (frequency-major: a GEMM operand per frequency)
//   rocBLAS cgemm_strided_batched         : the plain library GEMM, batch = NY * (NX/2+1) frequencies
//   cols_inv   (image, kx, 64 channels)   : inverse along ky, rows pad .. pad+H-1 kept -> T[b][y][kx][co]
//   rows_inv   (image, row, 64 channels)  : Z = Y_c + i Y_{c+1} (Hermitian extension), inverse complex FFT along kx, columns 4 .. W+3,
//                                           1/(NY NX), bias, ReLU, folded BatchNorm -> NHWC fp32
// The filter spectra Wf[f][ci][co] (flipped kernel: TF's conv2d is a correlation) are computed once per (layer, map size) at
// first use: 7.3 GB for conv5.  FFTs: the in-LDS decimation-in-frequency stages of sm_fused.hip, channel-vectorised
// (consecutive lanes = consecutive channels: every LDS and HBM access of a wave is one contiguous 512-byte run).
// Reference semantics: conv2d SAME stride 1 + bias + ReLU + BatchNorm (main.py:133-135,156-169).
#include <rocblas/rocblas.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "fft_lds.h"
#include "kernels.h"
#include "resize_tf1.h"

namespace jcm {

namespace cfft {
using namespace fftl;
constexpr int CB = 64, NT = 256;      // channels per work group; kernel sizes 9 and 5 (pad = (ks - 1) / 2 passed at launch)

// radix chains (decimation in frequency, in place): the output X[n] sits at pos(n)
template <int N> struct Plan;
template <> struct Plan<192> { static constexpr int R1 = 8, R2 = 8, R3 = 3; };
template <> struct Plan<128> { static constexpr int R1 = 8, R2 = 4, R3 = 4; };
template <> struct Plan<100> { static constexpr int R1 = 4, R2 = 5, R3 = 5; };
template <> struct Plan<98> { static constexpr int R1 = 7, R2 = 14, R3 = 1; };
template <> struct Plan<70> { static constexpr int R1 = 7, R2 = 10, R3 = 1; };
template <> struct Plan<72> { static constexpr int R1 = 8, R2 = 3, R3 = 3; };
template <> struct Plan<60> { static constexpr int R1 = 4, R2 = 15, R3 = 1; };
template <> struct Plan<40> { static constexpr int R1 = 8, R2 = 5, R3 = 1; };
template <> struct Plan<32> { static constexpr int R1 = 8, R2 = 4, R3 = 1; };
template <> struct Plan<24> { static constexpr int R1 = 8, R2 = 3, R3 = 1; };
template <int N> __device__ __forceinline__ int pos(int n) {
  using P = Plan<N>;
  if constexpr (P::R3 == 1) return (n % P::R1) * (N / P::R1) + n / P::R1;
  else return (n % P::R1) * (N / P::R1) + ((n / P::R1) % P::R2) * (N / (P::R1 * P::R2)) + n / (P::R1 * P::R2);
}

// one stage over CH channel lanes: buf[position][channel], tw[k] = e^{+2 pi i k / N}
template <int N, int R, int L, int S, int CH>
__device__ __forceinline__ void stage(cf* buf, const cf* tw, int tid) {
  constexpr int M = L / R, BF = N / R;
  for (int t = tid; t < BF * CH; t += NT) {
    const int bf = t / CH, v = t % CH;
    const int blk = bf / M, k = bf - blk * M;
    cf* p = buf + (blk * L + k) * CH + v;
    cf x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = p[m * M * CH];
    Dft<R, S>::run(x);
    if (M > 1) {
#pragma unroll
      for (int m = 1; m < R; ++m) {
        cf w = tw[(N / L) * k * m];
        if (S < 0) w.y = -w.y;
        x[m] = cmul(x[m], w);
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) p[m * M * CH] = x[m];
  }
}
template <int N, int S, int CH>
__device__ __forceinline__ void fft(cf* buf, const cf* tw, int tid) {
  using P = Plan<N>;
  stage<N, P::R1, N, S, CH>(buf, tw, tid);
  __syncthreads();
  stage<N, P::R2, N / P::R1, S, CH>(buf, tw, tid);
  __syncthreads();
  if constexpr (P::R3 > 1) {
    stage<N, P::R3, N / (P::R1 * P::R2), S, CH>(buf, tw, tid);
    __syncthreads();
  }
}
template <int N>
__device__ __forceinline__ void twiddles(cf* tw, int tid) {
  for (int k = tid; k < N; k += NT) {
    double sn, cs;
    sincospi(2.0 * (double)k / (double)N, &sn, &cs);
    tw[k] = cf{(float)cs, (float)sn};
  }
}

// ---- rows, forward: NHWC fp32 / NHWC bf16 / planar bf16 [B][C/8][H*W][8] -> T[b][kx][y][c] complex, kx < NX/2+1
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar.  Two adjacent channels are one complex number.
__device__ __forceinline__ cf bf16pair(unsigned bits) { return cf{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)}; }
template <int NX, int LAYOUT>
__global__ __launch_bounds__(NT) void rows_fwd_kernel(const void* __restrict__ in, cf* __restrict__ T, int H, int W, int C, int b0) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, bl = by / H, b = b0 + bl;       // bl: image inside this slice of the batch (indexes T), b: image of the batch
  twiddles<NX>(tw, tid);
  for (int t = tid; t < NX * CH; t += NT) {
    const int x = t / CH, v = t % CH;
    cf z = {0.f, 0.f};
    if (x < W) {
      if constexpr (LAYOUT == 0) {
        z = reinterpret_cast<const cf*>(static_cast<const float*>(in) + ((size_t)(b * H + y) * W + x) * C + cblk * CB)[v];
      } else if constexpr (LAYOUT == 1) {
        z = bf16pair(reinterpret_cast<const unsigned*>(static_cast<const __bf16*>(in) + ((size_t)(b * H + y) * W + x) * C + cblk * CB)[v]);
      } else {
        const int c = cblk * CB + 2 * v;
        z = bf16pair(*reinterpret_cast<const unsigned*>(static_cast<const __bf16*>(in) + (((size_t)b * (C >> 3) + (c >> 3)) * H * W + (size_t)y * W + x) * 8 + (c & 7)));
      }
    }
    buf[t] = z;
  }
  __syncthreads();
  fft<NX, -1, CH>(buf, tw, tid);
  // Z = FFT(x_c + i x_{c+1}):  X_c[k] = (Z[k] + conj Z[-k]) / 2,  X_{c+1}[k] = (Z[k] - conj Z[-k]) / (2i)
  float4* dst = reinterpret_cast<float4*>(T);
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    const cf zk = buf[pos<NX>(k) * CH + v], zn = buf[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
    dst[(((size_t)(bl * NXH + k) * H + y) * C + cblk * CB) / 2 + v] =
        make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
  }
}

// ---- rows, forward, of the MERGED map (fp32 NHWC): x = ((x1 + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70; the arithmetic and association
// order of upsample_merge3_kernel) is formed while the row is loaded: the merged tensor never goes to HBM.
template <int NX>
__global__ __launch_bounds__(NT) void rows_fwd_merge_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int H2, int W2,
                                                            const float* __restrict__ x3, int H3, int W3, cf* __restrict__ T, int H, int W, int C, float sy2,
                                                            float sx2, float sy3, float sx3) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, tid);
  const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);
  const int C2 = C / 2;
  const cf* p1 = reinterpret_cast<const cf*>(x1 + ((size_t)(b * H + y) * W) * C + cblk * CB);
  const cf* p2 = reinterpret_cast<const cf*>(x2 + (size_t)b * H2 * W2 * C + cblk * CB);
  const cf* p3 = reinterpret_cast<const cf*>(x3 + (size_t)b * H3 * W3 * C + cblk * CB);
  auto bil = [&](const cf* p, int Wl, Tap ty, Tap tx, int v) __attribute__((always_inline)) {
    const cf tl = p[((size_t)ty.lo * Wl + tx.lo) * C2 + v], tr = p[((size_t)ty.lo * Wl + tx.hi) * C2 + v];
    const cf bl = p[((size_t)ty.hi * Wl + tx.lo) * C2 + v], br = p[((size_t)ty.hi * Wl + tx.hi) * C2 + v];
    return cf{lerp2(tl.x, tr.x, bl.x, br.x, tx.t, ty.t), lerp2(tl.y, tr.y, bl.y, br.y, tx.t, ty.t)};
  };
  for (int t = tid; t < NX * CH; t += NT) {
    const int x = t / CH, v = t % CH;
    cf z = {0.f, 0.f};
    if (x < W) {
      const cf a = p1[(size_t)x * C2 + v];
      const cf u2 = (H2 == H && W2 == W) ? p2[((size_t)y * W + x) * C2 + v] : bil(p2, W2, ty2, tf1_tap(x, W2, sx2), v);
      const cf u3 = (H3 == H && W3 == W) ? p3[((size_t)y * W + x) * C2 + v] : bil(p3, W3, ty3, tf1_tap(x, W3, sx3), v);
      z = cf{((a.x + u2.x) + u3.x) / 3.0f, ((a.y + u2.y) + u3.y) / 3.0f};
    }
    buf[t] = z;
  }
  __syncthreads();
  fft<NX, -1, CH>(buf, tw, tid);
  float4* dst = reinterpret_cast<float4*>(T);
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    const cf zk = buf[pos<NX>(k) * CH + v], zn = buf[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
    dst[(((size_t)(b * NXH + k) * H + y) * C + cblk * CB) / 2 + v] =
        make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
  }
}

// ---- columns, forward: T[b][kx][y][c] -> Xf[ky][kx][b][c]
template <int NY> constexpr int colblk() { return NY > 100 ? 32 : 64; }      // channels per work group of the column kernels (<= 64 KB of LDS)
template <int NY>
__global__ __launch_bounds__(NT) void cols_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Xf, int B, int H, int NXH, int C, int b0, int kx0, int nkx) {
  constexpr int CH = colblk<NY>(), CB = CH;
  __shared__ cf buf[NY * CH];
  __shared__ cf tw[NY];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), bk = blockIdx.x / (C / CB);
  const int kx = kx0 + bk % nkx, bl = bk / nkx, b = b0 + bl;
  twiddles<NY>(tw, tid);
  const cf* src = T + ((size_t)(bl * NXH + kx) * H) * C + cblk * CB;
  for (int t = tid; t < NY * CH; t += NT) {
    const int y = t / CH, v = t % CH;
    buf[t] = y < H ? src[(size_t)y * C + v] : cf{0.f, 0.f};
  }
  __syncthreads();
  fft<NY, -1, CH>(buf, tw, tid);
  for (int t = tid; t < NY * CH; t += NT) {
    const int ky = t / CH, v = t % CH;
    Xf[((size_t)(kx * NY + ky) * B + b) * C + cblk * CB + v] = buf[pos<NY>(ky) * CH + v];
  }
}

// ---- columns, inverse: Yf[ky][kx][b][c] -> T[b][y][kx][c], y < H (row y of the output is row y + pad of the linear convolution)
template <int NY>
__global__ __launch_bounds__(NT) void cols_inv_kernel(const cf* __restrict__ Yf, cf* __restrict__ T, int B, int H, int NXH, int C, int pad, int b0, int kx0,
                                                      int nkx) {
  constexpr int CH = colblk<NY>(), CB = CH;
  __shared__ cf buf[NY * CH];
  __shared__ cf tw[NY];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), bk = blockIdx.x / (C / CB);
  const int kx = kx0 + bk % nkx, bl = bk / nkx, b = b0 + bl;
  twiddles<NY>(tw, tid);
  for (int t = tid; t < NY * CH; t += NT) {
    const int ky = t / CH, v = t % CH;
    buf[t] = Yf[((size_t)(kx * NY + ky) * B + b) * C + cblk * CB + v];
  }
  __syncthreads();
  fft<NY, 1, CH>(buf, tw, tid);
  for (int t = tid; t < H * CH; t += NT) {
    const int y = t / CH, v = t % CH;
    T[((size_t)(bl * H + y) * NXH + kx) * C + cblk * CB + v] = buf[pos<NY>(y + pad) * CH + v];
  }
}

// ---- rows, inverse + epilogue: T[b][y][kx][c] (C channels, padded to a multiple of 64) -> out with Cout channels
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar (Cout % 8 == 0)
template <int NX, int LAYOUT>
__global__ __launch_bounds__(NT) void rows_inv_kernel(const cf* __restrict__ T, void* __restrict__ out, const float* __restrict__ bias,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int H, int W, int C,
                                                      int Cout, int pad, float norm, int b0) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, bl = by / H, b = b0 + bl;
  twiddles<NX>(tw, tid);
  const float4* src = reinterpret_cast<const float4*>(T + ((size_t)(bl * H + y) * NXH) * C + cblk * CB);
  // Z = Y_c + i Y_{c+1} with the Hermitian extension Y[NX - k] = conj Y[k]; DC and Nyquist are real by symmetry
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    float4 q = src[(size_t)k * (C / 2) + v];            // (Ya.re, Ya.im, Yb.re, Yb.im)
    const bool edge = k == 0 || k == NX / 2;
    if (edge) { q.y = 0.f; q.w = 0.f; }
    buf[k * CH + v] = cf{q.x - q.w, q.y + q.z};
    if (!edge) buf[(NX - k) * CH + v] = cf{q.x + q.w, q.z - q.y};
  }
  __syncthreads();
  fft<NX, 1, CH>(buf, tw, tid);
  // a thread keeps its channel pair for the whole row (NT is a multiple of CH): bias / scale / shift are loaded once
  const int v = tid % CH, c = cblk * CB + 2 * v;
  if (c < Cout) {
    const bool two = c + 1 < Cout, pairs = (Cout & 1) == 0;      // two channels = one aligned store
    const float b0v = bias[c], b1v = two ? bias[c + 1] : 0.f;
    float s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
    if (relu_bn) { s0 = scale[c]; h0 = shift[c]; if (two) { s1 = scale[c + 1]; h1 = shift[c + 1]; } }
    for (int x = tid / CH; x < W; x += NT / CH) {
      const cf z = buf[pos<NX>(x + pad) * CH + v];
      float v0 = z.x * norm + b0v, v1 = z.y * norm + b1v;
      if (relu_bn) { v0 = fmaxf(v0, 0.f) * s0 + h0; v1 = fmaxf(v1, 0.f) * s1 + h1; }
      if constexpr (LAYOUT == 0) {
        float* o = static_cast<float*>(out) + ((size_t)(b * H + y) * W + x) * Cout + c;
        if (pairs) *reinterpret_cast<cf*>(o) = cf{v0, v1};
        else { o[0] = v0; if (two) o[1] = v1; }
      } else {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        __bf16* o = static_cast<__bf16*>(out) + (LAYOUT == 1 ? ((size_t)(b * H + y) * W + x) * Cout + c
                                                             : (((size_t)b * (Cout >> 3) + (c >> 3)) * H * W + (size_t)y * W + x) * 8 + (c & 7));
        if (pairs) *reinterpret_cast<bf16x2*>(o) = bf16x2{static_cast<__bf16>(v0), static_cast<__bf16>(v1)};
        else { o[0] = static_cast<__bf16>(v0); if (two) o[1] = static_cast<__bf16>(v1); }
      }
    }
  }
}

// ---- rows, inverse of layer L + epilogue + rows, forward of layer L+1 in one kernel (fp32 handles; same map, same NX, C % 64 == 0):
// the activation between two frequency-domain layers never goes to HBM.  T_in[b][y][kx][c] -> T_out[b][kx][y][c].
// The epilogue's result IS the next layer's packed input: channel pair (c, c+1) = one complex number.
template <int NX>
__global__ __launch_bounds__(NT) void rows_inv_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Tn, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int H, int W, int C,
                                                          int pad, float norm) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf nxt[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, tid);
  const float4* src = reinterpret_cast<const float4*>(T + ((size_t)(b * H + y) * NXH) * C + cblk * CB);
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    float4 q = src[(size_t)k * (C / 2) + v];
    const bool edge = k == 0 || k == NX / 2;
    if (edge) { q.y = 0.f; q.w = 0.f; }
    buf[k * CH + v] = cf{q.x - q.w, q.y + q.z};
    if (!edge) buf[(NX - k) * CH + v] = cf{q.x + q.w, q.z - q.y};
  }
  __syncthreads();
  fft<NX, 1, CH>(buf, tw, tid);
  {
    const int v = tid % CH, c = cblk * CB + 2 * v;
    const float b0v = bias[c], b1v = bias[c + 1];
    float s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
    if (relu_bn) { s0 = scale[c]; h0 = shift[c]; s1 = scale[c + 1]; h1 = shift[c + 1]; }
    for (int x = tid / CH; x < NX; x += NT / CH) {
      cf o = {0.f, 0.f};
      if (x < W) {
        const cf z = buf[pos<NX>(x + pad) * CH + v];
        float v0 = z.x * norm + b0v, v1 = z.y * norm + b1v;
        if (relu_bn) { v0 = fmaxf(v0, 0.f) * s0 + h0; v1 = fmaxf(v1, 0.f) * s1 + h1; }
        o = cf{v0, v1};
      }
      nxt[x * CH + v] = o;
    }
  }
  __syncthreads();
  fft<NX, -1, CH>(nxt, tw, tid);
  float4* dst = reinterpret_cast<float4*>(Tn);
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    const cf zk = nxt[pos<NX>(k) * CH + v], zn = nxt[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
    dst[(((size_t)(b * NXH + k) * H + y) * C + cblk * CB) / 2 + v] =
        make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
  }
}

// ---- filter spectra: HWIO fp32 [k][k][Cin][Cout] -> Wf[ky][kx][ci][co < CoutP] = sum_{a,b} w[k-1-a][k-1-b][ci][co] e^{-2 pi i (ky a / NY + kx b / NX)}
// (the flipped kernel: TF's conv2d is a correlation; channels Cout .. CoutP-1 are zero)
template <int KS>
__global__ __launch_bounds__(256) void weight_spectra_kernel(const float* __restrict__ w, cf* __restrict__ Wf, int Cin, int Cout, int CoutP, int NY, int NX,
                                                             int round_bf16) {
  __shared__ cf twy[256], twx[256];
  const int tid = threadIdx.x;
  for (int k = tid; k < NY + NX; k += 256) {
    const bool isy = k < NY;
    const int kk = isy ? k : k - NY;
    double sn, cs;
    sincospi(-2.0 * (double)kk / (double)(isy ? NY : NX), &sn, &cs);
    (isy ? twy : twx)[kk] = cf{(float)cs, (float)sn};
  }
  __syncthreads();
  const size_t io = (size_t)blockIdx.x * 256 + tid;           // ci * CoutP + co
  if (io >= (size_t)Cin * CoutP) return;
  const int ci = (int)(io / CoutP), co = (int)(io - (size_t)ci * CoutP);
  float g[KS][KS];                                            // flipped kernel
#pragma unroll
  for (int a = 0; a < KS; ++a)
#pragma unroll
    for (int b = 0; b < KS; ++b) {
      float wv = co < Cout ? w[(((size_t)((KS - 1 - a) * KS + (KS - 1 - b))) * Cin + ci) * Cout + co] : 0.f;
      if (round_bf16) wv = static_cast<float>(static_cast<__bf16>(wv));       // bf16 handles: the filter the bf16 MFMA kernels multiply with
      g[a][b] = wv;
    }
  const int NXH = NX / 2 + 1;
  for (int kx = blockIdx.y; kx < NXH; kx += gridDim.y) {
    cf ra[KS];
#pragma unroll
    for (int a = 0; a < KS; ++a) {
      cf s = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < KS; ++b) s = sfma(g[a][b], twx[(kx * b) % NX], s);
      ra[a] = s;
    }
    for (int ky = 0; ky < NY; ++ky) {
      cf s = ra[0];
#pragma unroll
      for (int a = 1; a < KS; ++a) s = s + cmul(ra[a], twy[(ky * a) % NY]);
      Wf[(size_t)(kx * NY + ky) * Cin * CoutP + io] = s;
    }
  }
}

struct Sizes { int NY, NX; };
static bool pick(int need, int* n) {
  static const int ok[] = {24, 32, 40, 60, 70, 72, 98, 100, 128, 192};      // 70 x 98 is the exact linear-convolution size of a 9x9 kernel on the 60x90 maps
  for (int v : ok)
    if (v >= need) { *n = v; return true; }
  return false;
}
static bool sizes_of(int H, int W, int ks, Sizes* s) { return (ks == 9 || ks == 5) && pick(H + ks - 1, &s->NY) && pick(W + ks - 1, &s->NX); }
static int pad64(int c) { return (c + CB - 1) / CB * CB; }

// (b0, nb): the slice of the batch a launch covers; T is the slice's scratch
template <int NX> static void launch_rows_fwd(const ConvArgs& a, int layout, cf* T, int b0, int nb, hipStream_t st) {
  const dim3 grid(nb * a.H * (a.Cin / CB));
  if (layout == 0) hipLaunchKernelGGL((rows_fwd_kernel<NX, 0>), grid, dim3(NT), 0, st, a.x, T, a.H, a.W, a.Cin, b0);
  else if (layout == 1) hipLaunchKernelGGL((rows_fwd_kernel<NX, 1>), grid, dim3(NT), 0, st, a.x, T, a.H, a.W, a.Cin, b0);
  else hipLaunchKernelGGL((rows_fwd_kernel<NX, 2>), grid, dim3(NT), 0, st, a.x, T, a.H, a.W, a.Cin, b0);
}
// (kx0, nkx): the range of frequency columns a launch covers
template <int NY> static void launch_cols_fwd(const ConvArgs& a, const cf* T, cf* Xf, int NXH, int b0, int nb, int kx0, int nkx, hipStream_t st) {
  hipLaunchKernelGGL(cols_fwd_kernel<NY>, dim3(nb * nkx * (a.Cin / colblk<NY>())), dim3(NT), 0, st, T, Xf, a.B, a.H, NXH, a.Cin, b0, kx0, nkx);
}
template <int NY> static void launch_cols_inv(const ConvArgs& a, const cf* Yf, cf* T, int NXH, int pad, int b0, int nb, int kx0, int nkx, hipStream_t st) {
  hipLaunchKernelGGL(cols_inv_kernel<NY>, dim3(nb * nkx * (a.CoutP / colblk<NY>())), dim3(NT), 0, st, Yf, T, a.B, a.H, NXH, a.CoutP, pad, b0, kx0, nkx);
}
template <int NX> static void launch_rows_inv(const ConvArgs& a, int layout, const cf* T, int pad, float norm, int b0, int nb, hipStream_t st) {
  const dim3 grid(nb * a.H * (a.CoutP / CB));
  if (layout == 0)
    hipLaunchKernelGGL((rows_inv_kernel<NX, 0>), grid, dim3(NT), 0, st, T, a.out, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm, b0);
  else if (layout == 1)
    hipLaunchKernelGGL((rows_inv_kernel<NX, 1>), grid, dim3(NT), 0, st, T, a.out, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm, b0);
  else
    hipLaunchKernelGGL((rows_inv_kernel<NX, 2>), grid, dim3(NT), 0, st, T, a.out, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm, b0);
}
template <int NX> static void launch_rows_inv_fwd(const ConvArgs& a, const cf* T, cf* Tn, int pad, float norm, hipStream_t st) {
  hipLaunchKernelGGL(rows_inv_fwd_kernel<NX>, dim3(a.B * a.H * (a.Cout / CB)), dim3(NT), 0, st, T, Tn, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.Cout, pad, norm);
}
template <int NX> static void launch_rows_fwd_merge(const ConvArgs& a, const FftMerge& m, cf* T, hipStream_t st) {
  hipLaunchKernelGGL(rows_fwd_merge_kernel<NX>, dim3(a.B * a.H * (a.Cin / CB)), dim3(NT), 0, st, static_cast<const float*>(a.x), m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, a.H,
                     a.W, a.Cin, (float)m.H2 / (float)a.H, (float)m.W2 / (float)a.W, (float)m.H3 / (float)a.H, (float)m.W3 / (float)a.W);
}
// Images per slice of the transform passes.  JCM_FFT_SLICE_MB=n keeps the row-transformed tensor T of a slice below n MB (so that it
// could stay in the 256 MB Infinity Cache between the row and the column kernel); measured at 160 / 96 / 48 MB: 2.6 / 2.1 / 9 % SLOWER
// than the whole batch in one launch (fp32 B=64; bf16 B=256: 3.9 %), so the default is 0 = whole batch.
static int slice_images(const ConvArgs& a, const Sizes& s) {
  static const long long cap = [] { const char* e = std::getenv("JCM_FFT_SLICE_MB"); return (long long)(e ? std::atoi(e) : 0) << 20; }();
  const size_t cmax = a.Cin > a.CoutP ? a.Cin : a.CoutP;
  const long long per = (long long)(s.NX / 2 + 1) * a.H * cmax * sizeof(cf);
  if (cap <= 0) return a.B;
  const long long n = cap / per;
  return n < 1 ? 1 : n > a.B ? a.B : (int)n;
}
#define CFFT_BY_SIZE(N, CALL)                    \
  switch (N) {                                   \
    case 24: CALL(24); break;                    \
    case 32: CALL(32); break;                    \
    case 40: CALL(40); break;                    \
    case 60: CALL(60); break;                    \
    case 70: CALL(70); break;                    \
    case 72: CALL(72); break;                    \
    case 98: CALL(98); break;                    \
    case 100: CALL(100); break;                  \
    case 128: CALL(128); break;                  \
    default: CALL(192); break;                   \
  }

// one rocBLAS handle per (device, stream), shared by the engines of the process and created on first use: a handle owns scratch
// memory, so concurrent streams (the three resolution branches) each need their own
static rocblas_handle blas_for(int dev, hipStream_t st) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, rocblas_handle> hs;
  std::lock_guard<std::mutex> lk(mu);
  auto it = hs.find({dev, st});
  if (it != hs.end()) return it->second;
  rocblas_handle h = nullptr;
  if (rocblas_create_handle(&h) != rocblas_status_success) return nullptr;
  if (rocblas_set_stream(h, st) != rocblas_status_success) { rocblas_destroy_handle(h); return nullptr; }
  hs[{dev, st}] = h;
  return h;
}
// A side stream and four events per (device, main stream): the column passes of one half of the frequency columns beside the GEMM of the
// other half (the GEMM is MFMA-bound, the column kernels HBM-bound).  EXPERIMENT, off unless JCM_FFT_OVERLAP=1: measured 16.94 against
// 16.58 ms per fp32 step and 45.15 against 45.59 ms per bf16 step -- the GEMM fills every CU and the halves cost it more than the
// overlap returns.
struct Side { hipStream_t s = nullptr; hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr}; };
static Side* side_for(int dev, hipStream_t st) {
  static const bool on = [] { const char* e = std::getenv("JCM_FFT_OVERLAP"); return e && e[0] == '1'; }();
  if (!on) return nullptr;
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Side> sides;
  std::lock_guard<std::mutex> lk(mu);
  Side& sd = sides[{dev, st}];
  if (!sd.s) {
    if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess) { sd.s = nullptr; return nullptr; }
    for (auto& ev : sd.e)
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &sd;
}
}  // namespace cfft

using namespace cfft;

bool conv_fft_supported(const ConvArgs& a, int ks) {
  Sizes s;
  return a.Cin % CB == 0 && a.Cin >= CB && a.Cout >= 1 && a.B >= 1 && sizes_of(a.H, a.W, ks, &s);
}
size_t conv_fft_weight_bytes(int H, int W, int ks, int Cin, int Cout) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s)) return 0;
  return (size_t)s.NY * (s.NX / 2 + 1) * Cin * pad64(Cout) * sizeof(cf);
}
hipError_t conv_fft_pack_weights(const float* w_hwio, void* wf, int H, int W, int ks, int Cin, int Cout, bool round_bf16, hipStream_t st) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s)) return hipErrorInvalidValue;
  const int CoutP = pad64(Cout);
  const dim3 grid((unsigned)(((size_t)Cin * CoutP + 255) / 256), 8);
  if (ks == 9) hipLaunchKernelGGL(weight_spectra_kernel<9>, grid, dim3(256), 0, st, w_hwio, static_cast<cf*>(wf), Cin, Cout, CoutP, s.NY, s.NX, round_bf16 ? 1 : 0);
  else hipLaunchKernelGGL(weight_spectra_kernel<5>, grid, dim3(256), 0, st, w_hwio, static_cast<cf*>(wf), Cin, Cout, CoutP, s.NY, s.NX, round_bf16 ? 1 : 0);
  return hipGetLastError();
}
// scratch: T (the larger of the two row-transformed tensors) + Xf + Yf
size_t conv_fft_workspace_bytes(const ConvArgs& a, int ks) {
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s)) return 0;
  const size_t NXH = s.NX / 2 + 1, cop = pad64(a.Cout), cmax = (size_t)a.Cin > cop ? a.Cin : cop;
  ConvArgs ap = a;
  ap.CoutP = (int)cop;
  return ((size_t)slice_images(ap, s) * NXH * a.H * cmax + (size_t)s.NY * NXH * a.B * (a.Cin + cop)) * sizeof(cf);
}
// Can layer L (a, ks) hand its output to layer L+1 (kernel size ks_next, same map) in row-transformed form?  Same NX for both kernel
// sizes, unpadded channel count, whole batch in one pass.
bool conv_fft_fusable(const ConvArgs& a, int ks, int ks_next) {
  Sizes s, n;
  if (!sizes_of(a.H, a.W, ks, &s) || !sizes_of(a.H, a.W, ks_next, &n) || s.NX != n.NX || s.NX > 100 || a.Cout % CB) return false;   // (two row buffers in LDS)
  ConvArgs ap = a;
  ap.CoutP = a.Cout;
  return slice_images(ap, s) == a.B;
}
size_t conv_fft_handover_bytes(const ConvArgs& a, int ks) {      // T[b][kx][y][Cout] of the next layer
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s)) return 0;
  return (size_t)a.B * (s.NX / 2 + 1) * a.H * a.Cout * sizeof(cf);
}
// a.wp = the filter spectra of THIS map size and kernel size; `work` = conv_fft_workspace_bytes(a, ks) bytes.  g0 / g1: optional events
// recorded around the GEMM (the dominant kernel of the layer) for the roofline record.
// in_layout / out_layout: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar
// t_in (fp32 handles): the row-transformed input T[b][kx][y][ci] left by the previous layer's fused kernel -- the forward row pass is skipped;
// t_next: write the NEXT layer's row-transformed input there instead of the spatial output (conv_fft_fusable() says when that is legal).
hipError_t conv_fft_f32(const ConvArgs& a0, int ks, int in_layout, int out_layout, void* work, const void* t_in, void* t_next, const FftMerge* merge,
                        hipEvent_t g0, hipEvent_t g1, hipStream_t st) {
  Sizes s;
  if (!conv_fft_supported(a0, ks) || !sizes_of(a0.H, a0.W, ks, &s) || (out_layout == 2 && a0.Cout % 8)) return hipErrorInvalidValue;
  ConvArgs a = a0;
  a.CoutP = pad64(a.Cout);
  const int pad = (ks - 1) / 2;
  const int NXH = s.NX / 2 + 1, F = s.NY * NXH;
  const size_t cmax = a.Cin > a.CoutP ? a.Cin : a.CoutP;
  const int SL = (t_in || t_next || merge) ? a.B : slice_images(a, s);      // the fused hand-over covers the whole batch
  if ((t_in || t_next || merge) && (in_layout != 0 || out_layout != 0 || (t_next && a.Cout % CB))) return hipErrorInvalidValue;
  cf* T = static_cast<cf*>(work);
  cf* Xf = T + (size_t)SL * NXH * a.H * cmax;
  cf* Yf = Xf + (size_t)F * a.B * a.Cin;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  rocblas_handle bh = blas_for(dev, st);
  if (!bh) return hipErrorUnknown;
  const float norm = 1.0f / (float)(s.NY * s.NX);
  const cf* Tin = t_in ? static_cast<const cf*>(t_in) : T;
  auto gemm = [&](int kx0, int nkx) -> bool {      // the frequency columns kx0 .. kx0 + nkx - 1: nkx * NY matrix products
    const rocblas_float_complex one{1.f, 0.f}, zero{0.f, 0.f};
    const size_t f0 = (size_t)kx0 * s.NY;
    return rocblas_cgemm_strided_batched(bh, rocblas_operation_none, rocblas_operation_none, a.CoutP, a.B, a.Cin, &one,
                                         static_cast<const rocblas_float_complex*>(a.wp) + f0 * a.Cin * a.CoutP, a.CoutP, (rocblas_stride)a.Cin * a.CoutP,
                                         reinterpret_cast<const rocblas_float_complex*>(Xf) + f0 * a.B * a.Cin, a.Cin, (rocblas_stride)a.B * a.Cin, &zero,
                                         reinterpret_cast<rocblas_float_complex*>(Yf) + f0 * a.B * a.CoutP, a.CoutP, (rocblas_stride)a.B * a.CoutP,
                                         nkx * s.NY) == rocblas_status_success;
  };
  Side* sd = SL == a.B && NXH >= 8 ? side_for(dev, st) : nullptr;
  if (sd) {
    // ---- two halves of the frequency columns, the column passes of one beside the GEMM of the other:
    //   main: rows_fwd | cols_fwd(H0) GEMM(H0) ........ GEMM(H1) cols_inv(H1) | rows_inv
    //   side:          | cols_fwd(H1) ......... cols_inv(H0)                  |
    const int k1 = NXH / 2, n1 = NXH - k1;
    if (merge && !t_in) {
#define CALL(N) launch_rows_fwd_merge<N>(a, *merge, T, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    } else if (!t_in) {
#define CALL(N) launch_rows_fwd<N>(a, in_layout, T, 0, a.B, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    }
    if (hipEventRecord(sd->e[0], st) != hipSuccess || hipStreamWaitEvent(sd->s, sd->e[0], 0) != hipSuccess) return hipErrorUnknown;
#define CALL(N) launch_cols_fwd<N>(a, Tin, Xf, NXH, 0, a.B, k1, n1, sd->s)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (hipEventRecord(sd->e[1], sd->s) != hipSuccess) return hipErrorUnknown;
#define CALL(N) launch_cols_fwd<N>(a, Tin, Xf, NXH, 0, a.B, 0, k1, st)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (g0 && hipEventRecord(g0, st) != hipSuccess) return hipErrorUnknown;
    if (!gemm(0, k1)) return hipErrorUnknown;
    if (hipEventRecord(sd->e[2], st) != hipSuccess || hipStreamWaitEvent(sd->s, sd->e[2], 0) != hipSuccess) return hipErrorUnknown;
    // the side stream has finished its cols_fwd (same stream) and the main stream's cols_fwd(H0) precedes e[2]: nobody reads T any more
#define CALL(N) launch_cols_inv<N>(a, Yf, T, NXH, pad, 0, a.B, 0, k1, sd->s)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (hipEventRecord(sd->e[3], sd->s) != hipSuccess) return hipErrorUnknown;
    if (hipStreamWaitEvent(st, sd->e[1], 0) != hipSuccess) return hipErrorUnknown;
    if (!gemm(k1, n1)) return hipErrorUnknown;
    if (g1 && hipEventRecord(g1, st) != hipSuccess) return hipErrorUnknown;
#define CALL(N) launch_cols_inv<N>(a, Yf, T, NXH, pad, 0, a.B, k1, n1, st)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (hipStreamWaitEvent(st, sd->e[3], 0) != hipSuccess) return hipErrorUnknown;
    if (t_next) {
#define CALL(N) launch_rows_inv_fwd<N>(a, T, static_cast<cf*>(t_next), pad, norm, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    } else {
#define CALL(N) launch_rows_inv<N>(a, out_layout, T, pad, norm, 0, a.B, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    }
    return hipGetLastError();
  }
  // ---- one stream
  for (int b0 = 0; b0 < a.B; b0 += SL) {
    const int nb = a.B - b0 < SL ? a.B - b0 : SL;
    if (merge && !t_in) {
#define CALL(N) launch_rows_fwd_merge<N>(a, *merge, T, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    } else if (!t_in) {
#define CALL(N) launch_rows_fwd<N>(a, in_layout, T, b0, nb, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    }
#define CALL(N) launch_cols_fwd<N>(a, Tin, Xf, NXH, b0, nb, 0, NXH, st)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  if (g0 && hipEventRecord(g0, st) != hipSuccess) return hipErrorUnknown;
  if (!gemm(0, NXH)) return hipErrorUnknown;
  if (g1 && hipEventRecord(g1, st) != hipSuccess) return hipErrorUnknown;
  for (int b0 = 0; b0 < a.B; b0 += SL) {
    const int nb = a.B - b0 < SL ? a.B - b0 : SL;
#define CALL(N) launch_cols_inv<N>(a, Yf, T, NXH, pad, b0, nb, 0, NXH, st)
    CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
    if (t_next) {
#define CALL(N) launch_rows_inv_fwd<N>(a, T, static_cast<cf*>(t_next), pad, norm, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    } else {
#define CALL(N) launch_rows_inv<N>(a, out_layout, T, pad, norm, b0, nb, st)
      CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace jcm
