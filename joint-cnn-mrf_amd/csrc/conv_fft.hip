// Stride-1 SAME convolution (9x9, 5x5) in the FREQUENCY domain: every such layer of an fp32 handle, the wide 9x9 layers (conv4_*, conv5:
// 256/512 -> 512 channels) of a bf16 handle.
//
// A 9x9 layer with 512 x 512 channels is 229 GFLOP per image as a direct convolution.  With the maps transformed once the layer is, for
// every frequency, one complex matrix product over the channels:  Y[f][b][co] = sum_ci X[f][b][ci] * Wf[f][ci][co].  The transform is a
// CIRCULAR convolution of size NY x NX >= (H + pad) x (W + pad), pad = (k-1)/2: the SAME output y in [0, H) reads inputs y-pad .. y+pad,
// so the wrap-around only has to land in the zero rows H .. NY-1 -- which H + pad rows guarantee (a full linear convolution would need
// H + k - 1).  60x90 maps: 64 x 96 transforms, 64 * 49 = 3136 frequencies, 8*Cin*Cout*3136 = 6.6 GFLOP per image for conv5, 35x fewer
// than the direct form.  In fp32 the route is MORE accurate than the fp32 MFMA accumulation chain it replaces.
//
//   rows_fwd        (image, row, 64 channels)        : NHWC fp32 / bf16 (or planar bf16); two adjacent channels = one complex number
//                                                      z = x_c + i x_{c+1}; complex FFT along x in LDS; X_c, X_{c+1} through the Hermitian
//                                                      symmetry -> T[kx][c/16][b][y][16]   (a column work group's input is contiguous)
//   cols_fwd_split  (8 images, kx, 16 channels)      : FFT along y, then the spectra are SPLIT into bf16 parts and written in the exact
//                                                      LDS image of the channel GEMM: Xs[f][m-tile][c/16][re|im][part][k-half][row][8]
//   cgemm_split     (cgemm_split.hip)                : the channel GEMM on the bf16 matrix cores, one complex product per frequency
//   cols_inv        (image, kx, 64 channels)         : inverse along ky, rows pad .. pad+H-1 kept -> T[b][y][kx][co]
//   rows_inv        (image, row, 64 channels)        : Z = Y_c + i Y_{c+1} (Hermitian extension), inverse complex FFT along kx, columns
//                                                      pad .. pad+W-1, 1/(NY NX), bias, ReLU, folded BatchNorm -> NHWC fp32 / bf16 / planar
// The filter spectra (flipped kernel: TF's conv2d is a correlation) are computed once per (layer, map size) at first use, split into the
// same bf16 parts, in the GEMM's tile-major layout.  FFTs: the in-LDS decimation-in-frequency stages of sm_fused.hip, channel-vectorised
// (consecutive lanes = consecutive channels).  Twiddles come from one table per device, built on the host in double precision.
// Reference semantics: conv2d SAME stride 1 + bias + ReLU + BatchNorm (main.py:133-135,156-169).
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

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
template <> struct Plan<96> { static constexpr int R1 = 8, R2 = 12, R3 = 1; };
template <> struct Plan<72> { static constexpr int R1 = 8, R2 = 3, R3 = 3; };
template <> struct Plan<64> { static constexpr int R1 = 8, R2 = 8, R3 = 1; };
template <> struct Plan<60> { static constexpr int R1 = 4, R2 = 15, R3 = 1; };
template <> struct Plan<50> { static constexpr int R1 = 5, R2 = 10, R3 = 1; };
template <> struct Plan<40> { static constexpr int R1 = 8, R2 = 5, R3 = 1; };
template <> struct Plan<36> { static constexpr int R1 = 4, R2 = 3, R3 = 3; };
template <> struct Plan<32> { static constexpr int R1 = 8, R2 = 4, R3 = 1; };
template <> struct Plan<28> { static constexpr int R1 = 4, R2 = 7, R3 = 1; };
template <> struct Plan<24> { static constexpr int R1 = 8, R2 = 3, R3 = 1; };
template <> struct Plan<20> { static constexpr int R1 = 4, R2 = 5, R3 = 1; };
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
// tw[k] = e^{+2 pi i k / N} from the per-device table (host-built in double precision; tw_offset(N) entries in)
template <int N>
__device__ __forceinline__ void twiddles(cf* tw, const cf* __restrict__ twg, int tid) {
  for (int k = tid; k < N; k += NT) tw[k] = twg[k];
}

// ---- rows, forward: NHWC fp32 / NHWC bf16 / planar bf16 [B][C/8][H*W][8] -> T[kx][c/16][b][y][16] complex, kx < NX/2+1
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar.  Two adjacent channels are one complex number.
// T is chunk-major: the (8 images x H rows x 16 channels) block a column work group transforms is one contiguous run, and this kernel
// writes it in whole 128-byte lines (8 lanes x float4 = the 16 channels of one (kx, chunk, image, row)).
__device__ __forceinline__ cf bf16pair(unsigned bits) { return cf{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)}; }
// index, in float4 = two channels, of channel pair v (0..31) of 64-channel block cblk
__device__ __forceinline__ size_t t_fwd_index(int k, int cblk, int v, int b, int y, int B, int H, int C) {
  return ((((size_t)k * (C >> 4) + cblk * 4 + (v >> 3)) * B + b) * H + y) * 8 + (v & 7);
}
// Z = FFT(x_c + i x_{c+1}):  X_c[k] = (Z[k] + conj Z[-k]) / 2,  X_{c+1}[k] = (Z[k] - conj Z[-k]) / (2i)
template <int NX>
__device__ __forceinline__ void rows_fwd_store(const cf* buf, cf* __restrict__ T, int tid, int cblk, int b, int y, int B, int H, int C) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  float4* dst = reinterpret_cast<float4*>(T);
  for (int t = tid; t < NXH * CH; t += NT) {
    const int k = t / CH, v = t % CH;
    const cf zk = buf[pos<NX>(k) * CH + v], zn = buf[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
    dst[t_fwd_index(k, cblk, v, b, y, B, H, C)] = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
  }
}
template <int NX, int LAYOUT>
__global__ __launch_bounds__(NT) void rows_fwd_kernel(const void* __restrict__ in, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H, int W, int C) {
  constexpr int CH = CB / 2;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, twg, tid);
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
  rows_fwd_store<NX>(buf, T, tid, cblk, b, y, B, H, C);
}

// ---- rows, forward, of the MERGED map (fp32 NHWC): x = ((x1 + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70; the arithmetic and association
// order of upsample_merge3_kernel) is formed while the row is loaded: the merged tensor never goes to HBM.
template <int NX>
__global__ __launch_bounds__(NT) void rows_fwd_merge_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int H2, int W2,
                                                            const float* __restrict__ x3, int H3, int W3, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H,
                                                            int W, int C, float sy2, float sx2, float sy3, float sx3) {
  constexpr int CH = CB / 2;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, twg, tid);
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
  rows_fwd_store<NX>(buf, T, tid, cblk, b, y, B, H, C);
}

// ---- columns, forward + operand split: T[kx][c/16][b][y][16] -> Xs[f = kx NY + ky][m-tile][c/16][re|im][part][k-half][row][8] bf16
// One work group = (IMG images, kx, one 16-channel chunk): its input is one contiguous run of T; after the FFT along y every spectrum is
// split into NP bf16 parts (x = x0 + x1 (+ x2), each rounded to nearest: exact for NP = 3, 16 significant bits for NP = 2) and stored as
// 16-byte MFMA operand units -- 8 consecutive channels of one image -- with the units of the work group's IMG images consecutive: 128-byte
// lines for IMG = 8.  The result is the channel GEMM's LDS image (cgemm_split.hip), which that kernel fetches by LDS-DMA.
template <int NY> constexpr int colimg() { return NY > 96 ? 4 : 8; }         // images per work group (LDS: NY * IMG * 16 complex numbers)
template <int NY> constexpr int colblk() { return NY > 100 ? 32 : 64; }      // channels per work group of the inverse column kernel (<= 64 KB of LDS)
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], uint4 (&out)[NP]) {
  unsigned short h[NP][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = x[e];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const __bf16 q = static_cast<__bf16>(v);           // round to nearest even
      h[p][e] = __builtin_bit_cast(unsigned short, q);
      v = v - static_cast<float>(q);                      // exact
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
    out[p] = make_uint4((unsigned)h[p][0] | ((unsigned)h[p][1] << 16), (unsigned)h[p][2] | ((unsigned)h[p][3] << 16), (unsigned)h[p][4] | ((unsigned)h[p][5] << 16),
                        (unsigned)h[p][6] | ((unsigned)h[p][7] << 16));
}
template <int NY, int NP>
__global__ __launch_bounds__(NT) void cols_fwd_split_kernel(const cf* __restrict__ T, uint4* __restrict__ Xs, const cf* __restrict__ twg, int B, int H, int KC, int MT,
                                                            int mtiles) {
  constexpr int IMG = colimg<NY>(), CH = IMG * 16;
  extern __shared__ __attribute__((aligned(16))) char smem_cf[];
  cf* buf = reinterpret_cast<cf*>(smem_cf);
  cf* tw = buf + NY * CH;
  const int tid = threadIdx.x;
  const int NG = (B + IMG - 1) / IMG;
  const int g = blockIdx.x % NG, kk = blockIdx.x / NG;
  const int kc = kk % KC, kx = kk / KC;
  const int b0 = g * IMG, nimg = min(IMG, B - b0);
  twiddles<NY>(tw, twg, tid);
  const cf* src = T + (((size_t)kx * KC + kc) * B + b0) * H * 16;
  for (int t = tid; t < NY * CH; t += NT) {
    const int y = t / CH, v = t % CH, img = v >> 4, c = v & 15;
    buf[t] = (y < H && img < nimg) ? src[((size_t)img * H + y) * 16 + c] : cf{0.f, 0.f};
  }
  __syncthreads();
  fft<NY, -1, CH>(buf, tw, tid);
  // item = (ky, k-half, image): 8 complex numbers -> NP units of the real parts + NP units of the imaginary parts
  const int mt = b0 / MT, r0 = b0 - mt * MT;
  for (int it = tid; it < NY * 2 * IMG; it += NT) {
    const int img = it % IMG, kg = (it / IMG) & 1, ky = it / (2 * IMG);
    const cf* z = buf + pos<NY>(ky) * CH + img * 16 + kg * 8;
    float re[8], im[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { re[e] = z[e].x; im[e] = z[e].y; }
    uint4 ur[NP], ui[NP];
    split8<NP>(re, ur);
    split8<NP>(im, ui);
    const size_t f = (size_t)kx * NY + ky;
    uint4* dst = Xs + (((f * mtiles + mt) * KC + kc) * (4 * NP) + kg) * MT + r0 + img;      // unit ((c * NP + p) * 2 + kg) * MT + row
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      dst[(size_t)(0 * NP + p) * 2 * MT] = ur[p];
      dst[(size_t)(1 * NP + p) * 2 * MT] = ui[p];
    }
  }
}

// ---- columns, inverse: Yf[ky][kx][b][ldy channels] -> T[b][y][kx][c < C], y < H (row y of the output is row y + pad of the circular convolution)
template <int NY>
__global__ __launch_bounds__(NT) void cols_inv_kernel(const cf* __restrict__ Yf, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H, int NXH, int C, int ldy,
                                                      int pad) {
  constexpr int CH = colblk<NY>(), CB = CH;
  __shared__ cf buf[NY * CH];
  __shared__ cf tw[NY];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), bk = blockIdx.x / (C / CB);
  const int kx = bk % NXH, b = bk / NXH;
  twiddles<NY>(tw, twg, tid);
  for (int t = tid; t < NY * CH; t += NT) {
    const int ky = t / CH, v = t % CH;
    buf[t] = Yf[((size_t)(kx * NY + ky) * B + b) * ldy + cblk * CB + v];
  }
  __syncthreads();
  fft<NY, 1, CH>(buf, tw, tid);
  for (int t = tid; t < H * CH; t += NT) {
    const int y = t / CH, v = t % CH;
    T[((size_t)(b * H + y) * NXH + kx) * C + cblk * CB + v] = buf[pos<NY>(y + pad) * CH + v];
  }
}

// ---- rows, inverse + epilogue: T[b][y][kx][c] (C channels, padded to a multiple of 64) -> out with Cout channels
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar (Cout % 8 == 0)
template <int NX, int LAYOUT>
__global__ __launch_bounds__(NT) void rows_inv_kernel(const cf* __restrict__ T, void* __restrict__ out, const cf* __restrict__ twg, const float* __restrict__ bias,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int H, int W, int C,
                                                      int Cout, int pad, float norm) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, twg, tid);
  const float4* src = reinterpret_cast<const float4*>(T + ((size_t)(b * H + y) * NXH) * C + cblk * CB);
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
__global__ __launch_bounds__(NT) void rows_inv_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Tn, const cf* __restrict__ twg, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int B, int H, int W, int C,
                                                          int pad, float norm) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  __shared__ cf buf[NX * CH];
  __shared__ cf nxt[NX * CH];
  __shared__ cf tw[NX];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX>(tw, twg, tid);
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
  rows_fwd_store<NX>(nxt, Tn, tid, cblk, b, y, B, H, C);
}

// ---- filter spectra, split: HWIO fp32 [k][k][Cin][Cout] -> Wf[f][ci][co] = sum_{a,b} w[k-1-a][k-1-b][ci][co] e^{-2 pi i (ky a / NY + kx b / NX)}
// (the flipped kernel: TF's conv2d is a correlation; output channels Cout .. CoutP-1 are zero), written as the channel GEMM's operand
// image Ws[f][co/128][ci/16][re|im][part][k-half][128 columns][8 bf16] (cgemm_split.hip).  A work group owns 8 input channels (one
// k-half) x 32 output channels: thread (ci, co) evaluates the separable 81- (25-) term DFT of its filter for one frequency after the
// other, the 8 x 32 spectra meet in LDS, and 64 threads split them and store 16-byte units (32 columns = 512 contiguous bytes).
template <int KS, int NP>
__global__ __launch_bounds__(256) void weight_spectra_split_kernel(const float* __restrict__ w, uint4* __restrict__ Ws, int Cin, int Cout, int CoutP, int NY, int NX,
                                                                   int round_bf16) {
  __shared__ cf twy[256], twx[256];
  __shared__ cf sp[32][9];
  const int tid = threadIdx.x;
  for (int k = tid; k < NY + NX; k += 256) {
    const bool isy = k < NY;
    const int kk = isy ? k : k - NY;
    double sn, cs;
    sincospi(-2.0 * (double)kk / (double)(isy ? NY : NX), &sn, &cs);
    (isy ? twy : twx)[kk] = cf{(float)cs, (float)sn};
  }
  __syncthreads();
  const int nco = CoutP / 32;
  const int ci8 = blockIdx.x / nco, co32 = blockIdx.x % nco;
  const int cil = tid & 7, col = tid >> 3;
  const int ci = ci8 * 8 + cil, co = co32 * 32 + col;
  float g[KS][KS];                                            // flipped kernel
#pragma unroll
  for (int a = 0; a < KS; ++a)
#pragma unroll
    for (int b = 0; b < KS; ++b) {
      float wv = co < Cout ? w[(((size_t)((KS - 1 - a) * KS + (KS - 1 - b))) * Cin + ci) * Cout + co] : 0.f;
      if (round_bf16) wv = static_cast<float>(static_cast<__bf16>(wv));       // bf16 handles: the filter the bf16 MFMA kernels multiply with
      g[a][b] = wv;
    }
  const int NXH = NX / 2 + 1, KC = Cin / 16, ntiles = CoutP / kCgemmNT;
  const int kc = ci8 >> 1, kg = ci8 & 1;
  // the 64 storing threads: (column, re|im)
  const int scol = tid & 31, sc = (tid >> 5) & 1;
  const int sco = co32 * 32 + scol, snt = sco / kCgemmNT, sn = sco % kCgemmNT;
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
      sp[col][cil] = s;
      __syncthreads();
      if (tid < 64) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sc ? sp[scol][e].y : sp[scol][e].x;
        uint4 u[NP];
        split8<NP>(x, u);
        const size_t f = (size_t)kx * NY + ky;
        uint4* dst = Ws + (((f * ntiles + snt) * KC + kc) * (4 * NP) + (sc * NP) * 2 + kg) * kCgemmNT + sn;
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[(size_t)p * 2 * kCgemmNT] = u[p];
      }
      __syncthreads();
    }
  }
}

struct Sizes { int NY, NX; };
// transform lengths with a radix plan (all even: the row pass packs two real rows into one complex transform and needs a Nyquist bin)
static const int kLens[] = {20, 24, 28, 32, 36, 40, 50, 60, 64, 72, 96, 100, 128, 192};
static bool pick(int need, int* n) {
  for (int v : kLens)
    if (v >= need) { *n = v; return true; }
  return false;
}
static int tw_offset(int n) {
  int off = 0;
  for (int v : kLens) {
    if (v == n) return off;
    off += v;
  }
  return -1;
}
// circular convolution of size >= (H + pad) x (W + pad); one size per map for every kernel size (pad of the 9x9 layers), so that two
// consecutive layers can hand the row-transformed tensor over.  The old limit H + k - 1 <= 192 is kept.
static bool sizes_of(int H, int W, int ks, Sizes* s) {
  return (ks == 9 || ks == 5) && H + ks - 1 <= 192 && W + ks - 1 <= 192 && pick(H + 4, &s->NY) && pick(W + 4, &s->NX);
}
static int pad64(int c) { return (c + CB - 1) / CB * CB; }
static int pad128(int c) { return (c + kCgemmNT - 1) / kCgemmNT * kCgemmNT; }

// e^{+2 pi i k / n} for every supported length, one table per device (built on the host in double precision, uploaded at first use)
static const cf* twiddle_table(int dev) {
  static std::mutex mu;
  static std::map<int, cf*> tabs;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find(dev);
  if (it != tabs.end()) return it->second;
  std::vector<float> h;
  for (int n : kLens)
    for (int k = 0; k < n; ++k) {
      const double ang = 2.0 * 3.14159265358979323846 * (double)k / (double)n;
      h.push_back((float)std::cos(ang));
      h.push_back((float)std::sin(ang));
    }
  cf* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  tabs[dev] = d;
  return d;
}

template <int NX> static void launch_rows_fwd(const ConvArgs& a, int layout, cf* T, const cf* tw, hipStream_t st) {
  const dim3 grid(a.B * a.H * (a.Cin / CB));
  if (layout == 0) hipLaunchKernelGGL((rows_fwd_kernel<NX, 0>), grid, dim3(NT), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin);
  else if (layout == 1) hipLaunchKernelGGL((rows_fwd_kernel<NX, 1>), grid, dim3(NT), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin);
  else hipLaunchKernelGGL((rows_fwd_kernel<NX, 2>), grid, dim3(NT), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin);
}
template <int NY> static hipError_t launch_cols_fwd(const ConvArgs& a, int np, const cf* T, void* Xs, const cf* tw, int NXH, int MT, hipStream_t st) {
  constexpr int IMG = colimg<NY>();
  constexpr int lds = (NY * IMG * 16 + NY) * (int)sizeof(cf);
  const int KC = a.Cin / 16, mtiles = (a.B + MT - 1) / MT;
  const dim3 grid((unsigned)(NXH * KC * ((a.B + IMG - 1) / IMG)));
  static LdsAttr attr2, attr3;
  if (np == 2) {
    if (hipError_t e = attr2.ensure(reinterpret_cast<const void*>(cols_fwd_split_kernel<NY, 2>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((cols_fwd_split_kernel<NY, 2>), grid, dim3(NT), lds, st, T, static_cast<uint4*>(Xs), tw, a.B, a.H, KC, MT, mtiles);
  } else {
    if (hipError_t e = attr3.ensure(reinterpret_cast<const void*>(cols_fwd_split_kernel<NY, 3>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((cols_fwd_split_kernel<NY, 3>), grid, dim3(NT), lds, st, T, static_cast<uint4*>(Xs), tw, a.B, a.H, KC, MT, mtiles);
  }
  return hipSuccess;
}
// a.CoutP = output channels the inverse passes transform (Cout padded to 64); ldy = channel stride of Yf (Cout padded to the GEMM's N tile)
template <int NY> static void launch_cols_inv(const ConvArgs& a, const cf* Yf, cf* T, const cf* tw, int NXH, int ldy, int pad, hipStream_t st) {
  hipLaunchKernelGGL(cols_inv_kernel<NY>, dim3(a.B * NXH * (a.CoutP / colblk<NY>())), dim3(NT), 0, st, Yf, T, tw, a.B, a.H, NXH, a.CoutP, ldy, pad);
}
template <int NX> static void launch_rows_inv(const ConvArgs& a, int layout, const cf* T, const cf* tw, int pad, float norm, hipStream_t st) {
  const dim3 grid(a.B * a.H * (a.CoutP / CB));
  if (layout == 0)
    hipLaunchKernelGGL((rows_inv_kernel<NX, 0>), grid, dim3(NT), 0, st, T, a.out, tw, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm);
  else if (layout == 1)
    hipLaunchKernelGGL((rows_inv_kernel<NX, 1>), grid, dim3(NT), 0, st, T, a.out, tw, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm);
  else
    hipLaunchKernelGGL((rows_inv_kernel<NX, 2>), grid, dim3(NT), 0, st, T, a.out, tw, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm);
}
template <int NX> static void launch_rows_inv_fwd(const ConvArgs& a, const cf* T, cf* Tn, const cf* tw, int pad, float norm, hipStream_t st) {
  hipLaunchKernelGGL(rows_inv_fwd_kernel<NX>, dim3(a.B * a.H * (a.Cout / CB)), dim3(NT), 0, st, T, Tn, tw, a.bias, a.scale, a.shift, a.relu_bn, a.B, a.H, a.W, a.Cout, pad, norm);
}
template <int NX> static void launch_rows_fwd_merge(const ConvArgs& a, const FftMerge& m, cf* T, const cf* tw, hipStream_t st) {
  hipLaunchKernelGGL(rows_fwd_merge_kernel<NX>, dim3(a.B * a.H * (a.Cin / CB)), dim3(NT), 0, st, static_cast<const float*>(a.x), m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, tw,
                     a.B, a.H, a.W, a.Cin, (float)m.H2 / (float)a.H, (float)m.W2 / (float)a.W, (float)m.H3 / (float)a.H, (float)m.W3 / (float)a.W);
}
#define CFFT_BY_SIZE(N, CALL)                    \
  switch (N) {                                   \
    case 20: CALL(20); break;                    \
    case 24: CALL(24); break;                    \
    case 28: CALL(28); break;                    \
    case 32: CALL(32); break;                    \
    case 36: CALL(36); break;                    \
    case 40: CALL(40); break;                    \
    case 50: CALL(50); break;                    \
    case 60: CALL(60); break;                    \
    case 64: CALL(64); break;                    \
    case 72: CALL(72); break;                    \
    case 96: CALL(96); break;                    \
    case 100: CALL(100); break;                  \
    case 128: CALL(128); break;                  \
    default: CALL(192); break;                   \
  }
}  // namespace cfft

using namespace cfft;

bool conv_fft_supported(const ConvArgs& a, int ks) {
  Sizes s;
  return a.Cin % CB == 0 && a.Cin >= CB && a.Cout >= 1 && a.B >= 1 && sizes_of(a.H, a.W, ks, &s);
}
// np: bf16 parts per operand (3 = fp32 handles, 2 = bf16 handles)
size_t conv_fft_weight_bytes(int H, int W, int ks, int Cin, int Cout, int np) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s)) return 0;
  return (size_t)s.NY * (s.NX / 2 + 1) * Cin * pad128(Cout) * 4 * np;      // 2 (re, im) x np parts x 2 bytes per element
}
hipError_t conv_fft_pack_weights(const float* w_hwio, void* wf, int H, int W, int ks, int Cin, int Cout, int np, bool round_bf16, hipStream_t st) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s) || Cin % 16 || (np != 2 && np != 3)) return hipErrorInvalidValue;
  const int CoutP = pad128(Cout);
  const dim3 grid((unsigned)((Cin / 8) * (CoutP / 32)), 8);
  uint4* dst = static_cast<uint4*>(wf);
  const int rb = round_bf16 ? 1 : 0;
  if (ks == 9 && np == 2) hipLaunchKernelGGL((weight_spectra_split_kernel<9, 2>), grid, dim3(256), 0, st, w_hwio, dst, Cin, Cout, CoutP, s.NY, s.NX, rb);
  else if (ks == 9) hipLaunchKernelGGL((weight_spectra_split_kernel<9, 3>), grid, dim3(256), 0, st, w_hwio, dst, Cin, Cout, CoutP, s.NY, s.NX, rb);
  else if (np == 2) hipLaunchKernelGGL((weight_spectra_split_kernel<5, 2>), grid, dim3(256), 0, st, w_hwio, dst, Cin, Cout, CoutP, s.NY, s.NX, rb);
  else hipLaunchKernelGGL((weight_spectra_split_kernel<5, 3>), grid, dim3(256), 0, st, w_hwio, dst, Cin, Cout, CoutP, s.NY, s.NX, rb);
  return hipGetLastError();
}
// scratch: T (the larger of the two row-transformed tensors) + the split activation spectra Xs + the product spectra Yf
namespace {
struct Plan3 { size_t t_bytes, xs_bytes, yf_bytes; int MT, NXH, F, ldy; };
Plan3 plan_of(const ConvArgs& a, const Sizes& s, int np) {
  Plan3 p;
  p.NXH = s.NX / 2 + 1;
  p.F = s.NY * p.NXH;
  p.ldy = pad128(a.Cout);
  p.MT = cgemm_split_mtile(np, a.B);
  const size_t cop = pad64(a.Cout), cmax = (size_t)a.Cin > cop ? a.Cin : cop;
  const size_t bp = (size_t)(a.B + p.MT - 1) / p.MT * p.MT;
  p.t_bytes = (size_t)a.B * p.NXH * a.H * cmax * sizeof(cf);
  p.xs_bytes = (size_t)p.F * bp * a.Cin * 4 * np;
  p.yf_bytes = (size_t)p.F * a.B * p.ldy * sizeof(cf);
  return p;
}
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}  // namespace
size_t conv_fft_workspace_bytes(const ConvArgs& a, int ks, int np) {
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s)) return 0;
  const Plan3 p = plan_of(a, s, np);
  return align256(p.t_bytes) + align256(p.xs_bytes) + align256(p.yf_bytes);
}
// Can layer L (a, ks) hand its output to layer L+1 (kernel size ks_next, same map) in row-transformed form?  Same NX for both kernel
// sizes (always: the size depends on the map only), unpadded channel count, two row buffers in LDS.
bool conv_fft_fusable(const ConvArgs& a, int ks, int ks_next) {
  Sizes s, n;
  return sizes_of(a.H, a.W, ks, &s) && sizes_of(a.H, a.W, ks_next, &n) && s.NX == n.NX && s.NX <= 100 && a.Cout % CB == 0;
}
size_t conv_fft_handover_bytes(const ConvArgs& a, int ks) {      // T[kx][c/16][b][y][16] of the next layer
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s)) return 0;
  return (size_t)a.B * (s.NX / 2 + 1) * a.H * a.Cout * sizeof(cf);
}
// a.wp = the split filter spectra of THIS map size and kernel size; `work` = conv_fft_workspace_bytes(a, ks, np) bytes.  g0 / g1: optional
// events recorded around the GEMM (the dominant kernel of the layer) for the roofline record.
// in_layout / out_layout: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar
// t_in (fp32 handles): the row-transformed input left by the previous layer's fused kernel -- the forward row pass is skipped;
// t_next: write the NEXT layer's row-transformed input there instead of the spatial output (conv_fft_fusable() says when that is legal).
hipError_t conv_fft_f32(const ConvArgs& a0, int ks, int np, int in_layout, int out_layout, void* work, const void* t_in, void* t_next, const FftMerge* merge,
                        hipEvent_t g0, hipEvent_t g1, hipStream_t st) {
  Sizes s;
  if (!conv_fft_supported(a0, ks) || !sizes_of(a0.H, a0.W, ks, &s) || (out_layout == 2 && a0.Cout % 8) || (np != 2 && np != 3)) return hipErrorInvalidValue;
  if ((t_in || t_next || merge) && (in_layout != 0 || out_layout != 0 || (t_next && a0.Cout % CB))) return hipErrorInvalidValue;
  ConvArgs a = a0;
  a.CoutP = pad64(a.Cout);
  const int opad = (ks - 1) / 2;                         // output row y = row y + pad of the circular convolution (whose size is H + 4 for both kernel sizes)
  const Plan3 p = plan_of(a, s, np);
  char* wk = static_cast<char*>(work);
  cf* T = reinterpret_cast<cf*>(wk);
  void* Xs = wk + align256(p.t_bytes);
  cf* Yf = reinterpret_cast<cf*>(wk + align256(p.t_bytes) + align256(p.xs_bytes));
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const cf* twb = twiddle_table(dev);
  if (!twb) return hipErrorOutOfMemory;
  const cf* twx = twb + tw_offset(s.NX);
  const cf* twy = twb + tw_offset(s.NY);
  const float norm = 1.0f / (float)(s.NY * s.NX);
  const cf* Tin = t_in ? static_cast<const cf*>(t_in) : T;
  if (merge && !t_in) {
#define CALL(N) launch_rows_fwd_merge<N>(a, *merge, T, twx, st)
    CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
  } else if (!t_in) {
#define CALL(N) launch_rows_fwd<N>(a, in_layout, T, twx, st)
    CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
  }
  hipError_t ce = hipSuccess;
#define CALL(N) ce = launch_cols_fwd<N>(a, np, Tin, Xs, twy, p.NXH, p.MT, st)
  CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
  if (ce != hipSuccess) return ce;
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (g0 && hipEventRecord(g0, st) != hipSuccess) return hipErrorUnknown;
  if (hipError_t e = cgemm_split(Xs, a.wp, Yf, np, p.F, a.B, a.Cin, p.ldy, st); e != hipSuccess) return e;
  if (g1 && hipEventRecord(g1, st) != hipSuccess) return hipErrorUnknown;
#define CALL(N) launch_cols_inv<N>(a, Yf, T, twy, p.NXH, p.ldy, opad, st)
  CFFT_BY_SIZE(s.NY, CALL)
#undef CALL
  if (t_next) {
#define CALL(N) launch_rows_inv_fwd<N>(a, T, static_cast<cf*>(t_next), twx, opad, norm, st)
    CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
  } else {
#define CALL(N) launch_rows_inv<N>(a, out_layout, T, twx, opad, norm, st)
    CFFT_BY_SIZE(s.NX, CALL)
#undef CALL
  }
  return hipGetLastError();
}

}  // namespace jcm
