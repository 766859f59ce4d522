// Forward row pass of the frequency-domain convolution (see conv_fft.hip for the whole route).
#include <type_traits>

#include "conv_fft_common.h"
#include "resize_tf1.h"

namespace jcm {
namespace cfft {

// ---- rows, forward: NHWC fp32 / NHWC bf16 / planar bf16 [B][C/8][H*W][8] -> T[kx][c/16][b][y][16] complex, kx < NX/2+1
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar.  Two adjacent channels are one complex number.
// T is chunk-major: the (8 images x H rows x 16 channels) block a column work group transforms is one contiguous run, and this kernel
// writes it in whole 128-byte lines (8 lanes x float4 = the 16 channels of one (kx, chunk, image, row)).
// Persistent work groups with register prefetch: the kernel is latency-bound (a tile is 12-23 KB in, 25 KB out, three barriers), so the
// loads of a work group's NEXT (image, row, channel block) are issued before the FFT of the current one and land while it computes and stores.
template <int NX, int LAYOUT>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_fwd_kernel(const void* __restrict__ in, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H, int W, int C,
                                                                      int ntiles, float* __restrict__ tmax) {
  constexpr int CH = CB / 2, NTR = rows_threads<NX>(), K = (NX * CH + NTR - 1) / NTR;
  using Raw = std::conditional_t<LAYOUT == 0, cf, unsigned>;      // what a thread keeps per element: an fp32 channel pair, or two bf16 in one register
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float red[NTR / 64];
  const int tid = threadIdx.x, ncb = C / CB;
  twiddles<NX, NTR>(tw, twg, tid);
  Raw pre[K];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int cblk = tile % ncb, by = tile / ncb;
    const int y = by % H, b = by / H;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int t = tid + i * NTR, x = t / CH, v = t % CH;
      Raw r{};
      if (x < W) {      // (x < W implies t < NX * CH)
        if constexpr (LAYOUT == 0) {
          r = reinterpret_cast<const cf*>(static_cast<const float*>(in) + ((size_t)(b * H + y) * W + x) * C + cblk * CB)[v];
        } else if constexpr (LAYOUT == 1) {
          r = reinterpret_cast<const unsigned*>(static_cast<const __bf16*>(in) + ((size_t)(b * H + y) * W + x) * C + cblk * CB)[v];
        } else {
          const int c = cblk * CB + 2 * v;
          r = *reinterpret_cast<const unsigned*>(static_cast<const __bf16*>(in) + (((size_t)b * (C >> 3) + (c >> 3)) * H * W + (size_t)y * W + x) * 8 + (c & 7));
        }
      }
      pre[i] = r;
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  while (tile < ntiles) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int t = tid + i * NTR;
      if (t < NX * CH) {
        if constexpr (LAYOUT == 0) buf[t] = pre[i];
        else buf[t] = bf16pair(pre[i]);
      }
    }
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    __syncthreads();
    fft<NX, -1, CH, NTR>(buf, tw, tid);
    const int cblk = tile % ncb, by = tile / ncb;
    const float tm = rows_fwd_store<NX, NTR>(buf, T, tid, cblk, by / H, by % H, B, H, C);
    if (tmax) wave_max_stash(tm, red);
    __syncthreads();      // every wave is done reading buf
    if (tmax) stash_to_word<NTR>(red, tmax + by / H);      // the word of this tile's image
    tile = next;
  }
}

// ---- rows, forward, of the MERGED map (fp32 NHWC): x = ((x1 + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70; the arithmetic and association
// order of upsample_merge3_kernel) is formed while the row is loaded: the merged tensor never goes to HBM.
template <int NX>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_fwd_merge_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int H2, int W2,
                                                            const float* __restrict__ x3, int H3, int W3, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H,
                                                            int W, int C, float sy2, float sx2, float sy3, float sx3, float* __restrict__ tmax) {
  constexpr int CH = CB / 2, NTR = rows_threads<NX>();
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float red[NTR / 64];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX, NTR>(tw, twg, tid);
  const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);
  // 16-byte loads: a thread-iteration forms two complex inputs (4 consecutive channels) of one pixel
  const int C4 = C / 4;
  const float4* p1 = reinterpret_cast<const float4*>(x1 + ((size_t)(b * H + y) * W) * C + cblk * CB);
  const float4* p2 = reinterpret_cast<const float4*>(x2 + (size_t)b * H2 * W2 * C + cblk * CB);
  const float4* p3 = reinterpret_cast<const float4*>(x3 + (size_t)b * H3 * W3 * C + cblk * CB);
  auto bil = [&](const float4* p, int Wl, Tap ty, Tap tx, int v) __attribute__((always_inline)) {
    const float4 tl = p[((size_t)ty.lo * Wl + tx.lo) * C4 + v], tr = p[((size_t)ty.lo * Wl + tx.hi) * C4 + v];
    const float4 bl = p[((size_t)ty.hi * Wl + tx.lo) * C4 + v], br = p[((size_t)ty.hi * Wl + tx.hi) * C4 + v];
    return make_float4(lerp2(tl.x, tr.x, bl.x, br.x, tx.t, ty.t), lerp2(tl.y, tr.y, bl.y, br.y, tx.t, ty.t), lerp2(tl.z, tr.z, bl.z, br.z, tx.t, ty.t),
                       lerp2(tl.w, tr.w, bl.w, br.w, tx.t, ty.t));
  };
  constexpr int CQ = CH / 2;
  for (int t = tid; t < NX * CQ; t += NTR) {
    const int x = t / CQ, v = t % CQ;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < W) {
      const float4 a = p1[(size_t)x * C4 + v];
      const float4 u2 = (H2 == H && W2 == W) ? p2[((size_t)y * W + x) * C4 + v] : bil(p2, W2, ty2, tf1_tap(x, W2, sx2), v);
      const float4 u3 = (H3 == H && W3 == W) ? p3[((size_t)y * W + x) * C4 + v] : bil(p3, W3, ty3, tf1_tap(x, W3, sx3), v);
      z = make_float4(((a.x + u2.x) + u3.x) / 3.0f, ((a.y + u2.y) + u3.y) / 3.0f, ((a.z + u2.z) + u3.z) / 3.0f, ((a.w + u2.w) + u3.w) / 3.0f);
    }
    *reinterpret_cast<float4*>(&buf[x * CH + 2 * v]) = z;
  }
  __syncthreads();
  fft<NX, -1, CH, NTR>(buf, tw, tid);
  const float tm = rows_fwd_store<NX, NTR>(buf, T, tid, cblk, b, y, B, H, C);
  if (tmax) block_max_to<NTR>(tm, tmax + b, red, tid);      // one atomic per work group at most (skipped when the word already holds more)
}


template <int NX> static void launch_rows_fwd(const ConvArgs& a, int layout, cf* T, const cf* tw, float* tmax, hipStream_t st) {
  const int ntiles = a.B * a.H * (a.Cin / CB);
  const void* fn = layout == 0 ? reinterpret_cast<const void*>(rows_fwd_kernel<NX, 0>) : layout == 1 ? reinterpret_cast<const void*>(rows_fwd_kernel<NX, 1>)
                                                                                                  : reinterpret_cast<const void*>(rows_fwd_kernel<NX, 2>);
  const dim3 grid(persistent_grid(fn, ntiles, rows_threads<NX>()));
  if (layout == 0) hipLaunchKernelGGL((rows_fwd_kernel<NX, 0>), grid, dim3(rows_threads<NX>()), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin, ntiles, tmax);
  else if (layout == 1) hipLaunchKernelGGL((rows_fwd_kernel<NX, 1>), grid, dim3(rows_threads<NX>()), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin, ntiles, tmax);
  else hipLaunchKernelGGL((rows_fwd_kernel<NX, 2>), grid, dim3(rows_threads<NX>()), 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin, ntiles, tmax);
}
template <int NX> static void launch_rows_fwd_merge(const ConvArgs& a, const FftMerge& m, cf* T, const cf* tw, float* tmax, hipStream_t st) {
  hipLaunchKernelGGL(rows_fwd_merge_kernel<NX>, dim3(a.B * a.H * (a.Cin / CB)), dim3(rows_threads<NX>()), 0, st, static_cast<const float*>(a.x), m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, tw,
                     a.B, a.H, a.W, a.Cin, (float)m.H2 / (float)a.H, (float)m.W2 / (float)a.W, (float)m.H3 / (float)a.H, (float)m.W3 / (float)a.W, tmax);
}
void cfft_rows_fwd(int NX, const ConvArgs& a, int layout, cf* T, const cf* tw, float* tmax, hipStream_t st) {
#define CALL(N) launch_rows_fwd<N>(a, layout, T, tw, tmax, st)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}
void cfft_rows_fwd_merge(int NX, const ConvArgs& a, const FftMerge& m, cf* T, const cf* tw, float* tmax, hipStream_t st) {
#define CALL(N) launch_rows_fwd_merge<N>(a, m, T, tw, tmax, st)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}

}  // namespace cfft
}  // namespace jcm
