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
// T16: T is written as complex fp16 in block floating point, one scale word per tile in t16 (conv_fft_common.h).
template <int NX, int LAYOUT, bool T16 = false>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_fwd_kernel(const void* __restrict__ in, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H, int W, int C,
                                                                      int ntiles, float* __restrict__ tmax, float* __restrict__ t16) {
  constexpr int CH = CB / 2, NTR = rows_threads<NX>(), K = (NX * CH + NTR - 1) / NTR;
  using Raw = std::conditional_t<LAYOUT == 0, cf, unsigned>;      // what a thread keeps per element: an fp32 channel pair, or two bf16 in one register
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float red[NTR / 64];
  __shared__ float red2[NTR / 64];
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
    float tm;
    if constexpr (T16) tm = rows_fwd_store16<NX, NTR>(buf, T, t16, red2, tid, cblk, by / H, by % H, B, H, C);
    else tm = rows_fwd_store<NX, NTR>(buf, T, tid, cblk, by / H, by % H, B, H, C);
    if (tmax) wave_max_stash(tm, red);
    __syncthreads();      // every wave is done reading buf
    if (tmax) stash_to_word<NTR>(red, tmax + by / H);      // the word of this tile's image
    tile = next;
  }
}

// ---- rows, forward, of the MERGED map (NHWC fp32, or NHWC bf16 on bf16 handles): x = ((x1 + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70; the
// arithmetic and association order of upsample_merge3_kernel, rounded to bf16 on a bf16 handle exactly as that kernel's output is) is
// formed while the row is loaded: the merged tensor never goes to HBM.  16-byte loads: 4 fp32 or 8 bf16 channels per thread-iteration.
template <bool BF> struct MergeVec;
template <> struct MergeVec<false> {
  static constexpr int N = 4;
  typedef float4 Raw;
  __device__ static void unpack(const Raw& r, float (&f)[4]) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
};
template <> struct MergeVec<true> {
  static constexpr int N = 8;
  typedef uint4 Raw;
  __device__ static void unpack(const Raw& r, float (&f)[8]) {
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
  }
};
template <int NX, bool BF, bool T16 = false>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_fwd_merge_kernel(const void* __restrict__ x1, const void* __restrict__ x2, int H2, int W2,
                                                            const void* __restrict__ x3, int H3, int W3, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H,
                                                            int W, int C, float sy2, float sx2, float sy3, float sx3, float* __restrict__ tmax, float* __restrict__ t16) {
  using V = MergeVec<BF>;
  using Raw = typename V::Raw;
  constexpr int CH = CB / 2, NTR = rows_threads<NX>(), VN = V::N, CQ = CB / VN;      // CQ items per pixel of the 64-channel block
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float red[NTR / 64];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), by = blockIdx.x / (C / CB);
  const int y = by % H, b = by / H;
  twiddles<NX, NTR>(tw, twg, tid);
  const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);
  const int CV = C / VN;
  const Raw* p1 = static_cast<const Raw*>(x1) + ((size_t)(b * H + y) * W) * CV + cblk * CQ;
  const Raw* p2 = static_cast<const Raw*>(x2) + (size_t)b * H2 * W2 * CV + cblk * CQ;
  const Raw* p3 = static_cast<const Raw*>(x3) + (size_t)b * H3 * W3 * CV + cblk * CQ;
  // the two source rows of each coarse map are the same for the whole tile: their base pointers are formed once, an item indexes them with 32-bit offsets
  const Raw* r2lo = p2 + (size_t)ty2.lo * W2 * CV;
  const Raw* r2hi = p2 + (size_t)ty2.hi * W2 * CV;
  const Raw* r3lo = p3 + (size_t)ty3.lo * W3 * CV;
  const Raw* r3hi = p3 + (size_t)ty3.hi * W3 * CV;
  auto bil = [&](const Raw* rlo, const Raw* rhi, float tyt, Tap tx, int v, float (&o)[VN]) __attribute__((always_inline)) {
    float tl[VN], tr[VN], bl[VN], br[VN];
    const unsigned ilo = (unsigned)(tx.lo * CV + v), ihi = (unsigned)(tx.hi * CV + v);
    V::unpack(rlo[ilo], tl);
    V::unpack(rlo[ihi], tr);
    V::unpack(rhi[ilo], bl);
    V::unpack(rhi[ihi], br);
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      if constexpr (BF) {
        // bf16 handles: the merged value is rounded to bf16 below; the lerps as FMAs (2 instead of 3 instructions each -- this kernel is bound by its
        // vector-ALU issue slots on bf16 handles).  fp32 handles keep the reference's separately rounded multiply and add.
        const float top = fmaf(tr[i] - tl[i], tx.t, tl[i]), bot = fmaf(br[i] - bl[i], tx.t, bl[i]);
        o[i] = fmaf(bot - top, tyt, top);
      } else {
        o[i] = lerp2(tl[i], tr[i], bl[i], br[i], tx.t, tyt);
      }
    }
  };
  for (int t = tid; t < NX * CQ; t += NTR) {
    const int x = t / CQ, v = t % CQ;
    float z[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) z[i] = 0.f;
    if (x < W) {
      float a[VN], u2[VN], u3[VN];
      V::unpack(p1[(size_t)x * CV + v], a);
      if (H2 == H && W2 == W) V::unpack(p2[((size_t)y * W + x) * CV + v], u2);
      else bil(r2lo, r2hi, ty2.t, tf1_tap(x, W2, sx2), v, u2);
      if (H3 == H && W3 == W) V::unpack(p3[((size_t)y * W + x) * CV + v], u3);
      else bil(r3lo, r3hi, ty3.t, tf1_tap(x, W3, sx3), v, u3);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        z[i] = div3((a[i] + u2[i]) + u3[i]);
        if constexpr (BF) z[i] = static_cast<float>(static_cast<__bf16>(z[i]));      // the merged map of a bf16 handle is a bf16 tensor
      }
    }
    float4* dst = reinterpret_cast<float4*>(&buf[x * CH + (VN / 2) * v]);
#pragma unroll
    for (int i = 0; i < VN / 4; ++i) dst[i] = make_float4(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3]);
  }
  __syncthreads();
  fft<NX, -1, CH, NTR>(buf, tw, tid);
  float tm;
  if constexpr (T16) {
    __shared__ float red2[NTR / 64];
    tm = rows_fwd_store16<NX, NTR>(buf, T, t16, red2, tid, cblk, b, y, B, H, C);
  } else {
    tm = rows_fwd_store<NX, NTR>(buf, T, tid, cblk, b, y, B, H, C);
  }
  if (tmax) block_max_to<NTR>(tm, tmax + b, red, tid);      // one atomic per work group at most (skipped when the word already holds more)
}


template <int NX> static void launch_rows_fwd(const ConvArgs& a, int layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16) {
  const int ntiles = a.B * a.H * (a.Cin / CB);
  const dim3 blk(rows_threads<NX>());
#define RF_LAUNCH(L, H16)                                                                                                              \
  do {                                                                                                                                 \
    const dim3 grid(persistent_grid(reinterpret_cast<const void*>(rows_fwd_kernel<NX, L, H16>), ntiles, rows_threads<NX>()));          \
    hipLaunchKernelGGL((rows_fwd_kernel<NX, L, H16>), grid, blk, 0, st, a.x, T, tw, a.B, a.H, a.W, a.Cin, ntiles, tmax, t16);          \
  } while (0)
  if (layout == 0) RF_LAUNCH(0, false);      // (fp32 handles keep T in fp32)
  else if (layout == 1) { if (t16) RF_LAUNCH(1, true); else RF_LAUNCH(1, false); }
  else { if (t16) RF_LAUNCH(2, true); else RF_LAUNCH(2, false); }
#undef RF_LAUNCH
}
template <int NX> static void launch_rows_fwd_merge(const ConvArgs& a, const FftMerge& m, int in_layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16) {
  const dim3 grid(a.B * a.H * (a.Cin / CB)), blk(rows_threads<NX>());
  const float sy2 = (float)m.H2 / (float)a.H, sx2 = (float)m.W2 / (float)a.W, sy3 = (float)m.H3 / (float)a.H, sx3 = (float)m.W3 / (float)a.W;
  if (in_layout == 1 && t16)
    hipLaunchKernelGGL((rows_fwd_merge_kernel<NX, true, true>), grid, blk, 0, st, a.x, m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, tw, a.B, a.H, a.W, a.Cin, sy2, sx2, sy3, sx3, tmax, t16);
  else if (in_layout == 1)
    hipLaunchKernelGGL((rows_fwd_merge_kernel<NX, true>), grid, blk, 0, st, a.x, m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, tw, a.B, a.H, a.W, a.Cin, sy2, sx2, sy3, sx3, tmax, nullptr);
  else
    hipLaunchKernelGGL((rows_fwd_merge_kernel<NX, false>), grid, blk, 0, st, a.x, m.x2, m.H2, m.W2, m.x3, m.H3, m.W3, T, tw, a.B, a.H, a.W, a.Cin, sy2, sx2, sy3, sx3, tmax, nullptr);
}
void cfft_rows_fwd(int NX, const ConvArgs& a, int layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16) {
#define CALL(N) launch_rows_fwd<N>(a, layout, T, tw, tmax, st, t16)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}
void cfft_rows_fwd_merge(int NX, const ConvArgs& a, const FftMerge& m, int in_layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16) {
#define CALL(N) launch_rows_fwd_merge<N>(a, m, in_layout, T, tw, tmax, st, t16)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}

}  // namespace cfft
}  // namespace jcm
