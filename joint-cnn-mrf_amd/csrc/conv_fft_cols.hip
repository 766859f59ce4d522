// Column passes of the frequency-domain convolution: forward with the bf16 operand split fused in, inverse (see conv_fft.hip).
#include "conv_fft_common.h"

namespace jcm {
namespace cfft {

// ---- columns, forward + operand split: T[kx][c/16][b][y][16] -> Xs[f = kx NY + ky][m-tile][c/16][re|im][part][k-half][row][8] bf16
// One work group = (IMG images, kx, one 16-channel chunk): its input is one contiguous run of T; after the FFT along y every spectrum is
// split into two bf16 parts (NP = 2: x = x0 + x1, each rounded to nearest, 16 significant bits) and stored as
// 16-byte MFMA operand units -- 8 consecutive channels of one image -- with the units of the work group's IMG images consecutive: 128-byte
// lines for IMG = 8.  The result is the channel GEMM's LDS image (cgemm_split.hip), which that kernel fetches by LDS-DMA.
// NP = 4: two FP16 parts of the spectrum times the power-of-two scale derived from max|T| (Fp16Scale, conv_fft_common.h); layout as NP = 2.
// NP = 5 (bf16 handles): ONE fp16 part of the scaled spectrum; the GEMM's stage is 32 channels and the two "part" planes of the NP = 2 layout hold
// its two 16-channel halves: this work group's chunk kc is half (kc & 1) of stage kc / 2.
// T16 (NP = 5): T arrives as complex fp16 in block floating point, t16[(b KC/4 + kc/4) H + y] = 1 / (scale of the row pass's tile) (conv_fft_common.h).
// Round 5: PERSISTENT work groups (grid = what the chip holds at once) that request the next tile's input -- 16 bytes per lane -- into registers
// before the butterflies of the current one, so the loads of tile n + 1 are in flight behind the transform and the stores of tile n (round 4: one
// tile per work group, 8-byte loads, nothing overlapped inside a work group: 3.4-4.2 TB/s).
template <int NY, int NP, bool T16 = false>
__global__ __launch_bounds__(colfwd_threads<NY>()) void cols_fwd_split_kernel(const cf* __restrict__ T, uint4* __restrict__ Xs, const cf* __restrict__ twg, int B, int H, int KC, int MT,
                                                            int mtiles, const float* __restrict__ tmax, int common, const float* __restrict__ t16, int ntiles) {
  constexpr int IMG = colimg<NY>(), CH = IMG * 16, NTC = colfwd_threads<NY>(), NPP = NP == 5 ? 1 : 2;      // NPP: 16-byte units this work group writes per plane
  constexpr int CPV = T16 ? 4 : 2;                       // complex numbers per 16-byte load
  constexpr int NV = NY * CH / CPV, KV = (NV + NTC - 1) / NTC, VPR = CH / CPV, VPI = 16 / CPV;      // vectors per tile / thread / row / (image, row)
  extern __shared__ __attribute__((aligned(16))) char smem_cf[];
  cf* buf = reinterpret_cast<cf*>(smem_cf);
  cf* tw = buf + NY * CH;
  float4* buf4 = reinterpret_cast<float4*>(buf);
  const int tid = threadIdx.x;
  const int NG = (B + IMG - 1) / IMG;
  twiddles<NY, NTC>(tw, twg, tid);
  __shared__ float xsc[IMG];      // NP >= 4: the power of two of each of the tile's images
  uint4 pre[KV];
  float psc[KV];                  // T16: 1 / scale of the row tile a vector comes from
  auto coords = [&](int tile, int& kc, int& kx, int& b0, int& nimg) __attribute__((always_inline)) {
    const int g = tile % NG, kk = tile / NG;
    kc = kk % KC; kx = kk / KC;
    b0 = g * IMG; nimg = min(IMG, B - b0);
  };
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    int kc, kx, b0, nimg;
    coords(tile, kc, kx, b0, nimg);
    const uint4* src = reinterpret_cast<const uint4*>(T) + (((size_t)kx * KC + kc) * B + b0) * H * VPI;
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int t = tid + i * NTC, y = t / VPR, v = t % VPR, img = v / VPI, cv = v % VPI;
      const bool ok = t < NV && y < H && img < nimg;
      pre[i] = ok ? src[((size_t)img * H + y) * VPI + cv] : make_uint4(0u, 0u, 0u, 0u);
      if constexpr (T16) psc[i] = ok ? t16[((size_t)(b0 + img) * (KC >> 2) + (kc >> 2)) * H + y] : 0.f;
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    int kc, kx, b0, nimg;
    coords(tile, kc, kx, b0, nimg);
    // registers -> LDS: buf[y][image * 16 + channel] complex fp32 (the previous tile's stores have read the buffer: barrier at the loop's end)
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int t = tid + i * NTC;
      if (t < NV) {
        if constexpr (T16) {
          const cf a = unpack_h2(pre[i].x, psc[i]), b2 = unpack_h2(pre[i].y, psc[i]), c2 = unpack_h2(pre[i].z, psc[i]), d2 = unpack_h2(pre[i].w, psc[i]);
          buf4[2 * t] = make_float4(a.x, a.y, b2.x, b2.y);
          buf4[2 * t + 1] = make_float4(c2.x, c2.y, d2.x, d2.y);
        } else {
          buf4[t] = make_float4(__uint_as_float(pre[i].x), __uint_as_float(pre[i].y), __uint_as_float(pre[i].z), __uint_as_float(pre[i].w));
        }
      }
    }
    if constexpr (NP >= 4)
      if (tid < IMG) xsc[tid] = tid < nimg ? fp16_scale(tmax_of(tmax, b0 + tid, B, common), (float)H) : 1.f;
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + (int)gridDim.x);      // in flight behind the butterflies and the stores below
    fft<NY, -1, CH, NTC>(buf, tw, tid);
    // item = (ky, k-half, image): 8 complex numbers -> NP units of the real parts + NP units of the imaginary parts
    const int mt = b0 / MT, r0 = b0 - mt * MT;
    for (int it = tid; it < NY * 2 * IMG; it += NTC) {
      const int img = it % IMG, kg = (it / IMG) & 1, ky = it / (2 * IMG);
      const cf* z = buf + pos<NY>(ky) * CH + img * 16 + kg * 8;
      float re[8], im[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { re[e] = z[e].x; im[e] = z[e].y; }
      uint4 ur[NPP], ui[NPP];
      if constexpr (NP == 5) {
        const float xscale = xsc[img];
        ur[0] = round8h(re, xscale);
        ui[0] = round8h(im, xscale);
      } else if constexpr (NP == 4) {
        const float xscale = xsc[img];      // this image's power of two
        split8h(re, xscale, ur);
        split8h(im, xscale, ui);
      } else {
        split8<NPP>(re, ur);
        split8<NPP>(im, ui);
      }
      const size_t f = (size_t)kx * NY + ky;
      if constexpr (NP == 5) {
        // stage kc / 2 of KC / 2, unit ((c * 2 + half) * 2 + kg) * MT + row with half = kc & 1
        uint4* dst = Xs + (((f * mtiles + mt) * (KC >> 1) + (kc >> 1)) * 8 + (size_t)(kc & 1) * 2 + kg) * MT + r0 + img;
        st_stream(dst, ur[0]);
        st_stream(dst + (size_t)4 * MT, ui[0]);
      } else {
        uint4* dst = Xs + (((f * mtiles + mt) * KC + kc) * (4 * NPP) + kg) * MT + r0 + img;      // unit ((c * NPP + p) * 2 + kg) * MT + row
#pragma unroll
        for (int p = 0; p < NPP; ++p) {
          st_stream(dst + (size_t)(0 * NPP + p) * 2 * MT, ur[p]);
          st_stream(dst + (size_t)(1 * NPP + p) * 2 * MT, ui[p]);
        }
      }
    }
    __syncthreads();      // every thread is done with buf and xsc before the next tile is deposited
  }
}

// ---- columns, inverse: Yf[ky][kx][b][ldy channels] -> T[b][y][kx][c < C], y < H (row y of the output is row y + pad of the circular convolution)
// T16: T' is written as complex fp16 in block floating point, t16[(b C/CH + cblk) NXH + kx] = 1 / (this work group's scale) (conv_fft_common.h)
// ... and Yf arrives as complex fp16 = product * 2^-k (cgemm_split.hip, Y16); yinv = 2^k goes into the tile's scale word
template <int NY, bool T16 = false>
__global__ __launch_bounds__(colinv_threads<NY>()) void cols_inv_kernel(const cf* __restrict__ Yf, cf* __restrict__ T, const cf* __restrict__ twg, int B, int H, int NXH, int C, int ldy,
                                                      int pad, float* __restrict__ t16, float yinv) {
  constexpr int CH = colblk<NY>(), CB = CH, NTC = colinv_threads<NY>();
  __shared__ cf buf[NY * CH];
  __shared__ cf tw[NY];
  const int tid = threadIdx.x;
  const int cblk = blockIdx.x % (C / CB), bk = blockIdx.x / (C / CB);
  const int kx = bk % NXH, b = bk / NXH;
  twiddles<NY, NTC>(tw, twg, tid);
  for (int t = tid; t < NY * CH; t += NTC) {
    const int ky = t / CH, v = t % CH;
    const size_t src = ((size_t)(kx * NY + ky) * B + b) * ldy + cblk * CB + v;
    if constexpr (T16) buf[t] = unpack_h2(reinterpret_cast<const unsigned*>(Yf)[src], 1.f);
    else buf[t] = Yf[src];
  }
  __syncthreads();
  fft<NY, 1, CH, NTC>(buf, tw, tid);
  if constexpr (T16) {
    constexpr int KM = (NY * CH + NTC - 1) / NTC;
    __shared__ float red[NTC / 64];
    cf val[KM];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < KM; ++i) {
      const int t = tid + i * NTC, y = t / CH, v = t % CH;
      val[i] = cf{0.f, 0.f};
      if (t < H * CH) {
        val[i] = buf[pos<NY>(y + pad) * CH + v];
        m = fmaxf(m, fmaxf(fabsf(val[i].x), fabsf(val[i].y)));
      }
    }
    const float s = bfp_scale(block_max_all<NTC>(m, red, tid));
    unsigned* dst = reinterpret_cast<unsigned*>(T);
#pragma unroll
    for (int i = 0; i < KM; ++i) {
      const int t = tid + i * NTC, y = t / CH, v = t % CH;
      if (t < H * CH) st_stream(&dst[((size_t)(b * H + y) * NXH + kx) * C + cblk * CB + v], pack_h2(val[i].x * s, val[i].y * s));
    }
    if (tid == 0) t16[((size_t)b * (C / CB) + cblk) * NXH + kx] = (1.0f / s) * yinv;
  } else {
    for (int t = tid; t < H * CH; t += NTC) {
      const int y = t / CH, v = t % CH;
      st_stream(&T[((size_t)(b * H + y) * NXH + kx) * C + cblk * CB + v], buf[pos<NY>(y + pad) * CH + v]);
    }
  }
}


template <int NY> static hipError_t launch_cols_fwd(const ConvArgs& a, int np, const cf* T, void* Xs, const cf* tw, int NXH, int MT, const Fp16Scale& sc, hipStream_t st) {
  constexpr int IMG = colimg<NY>();
  constexpr int lds = (NY * IMG * 16 + NY) * (int)sizeof(cf);
  const int KC = a.Cin / 16, mtiles = (a.B + MT - 1) / MT;
  const int ntiles = NXH * KC * ((a.B + IMG - 1) / IMG);
  static LdsAttr attr2, attr4, attr5, attr5h;
  const dim3 blk(colfwd_threads<NY>());
#define COLS_FWD(KERNEL, ATTR, ...)                                                                              \
  do {                                                                                                           \
    if (hipError_t e = ATTR.ensure(reinterpret_cast<const void*>(KERNEL), lds); e != hipSuccess) return e;        \
    const dim3 grid((unsigned)persistent_grid(reinterpret_cast<const void*>(KERNEL), ntiles, (int)blk.x, lds));  \
    hipLaunchKernelGGL(KERNEL, grid, blk, lds, st, T, static_cast<uint4*>(Xs), tw, a.B, a.H, KC, MT, mtiles, __VA_ARGS__, ntiles); \
  } while (0)
  if (np == 5 && sc.t16_fwd) {
    if (!sc.tmax || (KC & 3)) return hipErrorInvalidValue;
    COLS_FWD((cols_fwd_split_kernel<NY, 5, true>), attr5h, sc.tmax, sc.common, sc.t16_fwd);
  } else if (np == 5) {
    if (!sc.tmax || (KC & 1)) return hipErrorInvalidValue;
    COLS_FWD((cols_fwd_split_kernel<NY, 5>), attr5, sc.tmax, sc.common, (const float*)nullptr);
  } else if (np == 2) {
    COLS_FWD((cols_fwd_split_kernel<NY, 2>), attr2, (const float*)nullptr, 0, (const float*)nullptr);
  } else if (np == 4) {
    if (!sc.tmax) return hipErrorInvalidValue;
    COLS_FWD((cols_fwd_split_kernel<NY, 4>), attr4, sc.tmax, sc.common, (const float*)nullptr);
  } else {
    return hipErrorInvalidValue;
  }
#undef COLS_FWD
  return hipSuccess;
}
// a.CoutP = output channels the inverse passes transform (Cout padded to 64); ldy = channel stride of Yf (Cout padded to the GEMM's N tile)
template <int NY> static void launch_cols_inv(const ConvArgs& a, const cf* Yf, cf* T, const cf* tw, int NXH, int ldy, int pad, hipStream_t st, float* t16, float y16_inv) {
  const dim3 grid(a.B * NXH * (a.CoutP / colblk<NY>())), blk(colinv_threads<NY>());
  if (t16) hipLaunchKernelGGL((cols_inv_kernel<NY, true>), grid, blk, 0, st, Yf, T, tw, a.B, a.H, NXH, a.CoutP, ldy, pad, t16, y16_inv);      // (y16_inv != 0: conv_fft.hip)
  else hipLaunchKernelGGL((cols_inv_kernel<NY, false>), grid, blk, 0, st, Yf, T, tw, a.B, a.H, NXH, a.CoutP, ldy, pad, nullptr, 0.f);
}
hipError_t cfft_cols_fwd(int NY, const ConvArgs& a, int np, const cf* T, void* Xs, const cf* tw, int NXH, int MT, const Fp16Scale& sc, hipStream_t st) {
  hipError_t e = hipSuccess;
#define CALL(N) e = launch_cols_fwd<N>(a, np, T, Xs, tw, NXH, MT, sc, st)
  CFFT_BY_SIZE(NY, CALL)
#undef CALL
  return e;
}
void cfft_cols_inv(int NY, const ConvArgs& a, const cf* Yf, cf* T, const cf* tw, int NXH, int ldy, int pad, hipStream_t st, float* t16, float y16_inv) {
#define CALL(N) launch_cols_inv<N>(a, Yf, T, tw, NXH, ldy, pad, st, t16, y16_inv)
  CFFT_BY_SIZE(NY, CALL)
#undef CALL
}

}  // namespace cfft
}  // namespace jcm
