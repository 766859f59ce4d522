// Inverse row pass (+ epilogue, + the fused hand-over to the next layer) of the frequency-domain convolution (see conv_fft.hip).
#include "conv_fft_common.h"

namespace jcm {
namespace cfft {

// ---- rows, inverse + epilogue: T[b][y][kx][c] (C channels, padded to a multiple of 64) -> out with Cout channels
// LAYOUT: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar (Cout % 8 == 0)
// Persistent work groups with register prefetch, as rows_fwd_kernel: the next tile's half spectrum is in flight during the FFT and the stores.
// T16: T' arrives as complex fp16 in block floating point; sc.t16_inv[(b C / sc.t16_cb + c / sc.t16_cb) NXH + kx] = 1 / scale of the inverse column
// pass's (image, kx, sc.t16_cb channels) tile (conv_fft_common.h).
template <int NX, int LAYOUT, bool T16 = false>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_inv_kernel(const cf* __restrict__ T, void* __restrict__ out, const cf* __restrict__ twg, const float* __restrict__ bias,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int H, int W, int C,
                                                      int Cout, int pad, float norm0, int ntiles, Fp16Scale sc) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1, NTR = rows_threads<NX>(), K = (NXH * CH + NTR - 1) / NTR;
  const float ncommon = sc.tmax ? norm0 * sc.winv[0] * (sc.common ? fp16_unscale(tmax_of(sc.tmax, 0, sc.nb, 1), sc.hf) : 1.f) : norm0;      // powers of two: exact
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float par[3 * kParMax];      // bias | scale | shift of every channel: the epilogue issues no global load, so nothing drains the prefetch
  const int tid = threadIdx.x, ncb = C / CB;
  twiddles<NX, NTR>(tw, twg, tid);
  const bool par_lds = C <= kParMax;
  if (par_lds)
    for (int i = tid; i < C; i += NTR) {
      par[i] = i < Cout ? bias[i] : 0.f;
      par[kParMax + i] = relu_bn && i < Cout ? scale[i] : 1.f;
      par[2 * kParMax + i] = relu_bn && i < Cout ? shift[i] : 0.f;
    }
  float4 pre[T16 ? 1 : K];
  uint2 pre16[T16 ? K : 1];      // T16: two complex fp16 per item ...
  float pres[T16 ? K : 1];       // ... and 1 / scale of the tile they come from
  float pre_t = 0.f;      // max|T| word of the prefetched tile's image
  const bool per_image = sc.tmax && !sc.common;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int cblk = tile % ncb, by = tile / ncb;      // by = b * H + y
    if (per_image) pre_t = sc.tmax[by / H];
    if constexpr (T16) {
      const uint2* src = reinterpret_cast<const uint2*>(T) + (((size_t)by * NXH) * C + cblk * CB) / 2;
      const int nblk = C / sc.t16_cb;
      const float* ssrc = sc.t16_inv + (size_t)(by / H) * nblk * NXH;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const int t = tid + i * NTR, k = t / CH, v = t % CH;
        const bool in = t < NXH * CH;
        pre16[i] = in ? src[(size_t)k * (C / 2) + v] : make_uint2(0u, 0u);
        pres[i] = in ? ssrc[((cblk * CB + 2 * v) / sc.t16_cb) * NXH + k] : 0.f;
      }
    } else {
      const float4* src = reinterpret_cast<const float4*>(T + ((size_t)by * NXH) * C + cblk * CB);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const int t = tid + i * NTR, k = t / CH, v = t % CH;
        pre[i] = t < NXH * CH ? src[(size_t)k * (C / 2) + v] : make_float4(0.f, 0.f, 0.f, 0.f);      // (Ya.re, Ya.im, Yb.re, Yb.im)
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  while (tile < ntiles) {
    const float norm = per_image ? ncommon * fp16_unscale(pre_t, sc.hf) : ncommon;      // this image's scale (taken before the next tile's prefetch overwrites it)
    // Z = Y_c + i Y_{c+1} with the Hermitian extension Y[NX - k] = conj Y[k]; DC and Nyquist are real by symmetry
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int t = tid + i * NTR, k = t / CH, v = t % CH;
      if (t < NXH * CH) {
        float4 q;
        if constexpr (T16) {
          const cf ya = unpack_h2(pre16[i].x, pres[i]), yb = unpack_h2(pre16[i].y, pres[i]);
          q = make_float4(ya.x, ya.y, yb.x, yb.y);
        } else {
          q = pre[i];
        }
        const bool edge = k == 0 || k == NX / 2;
        if (edge) { q.y = 0.f; q.w = 0.f; }
        buf[k * CH + v] = cf{q.x - q.w, q.y + q.z};
        if (!edge) buf[(NX - k) * CH + v] = cf{q.x + q.w, q.z - q.y};
      }
    }
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    __syncthreads();
    fft<NX, 1, CH, NTR>(buf, tw, tid);
    const int cblk = tile % ncb, by = tile / ncb;
    const int y = by % H, b = by / H;
    if constexpr (LAYOUT == 2) {
      // planar bf16 [B][C/8][H*W][8]: an item = (pixel, 8-channel plane) = one 16-byte store; 8 consecutive lanes = 8 consecutive pixels of
      // one plane = a whole 128-byte line (the channel-pair mapping below would write 16-byte pieces of 16 different planes per instruction)
      if (par_lds) {
        typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
        const int nitems = ((W + 7) / 8) * 64;
        for (int i = tid; i < nitems; i += NTR) {
          const int xl = i & 7, g = (i >> 3) & 7, x = (i >> 6) * 8 + xl, c0 = cblk * CB + 8 * g;
          if (x < W && c0 < Cout) {
            const cf* zp = buf + pos<NX>(x + pad) * CH + 4 * g;
            bf16x8v o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const cf z = zp[j];
              const int c = c0 + 2 * j;
              float v0 = z.x * norm + par[c], v1 = z.y * norm + par[c + 1];
              if (relu_bn) { v0 = fmaxf(v0, 0.f) * par[kParMax + c] + par[2 * kParMax + c]; v1 = fmaxf(v1, 0.f) * par[kParMax + c + 1] + par[2 * kParMax + c + 1]; }
              o[2 * j] = static_cast<__bf16>(v0);
              o[2 * j + 1] = static_cast<__bf16>(v1);
            }
            *reinterpret_cast<bf16x8v*>(static_cast<__bf16*>(out) + (((size_t)b * (Cout >> 3) + (c0 >> 3)) * H * W + (size_t)y * W + x) * 8) = o;
          }
        }
        __syncthreads();      // every wave is done reading buf
        tile = next;
        continue;
      }
    }
    // a thread keeps its channel pair for the whole row (the thread count is a multiple of CH): bias / scale / shift are loaded once per tile
    const int v = tid % CH, c = cblk * CB + 2 * v;
    if (c < Cout) {
      const bool two = c + 1 < Cout, pairs = (Cout & 1) == 0;      // two channels = one aligned store
      float b0v, b1v, s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
      if (par_lds) {
        b0v = par[c]; b1v = par[c + 1]; s0 = par[kParMax + c]; s1 = par[kParMax + c + 1]; h0 = par[2 * kParMax + c]; h1 = par[2 * kParMax + c + 1];
      } else {
        b0v = bias[c]; b1v = two ? bias[c + 1] : 0.f;
        if (relu_bn) { s0 = scale[c]; h0 = shift[c]; if (two) { s1 = scale[c + 1]; h1 = shift[c + 1]; } }
      }
      for (int x = tid / CH; x < W; x += NTR / CH) {
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
    __syncthreads();      // every wave is done reading buf
    tile = next;
  }
}

// ---- rows, inverse of layer L + epilogue + rows, forward of layer L+1 in one kernel (fp32 handles; same map, same NX, C % 64 == 0):
// the activation between two frequency-domain layers never goes to HBM.  T_in[b][y][kx][c] -> T_out[kx][c/16][b][y][16].
// The epilogue's result IS the next layer's packed input: channel pair (c, c+1) = one complex number.  Persistent work groups with
// register prefetch as rows_inv_kernel; the epilogue takes the row out of LDS into registers and writes the activated row back into the
// SAME buffer (one row buffer instead of two: twice the work groups per CU).
template <int NX>
__global__ __launch_bounds__(rows_threads<NX>()) void rows_inv_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Tn, const cf* __restrict__ twg, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int B, int H, int W, int C,
                                                          int pad, float norm0, int ntiles, Fp16Scale sc) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1, NTR = rows_threads<NX>(), K = (NXH * CH + NTR - 1) / NTR, XP = NTR / CH, KX = (NX + XP - 1) / XP;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float par[3 * kParMax];
  __shared__ float red[NTR / 64];
  const float ncommon = sc.tmax ? norm0 * sc.winv[0] * (sc.common ? fp16_unscale(tmax_of(sc.tmax, 0, sc.nb, 1), sc.hf) : 1.f) : norm0;
  const int tid = threadIdx.x, ncb = C / CB;
  twiddles<NX, NTR>(tw, twg, tid);
  const bool par_lds = C <= kParMax;
  if (par_lds)
    for (int i = tid; i < C; i += NTR) {
      par[i] = bias[i];
      par[kParMax + i] = relu_bn ? scale[i] : 1.f;
      par[2 * kParMax + i] = relu_bn ? shift[i] : 0.f;
    }
  float4 pre[K];
  float pre_t = 0.f;
  const bool per_image = sc.tmax && !sc.common;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int cblk = tile % ncb, by = tile / ncb;
    if (per_image) pre_t = sc.tmax[by / H];
    const float4* src = reinterpret_cast<const float4*>(T + ((size_t)by * NXH) * C + cblk * CB);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int t = tid + i * NTR, k = t / CH, v = t % CH;
      pre[i] = t < NXH * CH ? src[(size_t)k * (C / 2) + v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  while (tile < ntiles) {
    const float norm = per_image ? ncommon * fp16_unscale(pre_t, sc.hf) : ncommon;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int t = tid + i * NTR, k = t / CH, v = t % CH;
      if (t < NXH * CH) {
        float4 q = pre[i];
        const bool edge = k == 0 || k == NX / 2;
        if (edge) { q.y = 0.f; q.w = 0.f; }
        buf[k * CH + v] = cf{q.x - q.w, q.y + q.z};
        if (!edge) buf[(NX - k) * CH + v] = cf{q.x + q.w, q.z - q.y};
      }
    }
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    __syncthreads();
    fft<NX, 1, CH, NTR>(buf, tw, tid);
    const int cblk = tile % ncb, by = tile / ncb;
    {
      const int v = tid % CH, c = cblk * CB + 2 * v;
      float b0v, b1v, s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
      if (par_lds) {
        b0v = par[c]; b1v = par[c + 1]; s0 = par[kParMax + c]; s1 = par[kParMax + c + 1]; h0 = par[2 * kParMax + c]; h1 = par[2 * kParMax + c + 1];
      } else {
        b0v = bias[c]; b1v = bias[c + 1];
        if (relu_bn) { s0 = scale[c]; h0 = shift[c]; s1 = scale[c + 1]; h1 = shift[c + 1]; }
      }
      cf o[KX];
#pragma unroll
      for (int i = 0; i < KX; ++i) {
        const int x = tid / CH + i * XP;
        o[i] = cf{0.f, 0.f};
        if (x < W) {
          const cf z = buf[pos<NX>(x + pad) * CH + v];
          float v0 = z.x * norm + b0v, v1 = z.y * norm + b1v;
          if (relu_bn) { v0 = fmaxf(v0, 0.f) * s0 + h0; v1 = fmaxf(v1, 0.f) * s1 + h1; }
          o[i] = cf{v0, v1};
        }
      }
      __syncthreads();      // every wave has taken its part of the inverse row out of buf
#pragma unroll
      for (int i = 0; i < KX; ++i) {
        const int x = tid / CH + i * XP;
        if (x < NX) buf[x * CH + v] = o[i];
      }
    }
    __syncthreads();
    fft<NX, -1, CH, NTR>(buf, tw, tid);
    const float tm = rows_fwd_store<NX, NTR>(buf, Tn, tid, cblk, by / H, by % H, B, H, C);
    if (sc.tmax_next) wave_max_stash(tm, red);
    __syncthreads();      // every wave is done reading buf
    if (sc.tmax_next) stash_to_word<NTR>(red, sc.tmax_next + by / H);
    tile = next;
  }
}

template <int NX> static void launch_rows_inv(const ConvArgs& a, int layout, const cf* T, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  const int ntiles = a.B * a.H * (a.CoutP / CB);
  const dim3 blk(rows_threads<NX>());
#define RI_LAUNCH(L, H16)                                                                                                                                           \
  do {                                                                                                                                                              \
    const dim3 grid(persistent_grid(reinterpret_cast<const void*>(rows_inv_kernel<NX, L, H16>), ntiles, rows_threads<NX>()));                                       \
    hipLaunchKernelGGL((rows_inv_kernel<NX, L, H16>), grid, blk, 0, st, T, a.out, tw, a.bias, a.scale, a.shift, a.relu_bn, a.H, a.W, a.CoutP, a.Cout, pad, norm, ntiles, sc); \
  } while (0)
  const bool h16 = sc.t16_inv != nullptr;
  if (layout == 0) RI_LAUNCH(0, false);      // (fp32 handles keep T' in fp32)
  else if (layout == 1) { if (h16) RI_LAUNCH(1, true); else RI_LAUNCH(1, false); }
  else { if (h16) RI_LAUNCH(2, true); else RI_LAUNCH(2, false); }
#undef RI_LAUNCH
}
template <int NX> static void launch_rows_inv_fwd(const ConvArgs& a, const cf* T, cf* Tn, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  const int ntiles = a.B * a.H * (a.Cout / CB);
  const dim3 grid(persistent_grid(reinterpret_cast<const void*>(rows_inv_fwd_kernel<NX>), ntiles, rows_threads<NX>()));
  hipLaunchKernelGGL(rows_inv_fwd_kernel<NX>, grid, dim3(rows_threads<NX>()), 0, st, T, Tn, tw, a.bias, a.scale, a.shift, a.relu_bn, a.B, a.H, a.W, a.Cout, pad, norm, ntiles, sc);
}
void cfft_rows_inv(int NX, const ConvArgs& a, int layout, const cf* T, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
#define CALL(N) launch_rows_inv<N>(a, layout, T, tw, pad, norm, sc, st)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}
void cfft_rows_inv_fwd(int NX, const ConvArgs& a, const cf* T, cf* Tn, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
#define CALL(N) launch_rows_inv_fwd<N>(a, T, Tn, tw, pad, norm, sc, st)
  CFFT_BY_SIZE(NX, CALL)
#undef CALL
}

}  // namespace cfft
}  // namespace jcm
