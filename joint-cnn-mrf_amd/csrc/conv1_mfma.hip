// conv1 on the bf16 matrix cores: 5x5 stride-2 SAME convolution 3 -> 64 of the (sub-sampled) RGB image
// fused with bias + ReLU + inference BatchNorm AND the 2x2/2 max-pool that follows it
// (main.py:44-45, 52-53, 61-62), on v_mfma_f32_32x32x16_bf16.
//
// GEMM view: M = conv-output pixels, N = 64 channels, K = 5 kernel rows x 16 (the 5 px x 3 ch
// = 15 contiguous input values of one kernel row, zero-padded to 16).  A workgroup owns a
// 16x16 patch of conv outputs (-> 8x8 pooled pixels): the 35x35x3 input window goes to LDS as
// bf16 once, every MFMA A fragment is 8 consecutive bf16 of a window row (4-byte aligned, read
// as 4 ds_read_b32; 12-byte lane stride = conflict-free), B fragments come from a 10 KB packed
// filter image.  The 32x32 accumulator layout keeps all four members of every 2x2 pooling
// window in one lane, so the pool is register-local and only the pooled map is written:
// 2.8 MB/image instead of 22 MB + 22 MB + 5.5 MB for the unfused conv1 -> pool pair.
#include <type_traits>

#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CM_T = 16;                        // conv-output patch edge
constexpr int CM_IN = 2 * (CM_T - 1) + 5;       // 35 input rows / cols
constexpr int CM_ROW = 112;                     // bf16 row pitch of the LDS window (35*3 = 105 -> 112)
constexpr int CM_WQ_F4 = 5 * 2 * 64;            // packed filter: [ky][h][co] x 16 B
namespace cfft { int persistent_grid(const void* kernel, int ntiles, int threads, int dyn_lds = 0); }      // conv_fft.hip: work groups the chip holds at once

// Persistent work groups (round 3): a group stages the packed filter once and walks tiles t, t + grid, ...; the window of the NEXT tile is
// loaded into registers (5 pixels per thread) before the MFMAs of the current one, so the global-load latency -- which bounded the
// one-tile-per-group version at 2 groups per CU -- hides behind them.
// NP = 1: bf16 operands (bf16 handles), bf16 output.  NP = 2 (fp32 handles, default route): every fp32 value as TWO FP16 parts of the value times a power
// of two, the three products x0w1 + x1w0 + x0w0 -- 22 significant bits, cgemm_split.hip's arithmetic, which the stride-1 layers of such a handle run on
// anyway; fp32 output.  The window's scale is ITS OWN: the work group takes the largest |value| of the 35x35x3 window it has just loaded (one extra
// barrier) and lifts it to [2^13, 2^14), the filter's scale comes with the packed filter -- so the kernel inherits fp32's range whatever the image holds,
// without a pass over the image.  60 MFMAs 32x32x16 per wave (rounds 3-5: three bf16 parts, six products, 120).
template <int NP, typename OutT>
__device__ __forceinline__ void conv1_mfma_pool_body(const float* __restrict__ x, const f32x4* __restrict__ wq, const float* __restrict__ bias,
                                                     const float* __restrict__ scale, const float* __restrict__ shift, OutT* __restrict__ out, int H0, int W0,
                                                     int sub, int Hin, int Win, int Hp, int Wp, int pad_t, int pad_l, int tiles_x, int tiles_img, int ntiles) {
  using ElT = std::conditional_t<NP == 1, __bf16, _Float16>;
  __shared__ __attribute__((aligned(16))) ElT win[NP][CM_IN * CM_ROW];
  __shared__ __attribute__((aligned(16))) f32x4 wl[NP][CM_WQ_F4];
  __shared__ float red[4];
  constexpr int PPT = (CM_IN * CM_IN + 255) / 256;      // window pixels per thread
  const int tid = threadIdx.x;
  // the 7-element tail of every 112-element row is zeroed once (it only ever meets zero weights; the pixel stores never touch it)
  for (int i = tid; i < NP * CM_IN * (CM_ROW - CM_IN * 3); i += 256) {
    const int p = i / (CM_IN * (CM_ROW - CM_IN * 3)), r = i - p * (CM_IN * (CM_ROW - CM_IN * 3));
    const int iy = r / (CM_ROW - CM_IN * 3), e = r - iy * (CM_ROW - CM_IN * 3);
    win[p][iy * CM_ROW + CM_IN * 3 + e] = (ElT)0.f;
  }
  for (int i = tid; i < NP * CM_WQ_F4; i += 256) (&wl[0][0])[i] = wq[i];
  const float winv = NP == 2 ? reinterpret_cast<const float*>(wq + NP * CM_WQ_F4)[0] : 1.f;      // 1 / (the filter's power of two), behind the packed filter

  const int lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int px = l31 & 15, pr = l31 >> 4;
  float bi[2], sc[2], sh[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) { bi[g] = bias[g * 32 + l31]; sc[g] = scale[g * 32 + l31]; sh[g] = shift[g * 32 + l31]; }

  float v[PPT][3];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int b = t / tiles_img, r = t - b * tiles_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* xb = x + (size_t)b * H0 * W0 * 3;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int i = tid + 256 * k;
      const int iy = i / CM_IN, ix = i - iy * CM_IN;
      const int gy = ty * CM_T * 2 - pad_t + iy, gx = tx * CM_T * 2 - pad_l + ix;
      v[k][0] = v[k][1] = v[k][2] = 0.f;
      if (i < CM_IN * CM_IN && (unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win) {
        const float* px3 = xb + ((size_t)(gy * sub) * W0 + gx * sub) * 3;
        v[k][0] = px3[0]; v[k][1] = px3[1]; v[k][2] = px3[2];
      }
    }
  };
  int t = blockIdx.x;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    // window -> LDS as 16-bit parts (round to nearest even; the remainders are exact)
    float xs = 1.f, xinv = 1.f;
    if constexpr (NP == 2) {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < PPT; ++k) m = fmaxf(m, fmaxf(fabsf(v[k][0]), fmaxf(fabsf(v[k][1]), fabsf(v[k][2]))));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if ((tid & 63) == 0) red[tid >> 6] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (m > 1.0e-30f && m < 3.0e38f) {      // (a zero / vanishing or non-finite window: no scaling)
        int ex = 0;
        (void)frexpf(m, &ex);                   // m in [2^(ex-1), 2^ex)
        xs = ldexpf(1.f, 14 - ex);
        xinv = ldexpf(1.f, ex - 14);
      }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int i = tid + 256 * k;
      if (i < CM_IN * CM_IN) {
        const int iy = i / CM_IN, ix = i - iy * CM_IN;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            if (NP == 2 && p == 0) v[k][e] *= xs;      // (exact: a power of two)
            const ElT q = (ElT)v[k][e];
            win[p][iy * CM_ROW + ix * 3 + e] = q;
            v[k][e] -= (float)q;
          }
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);      // in flight behind the MFMAs below

    f32x16 acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      bf16x8 af[NP][2], bf[NP][2];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned* win32 = reinterpret_cast<const unsigned*>(win[p]);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int py = 2 * (2 * wid + f) + pr;                                  // conv row inside the patch
          const int e0 = ((2 * py + ky) * CM_ROW + 6 * px + 8 * h) >> 1;          // dword index (4-byte aligned)
          u32x4 u;
          u[0] = win32[e0]; u[1] = win32[e0 + 1]; u[2] = win32[e0 + 2]; u[3] = win32[e0 + 3];
          af[p][f] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) bf[p][g] = __builtin_bit_cast(bf16x8, wl[p][(ky * 2 + h) * 64 + g * 32 + l31]);
      }
      // NP = 2, small terms first: (x part, w part) = (0,1) (1,0) (0,0)
      constexpr int NPROD = NP == 2 ? 3 : 1;
#pragma unroll
      for (int s = 0; s < NPROD; ++s) {
        const int pxp = NP == 1 ? 0 : (s == 1 ? 1 : 0);
        const int pwp = NP == 1 ? 0 : (s == 0 ? 1 : 0);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            if constexpr (NP == 1) acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pxp][f], bf[pwp][g], acc[f][g], 0, 0, 0);
            else acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[pxp][f]), __builtin_bit_cast(f16x8, bf[pwp][g]), acc[f][g], 0, 0, 0);
          }
      }
    }
    // epilogue.  Accumulator reg i of lane (h, co) is fragment pixel r = (i&3) + 8*(i>>2) + 4*h, i.e.
    // patch row (r>>4), column (r&15): regs {i, i+1} are horizontal neighbours (i even) and
    // {i, i+8} vertical ones, so one lane holds whole 2x2 windows: pooled column q = (r&15)>>1.
    const int b = t / tiles_img, rt = t - b * tiles_img;
    const int ty = rt / tiles_x, tx = rt - ty * tiles_x;
    const int oy0 = ty * CM_T, ox0 = tx * CM_T;
    const float un = xinv * winv;      // NP = 2: the two powers of two come off the sums (exact); 1 otherwise
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = g * 32 + l31;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int PY = (oy0 >> 1) + 2 * wid + f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // j -> regs (i0, i0+1, i0+8, i0+9), i0 = (j&1)*2 + (j>>1)*4
          const int i0 = (j & 1) * 2 + (j >> 1) * 4;
          const float v0 = fmaxf((NP == 2 ? acc[f][g][i0] * un : acc[f][g][i0]) + bi[g], 0.f) * sc[g] + sh[g];
          const float v1 = fmaxf((NP == 2 ? acc[f][g][i0 + 1] * un : acc[f][g][i0 + 1]) + bi[g], 0.f) * sc[g] + sh[g];
          const float v2 = fmaxf((NP == 2 ? acc[f][g][i0 + 8] * un : acc[f][g][i0 + 8]) + bi[g], 0.f) * sc[g] + sh[g];
          const float v3 = fmaxf((NP == 2 ? acc[f][g][i0 + 9] * un : acc[f][g][i0 + 9]) + bi[g], 0.f) * sc[g] + sh[g];
          const int r = (i0 & 3) + 8 * (i0 >> 2) + 4 * h;     // fragment pixel of reg i0 (row 0 of the pair)
          const int PX = (ox0 >> 1) + ((r & 15) >> 1);
          if (PY < Hp && PX < Wp) out[(((size_t)b * Hp + PY) * Wp + PX) * 64 + co] = (OutT)fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
        }
      }
    }
    __syncthreads();      // the window is overwritten by the next tile
  }
}

__global__ __launch_bounds__(256) void conv1_mfma_pool_kernel(const float* __restrict__ x, const f32x4* __restrict__ wq,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, __bf16* __restrict__ out,
                                                              int H0, int W0, int sub, int Hin, int Win, int Hp, int Wp,
                                                              int pad_t, int pad_l, int tiles_x, int tiles_img, int ntiles) {
  conv1_mfma_pool_body<1, __bf16>(x, wq, bias, scale, shift, out, H0, W0, sub, Hin, Win, Hp, Wp, pad_t, pad_l, tiles_x, tiles_img, ntiles);
}
// fp32 handles on the default route: split operands (NP = 2: two scaled fp16 parts), fp32 in, fp32 out
__global__ __launch_bounds__(256) void conv1_mfma_pool_split_kernel(const float* __restrict__ x, const f32x4* __restrict__ wq,
                                                                    const float* __restrict__ bias, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, float* __restrict__ out,
                                                                    int H0, int W0, int sub, int Hin, int Win, int Hp, int Wp,
                                                                    int pad_t, int pad_l, int tiles_x, int tiles_img, int ntiles) {
  conv1_mfma_pool_body<2, float>(x, wq, bias, scale, shift, out, H0, W0, sub, Hin, Win, Hp, Wp, pad_t, pad_l, tiles_x, tiles_img, ntiles);
}

// ---- the same fusion on the exact fp32 path: v_mfma_f32_32x32x2_f32 (an exact k-ordered fma chain), fp32 window in LDS.
// K = 5 kernel rows x 16 (15 values + a zero pad) = 40 MFMA k-steps of 2; lane (pixel, h) supplies window element 2s+h of its
// row, lane (channel, h) the matching filter value (one ds_read_b32 each).  Same accumulator layout, same register-local pool.
constexpr int CF_ROW = 35 * 3 + 3;              // fp32 row pitch of the LDS window (105 values + 3 zeros: element 15 of the last pixel's row window)
__global__ __launch_bounds__(256) void conv1_mfma_pool_f32_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                                  const float* __restrict__ bias, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ out,
                                                                  int H0, int W0, int sub, int Hin, int Win, int Hp, int Wp,
                                                                  int pad_t, int pad_l, int tiles_x) {
  __shared__ float win[CM_IN * CF_ROW];
  __shared__ float wl[5 * 16 * 64];            // [ky][k'][co]
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * CM_T, ox0 = tx * CM_T;
  const int tid = threadIdx.x;
  const float* xb = x + (size_t)b * H0 * W0 * 3;
  for (int i = tid; i < CM_IN * CM_IN; i += 256) {
    const int iy = i / CM_IN, ix = i - iy * CM_IN;
    const int gy = oy0 * 2 - pad_t + iy, gx = ox0 * 2 - pad_l + ix;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win) {
      const float* px3 = xb + ((size_t)(gy * sub) * W0 + gx * sub) * 3;
      v0 = px3[0]; v1 = px3[1]; v2 = px3[2];
    }
    float* w3 = win + iy * CF_ROW + ix * 3;
    w3[0] = v0; w3[1] = v1; w3[2] = v2;
  }
  for (int i = tid; i < CM_IN * 3; i += 256) win[(i / 3) * CF_ROW + CM_IN * 3 + i % 3] = 0.f;
  for (int i = tid; i < 5 * 16 * 64; i += 256) wl[i] = wq[i];
  __syncthreads();

  const int lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int px = l31 & 15, pr = l31 >> 4;
  f32x16 acc[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    int arow[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) arow[f] = (2 * (2 * (2 * wid + f) + pr) + ky) * CF_ROW + 6 * px + h;    // + 2s: window element 2s+h
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      float af[2], bf[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) af[f] = win[arow[f] + 2 * s8];
#pragma unroll
      for (int g = 0; g < 2; ++g) bf[g] = wl[(ky * 16 + 2 * s8 + h) * 64 + g * 32 + l31];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[f], bf[g], acc[f][g], 0, 0, 0);
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int co = g * 32 + l31;
    const float bi = bias[co], sc = scale[co], sh = shift[co];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int PY = (oy0 >> 1) + 2 * wid + f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {                 // one 2x2 pooling window: regs (i0, i0+1, i0+8, i0+9)
        const int i0 = (j & 1) * 2 + (j >> 1) * 4;
        const float v0 = fmaxf(acc[f][g][i0] + bi, 0.f) * sc + sh;
        const float v1 = fmaxf(acc[f][g][i0 + 1] + bi, 0.f) * sc + sh;
        const float v2 = fmaxf(acc[f][g][i0 + 8] + bi, 0.f) * sc + sh;
        const float v3 = fmaxf(acc[f][g][i0 + 9] + bi, 0.f) * sc + sh;
        const int r = (i0 & 3) + 8 * (i0 >> 2) + 4 * h;
        const int PX = (ox0 >> 1) + ((r & 15) >> 1);
        if (PY < Hp && PX < Wp) out[(((size_t)b * Hp + PY) * Wp + PX) * 64 + co] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      }
    }
  }
}

constexpr int CS_NP = 2;
// HWIO [5,5,3,64] fp32 -> [part][ky][h][co][8] fp16 (two parts of every weight times the power of two that lifts max|w| to [2^13, 2^14)), k' = 8h+i = 3*kx + c,
// k' = 15 is the zero pad; one float behind them: the inverse of that power of two.  One work group (the filter has 4800 values).
__global__ __launch_bounds__(256) void pack_conv1_split_kernel(const float* __restrict__ w, _Float16* __restrict__ wq) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < 5 * 5 * 3 * 64; i += 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float ws = 1.f, winv = 1.f;
  if (m > 1.0e-30f && m < 3.0e38f) {
    int ex = 0;
    (void)frexpf(m, &ex);
    ws = ldexpf(1.f, 14 - ex);
    winv = ldexpf(1.f, ex - 14);
  }
  for (int i = tid; i < CM_WQ_F4 * 8; i += 256) {
    const int e = i & 7, co = (i >> 3) & 63, h = (i >> 9) & 1, ky = i >> 10;
    const int k = 8 * h + e;
    float v = 0.f;
    if (k < 15) v = w[(((size_t)ky * 5 + k / 3) * 3 + k % 3) * 64 + co] * ws;
    for (int p = 0; p < CS_NP; ++p) {
      const _Float16 q = (_Float16)v;
      wq[(size_t)p * CM_WQ_F4 * 8 + i] = q;
      v -= (float)q;
    }
  }
  if (tid == 0) reinterpret_cast<float*>(wq + (size_t)CS_NP * CM_WQ_F4 * 8)[0] = winv;
}
size_t conv1_split_weight_bytes() { return (size_t)CS_NP * CM_WQ_F4 * 16 + 16; }
hipError_t pack_conv1_split(const float* w_hwio, void* wq, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv1_split_kernel, dim3(1), dim3(256), 0, st, w_hwio, static_cast<_Float16*>(wq));
  return hipGetLastError();
}
hipError_t conv1_mfma_pool_split(const float* x, const void* wq, const float* bias, const float* scale, const float* shift, float* out,
                                 int B, int H0, int W0, int sub, hipStream_t st) {
  if (H0 % (4 * sub) != 0 || W0 % (4 * sub) != 0) return hipErrorInvalidValue;
  const int Hin = H0 / sub, Win = W0 / sub;
  const int Ho = Hin / 2, Wo = Win / 2;
  const int tot_h = (Ho - 1) * 2 + 5 - Hin, tot_w = (Wo - 1) * 2 + 5 - Win;
  const int pad_t = tot_h / 2, pad_l = tot_w / 2;
  const int tiles_x = (Wo + CM_T - 1) / CM_T, tiles_y = (Ho + CM_T - 1) / CM_T;
  const int ntiles = tiles_x * tiles_y * B;
  const int grid = cfft::persistent_grid(reinterpret_cast<const void*>(conv1_mfma_pool_split_kernel), ntiles, 256);
  hipLaunchKernelGGL(conv1_mfma_pool_split_kernel, dim3(grid), dim3(256), 0, st, x, static_cast<const f32x4*>(wq), bias, scale, shift, out,
                     H0, W0, sub, Hin, Win, Ho / 2, Wo / 2, pad_t, pad_l, tiles_x, tiles_x * tiles_y, ntiles);
  return hipGetLastError();
}

// HWIO [5,5,3,64] fp32 -> [ky][k' = 3*kx + c (15 = zero pad)][co] fp32
__global__ void pack_conv1_f32_kernel(const float* __restrict__ w, float* __restrict__ wq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 5 * 16 * 64) return;
  const int co = i & 63, k = (i >> 6) & 15, ky = i >> 10;
  wq[i] = k < 15 ? w[(((size_t)ky * 5 + k / 3) * 3 + k % 3) * 64 + co] : 0.f;
}
hipError_t pack_conv1_f32(const float* w_hwio, float* wq, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv1_f32_kernel, dim3((5 * 16 * 64 + 255) / 256), dim3(256), 0, st, w_hwio, wq);
  return hipGetLastError();
}
// x [B,H0,W0,3] fp32 -> out [B,Hp,Wp,64] fp32 (conv s2 + bias/ReLU/BN + pool s2); H0/sub and W0/sub divisible by 4
hipError_t conv1_mfma_pool_f32(const float* x, const float* wq, const float* bias, const float* scale, const float* shift, float* out,
                               int B, int H0, int W0, int sub, hipStream_t st) {
  if (H0 % (4 * sub) != 0 || W0 % (4 * sub) != 0) return hipErrorInvalidValue;
  const int Hin = H0 / sub, Win = W0 / sub;
  const int Ho = Hin / 2, Wo = Win / 2;
  const int tot_h = (Ho - 1) * 2 + 5 - Hin, tot_w = (Wo - 1) * 2 + 5 - Win;
  const int pad_t = tot_h / 2, pad_l = tot_w / 2;
  const int tiles_x = (Wo + CM_T - 1) / CM_T, tiles_y = (Ho + CM_T - 1) / CM_T;
  hipLaunchKernelGGL(conv1_mfma_pool_f32_kernel, dim3(tiles_x * tiles_y, B), dim3(256), 0, st, x, wq, bias, scale, shift, out, H0, W0, sub,
                     Hin, Win, Ho / 2, Wo / 2, pad_t, pad_l, tiles_x);
  return hipGetLastError();
}

// HWIO [5,5,3,64] fp32 -> [ky][h][co][8] bf16, k' = 8h+i = 3*kx + c, k' = 15 is the zero pad.
__global__ void pack_conv1_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CM_WQ_F4 * 8) return;
  const int e = i & 7, co = (i >> 3) & 63, h = (i >> 9) & 1, ky = i >> 10;
  const int k = 8 * h + e;
  float v = 0.f;
  if (k < 15) v = w[(((size_t)ky * 5 + k / 3) * 3 + k % 3) * 64 + co];
  wq[i] = (__bf16)v;
}

hipError_t pack_conv1_bf16(const float* w_hwio, void* wq, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv1_bf16_kernel, dim3((CM_WQ_F4 * 8 + 255) / 256), dim3(256), 0, st, w_hwio, static_cast<__bf16*>(wq));
  return hipGetLastError();
}

// x [B,H0,W0,3] fp32 -> out [B,Hp,Wp,64] bf16 with Hp = (H0/sub)/4, Wp = (W0/sub)/4 (conv s2 then pool s2).
// Requires H0/sub and W0/sub divisible by 4 (all three branches of the 480x720 model).
hipError_t conv1_mfma_pool(const float* x, const void* wq, const float* bias, const float* scale, const float* shift,
                           void* out, int B, int H0, int W0, int sub, hipStream_t st) {
  if (H0 % (4 * sub) != 0 || W0 % (4 * sub) != 0) return hipErrorInvalidValue;
  const int Hin = H0 / sub, Win = W0 / sub;
  const int Ho = Hin / 2, Wo = Win / 2;
  const int tot_h = (Ho - 1) * 2 + 5 - Hin, tot_w = (Wo - 1) * 2 + 5 - Win;     // SAME: 3 -> (1 before, 2 after)
  const int pad_t = tot_h / 2, pad_l = tot_w / 2;
  const int tiles_x = (Wo + CM_T - 1) / CM_T, tiles_y = (Ho + CM_T - 1) / CM_T;
  const int ntiles = tiles_x * tiles_y * B;
  const int grid = cfft::persistent_grid(reinterpret_cast<const void*>(conv1_mfma_pool_kernel), ntiles, 256);
  hipLaunchKernelGGL(conv1_mfma_pool_kernel, dim3(grid), dim3(256), 0, st, x, static_cast<const f32x4*>(wq), bias,
                     scale, shift, static_cast<__bf16*>(out), H0, W0, sub, Hin, Win, Ho / 2, Wo / 2, pad_t, pad_l, tiles_x, tiles_x * tiles_y, ntiles);
  return hipGetLastError();
}

}  // namespace jcm
