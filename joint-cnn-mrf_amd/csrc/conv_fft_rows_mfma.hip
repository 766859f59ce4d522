// Row passes of the frequency-domain convolution of a bf16 handle (16-bit row-transformed tensors, np = 5) with the 96-point transform as a
// MATRIX PRODUCT on the matrix cores.
//
// The register kernels (conv_fft_rows_reg.hip) are bound by their vector-ALU issue slots on these handles (round 6, SQ_INSTS_VALU x 4 cycles = 85 % of the
// kernel time at 2.3-3.6 TB/s; the library runs without packed fp32 arithmetic, fft_lds.h): a 96-point transform is ~1000 scalar fp32 instructions per
// row and channel.  A real 96-point inverse transform of a Hermitian half spectrum is also the product of a constant 96 x 98 matrix with the row's 49
// complex entries, and the data of these handles is 16-bit already: T' is complex fp16 (11 significant bits).  v_mfma_f32_32x32x16_f16 multiplies it by
// the matrix -- held as TWO fp16 parts (hi + lo of the entry times 2^8: 22 significant bits, none of them denormal), so that the product is as exact as the
// fp32 butterflies' (relative error 2^-22 per term against their ~2^-23 per stage) -- at 1/30 of the vector ALU's cost per row.
//
//   x[n] = sum_kx w_kx (Re Y[kx] cos(2 pi kx n / 96) - Im Y[kx] sin(2 pi kx n / 96)),   w = 1 for kx = 0, 48 and 2 otherwise
//
// rows_inv_mfma_kernel (the contract of rows_inv_reg_kernel<96, 2, true>: T'[b][y][kx][c] complex fp16 in block floating point -> bias, ReLU, folded
// BatchNorm -> bf16 planar [B][C/8][H*W][8]):
//   * a WAVE owns one (image, row, 64 channels) tile = two 32-channel MFMA row tiles.  A operand = the data: lane (channel m, half h) holds the four
//     complex entries kx = 8 s + 4 h .. + 3 of k step s -- four 4-byte loads ARE the fragment (re, im adjacent, kx in separate registers), no transposition.
//     The block-floating-point scale of T' is per (image, kx, 64 channels): the entries are brought to the row's largest scale (exact: powers of two;
//     what falls below fp16's range is below 2^-29 of the row's largest bin) by one v_fma_mixlo/hi_f16 each.
//   * B operand = the matrix, [pixel tile 3][k step 7][hi, lo][lane] x 16 bytes = 42 KB in LDS, read once per k step for both row tiles.
//   * D[channel][pixel]: a lane holds ONE pixel and 16 channels -- with the channel order of the A rows permuted (bits 2 and 3 swapped) these are two whole
//     8-channel units of the planar layout: two 16-byte stores per fragment, 512-byte runs per store instruction.
#include <map>
#include <mutex>
#include <vector>

#include "conv_fft_common.h"

namespace jcm {
namespace cfft {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace rm {
constexpr int NX = 96, NXH = NX / 2 + 1;
constexpr int PT = 3;                 // 32-pixel tiles of the 96 samples
constexpr int KSI = 7;                // k steps of 16 = 8 complex entries: 49 entries -> 56
constexpr int TABI = PT * KSI * 2 * 64;      // uint4 units of the inverse matrix
constexpr float kTabScale = 256.f;    // the matrix entries are stored times 2^8 (the low parts stay normal fp16 numbers)
constexpr unsigned kZeroOff = 0x40000000u;      // a lane offset beyond any T' row: the buffer load returns zeros
}  // namespace rm

namespace {
// round to nearest even, |v| < 65504, result may be denormal (the low parts): via the compiler's own conversion
unsigned short f16_bits(double v) { return __builtin_bit_cast(unsigned short, static_cast<_Float16>(static_cast<float>(v))); }
double f16_val(unsigned short b) { return (double)static_cast<float>(__builtin_bit_cast(_Float16, b)); }

// [pixel tile][k step][hi, lo][lane][8 halves]: lane (n, h) = sample 32 T + n, entries k = 16 s + 8 h + e <-> kx = 8 s + 4 h + e / 2, re / im = e & 1
const uint4* inv_table(int dev) {
  static std::mutex mu;
  static std::map<int, uint4*> tabs;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find(dev);
  if (it != tabs.end()) return it->second;
  std::vector<unsigned short> hbuf((size_t)rm::TABI * 8);
  for (int T = 0; T < rm::PT; ++T)
    for (int s = 0; s < rm::KSI; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int n = 32 * T + (lane & 31), h = lane >> 5, kx = 8 * s + 4 * h + (e >> 1);
          double v = 0.0;
          if (kx < rm::NXH) {
            const double w = (kx == 0 || kx == rm::NX / 2) ? 1.0 : 2.0;
            const double ang = 2.0 * 3.14159265358979323846 * (double)((kx * n) % rm::NX) / (double)rm::NX;
            v = rm::kTabScale * w * ((e & 1) ? -std::sin(ang) : std::cos(ang));
          }
          const unsigned short hi = f16_bits(v), lo = f16_bits(v - f16_val(hi));
          hbuf[((((size_t)T * rm::KSI + s) * 2 + 0) * 64 + lane) * 8 + e] = hi;
          hbuf[((((size_t)T * rm::KSI + s) * 2 + 1) * 64 + lane) * 8 + e] = lo;
        }
  uint4* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), hbuf.size() * 2) != hipSuccess) return nullptr;
  if (hipMemcpy(d, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  tabs[dev] = d;
  return d;
}
}  // namespace

// both halves of a register times a per-lane power of two, as fp16 (exact unless the result is denormal)
__device__ __forceinline__ unsigned scale_h2(unsigned u, float r) {
  asm("v_fma_mixlo_f16 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(u) : "v"(r));
  asm("v_fma_mixhi_f16 %0, %0, %1, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u) : "v"(r));
  return u;
}

__global__ __launch_bounds__(256, 2) void rows_inv_mfma_kernel(const unsigned* __restrict__ T, __bf16* __restrict__ out, const float* __restrict__ bias,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int ntiles, int H, int W,
                                                               int C, int Cout, int pad, float norm0, Fp16Scale sc, const uint4* __restrict__ tab) {
  using namespace rm;
  __shared__ __attribute__((aligned(16))) uint4 tabl[TABI];
  __shared__ __attribute__((aligned(16))) float par[3][kParMax];
  __shared__ __attribute__((aligned(16))) float rts[4][64];      // per wave: the 49 scale factors of its tile (+ zeros)
  const int tid = threadIdx.x;
  for (int i = tid; i < TABI; i += 256) tabl[i] = tab[i];
  for (int i = tid; i < C; i += 256) {
    const bool in = i < Cout;
    par[0][i] = in ? bias[i] : 0.f;
    par[1][i] = in && relu_bn ? scale[i] : 1.f;
    par[2][i] = in && relu_bn ? shift[i] : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63, h = lane >> 5, n = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = C >> 6;
  const int chp = (n & 0x13) | ((n & 4) << 1) | ((n & 8) >> 1);      // channel of A row n: output register 4 q + i of lane half hq then is channel 16 (q / 2) + 8 hq + 4 (q & 1) + i
  const f16x8* tb = reinterpret_cast<const f16x8*>(tabl) + lane;
  float* rt = rts[wid];
  for (int wt = blockIdx.x * 4 + wid; wt < ntiles; wt += gridDim.x * 4) {
    const int by = __builtin_amdgcn_readfirstlane(wt / nblk), blk = __builtin_amdgcn_readfirstlane(wt % nblk);
    const int b = by / H, y = by - b * H;
    // ---- the tile's data: 2 row tiles x 7 k steps x 4 entries, all requested before anything else
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(T) + (size_t)by * NXH * C, 0, NXH * C * 4, 0x00020000);
    unsigned raw[2][KSI][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const unsigned vo = (unsigned)((blk * 64 + 32 * mt + chp) * 4 + 4 * h * C * 4);
#pragma unroll
      for (int s = 0; s < KSI; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (s < KSI - 1) raw[mt][s][i] = __builtin_amdgcn_raw_buffer_load_b32(d, vo, (8 * s + i) * C * 4, 0);
          else if (i == 0) raw[mt][s][i] = __builtin_amdgcn_raw_buffer_load_b32(d, h ? kZeroOff : vo, 8 * s * C * 4, 0);      // kx = 48: the last entry, lane half 0 only
          else raw[mt][s][i] = 0u;
        }
    }
    // ---- the scale words of T' for this (image, 64 channels): 49 powers of two, lane k holds word k; everything relative to the largest: 2^(e_k - e_max)
    // goes through the wave's LDS slice so that a lane picks up the four factors of a k step (entries 8 s + 4 h ..) with one 16-byte read.  (The first
    // version kept the 49 words in scalar registers and selected per lane half: a third of the kernel's vector instructions.)
    const float* ssrc = sc.t16_inv + ((size_t)b * nblk + blk) * NXH;
    const unsigned swl = lane < NXH ? __float_as_uint(ssrc[lane]) : 0u;
    unsigned smv = swl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned ot = (unsigned)__shfl_xor((int)smv, o); smv = ot > smv ? ot : smv; }
    const unsigned smax = (unsigned)__builtin_amdgcn_readfirstlane((int)smv);
    {
      const int rb = (int)(swl - smax) + 0x3f800000;      // 2^(e - e_max) as bits; an exponent field that would fall to 0 or below -> 0
      rt[lane] = (lane < NXH && rb >= 0x00800000) ? __int_as_float(rb) : 0.f;
    }
    float norm = norm0;
    if (sc.tmax) {
      float tm;
      if (sc.common) {
        tm = 0.f;
        for (int i = lane; i < sc.nb; i += 64) tm = fmaxf(tm, sc.tmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o));
      } else {
        tm = sc.tmax[b];
      }
      norm = norm0 * sc.winv[0] * fp16_unscale(tm, sc.hf);
    }
    norm *= __uint_as_float(smax) * (1.0f / kTabScale);      // powers of two: exact
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- to the common scale: entry kx of lane half h times 2^(e_kx - e_max)
#pragma unroll
    for (int s = 0; s < KSI; ++s) {
      const float4 r4 = *reinterpret_cast<const float4*>(rt + 4 * h + 8 * s);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        raw[mt][s][0] = scale_h2(raw[mt][s][0], r4.x);
        if (s < KSI - 1) {
          raw[mt][s][1] = scale_h2(raw[mt][s][1], r4.y);
          raw[mt][s][2] = scale_h2(raw[mt][s][2], r4.z);
          raw[mt][s][3] = scale_h2(raw[mt][s][3], r4.w);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();      // (the next tile's factors stay behind these reads)
    // ---- D[channel][pixel] += A[channel][k] B[k][pixel]
    f32x16 acc[2][PT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int s = 0; s < KSI; ++s) {
        const f16x8 bh = tb[((t * KSI + s) * 2 + 0) * 64], bl = tb[((t * KSI + s) * 2 + 1) * 64];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          typedef unsigned u4v __attribute__((ext_vector_type(4)));
          const f16x8 a = __builtin_bit_cast(f16x8, u4v{raw[mt][s][0], raw[mt][s][1], raw[mt][s][2], raw[mt][s][3]});
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc[mt][t], 0, 0, 0);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[mt][t], 0, 0, 0);
        }
      }
    // ---- epilogue: lane = pixel 32 t + n - pad, registers 8 g .. 8 g + 7 = channels 64 blk + 32 mt + 16 g + 8 h ..
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cu = blk * 64 + 32 * mt + 16 * g + 8 * h;
        float bi[8], scl[8], sh[8];
        *reinterpret_cast<float4*>(bi) = *reinterpret_cast<const float4*>(&par[0][cu]);
        *reinterpret_cast<float4*>(bi + 4) = *reinterpret_cast<const float4*>(&par[0][cu + 4]);
        *reinterpret_cast<float4*>(scl) = *reinterpret_cast<const float4*>(&par[1][cu]);
        *reinterpret_cast<float4*>(scl + 4) = *reinterpret_cast<const float4*>(&par[1][cu + 4]);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(&par[2][cu]);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(&par[2][cu + 4]);
        __bf16* ob = out + (((size_t)b * (Cout >> 3) + (cu >> 3)) * HW + (size_t)y * W) * 8;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const int xo = 32 * t + n - pad;
          unsigned w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v0 = fmaf(acc[mt][t][8 * g + 2 * j], norm, bi[2 * j]), v1 = fmaf(acc[mt][t][8 * g + 2 * j + 1], norm, bi[2 * j + 1]);
            if (relu_bn) { v0 = fmaf(fmaxf(v0, 0.f), scl[2 * j], sh[2 * j]); v1 = fmaf(fmaxf(v1, 0.f), scl[2 * j + 1], sh[2 * j + 1]); }
            w4[j] = __builtin_bit_cast(unsigned, bf16x2{static_cast<__bf16>(v0), static_cast<__bf16>(v1)});
          }
          if (xo >= 0 && xo < W && cu < Cout) *reinterpret_cast<uint4*>(ob + (size_t)xo * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
      }
  }
}


bool cfft_rows_inv_mfma_supported(int NX, const ConvArgs& a, int layout, int pad, const Fp16Scale& sc) {
  return NX == rm::NX && layout == 2 && sc.t16_inv && sc.t16_cb == 64 && a.CoutP % 64 == 0 && a.CoutP <= kParMax && a.Cout % 8 == 0 && a.W + pad <= rm::NX && pad >= 0 &&
         a.wout_TX == 0 && (size_t)rm::NXH * a.CoutP * 4 < rm::kZeroOff;
}
// true: launched
bool cfft_rows_inv_mfma(int NX, const ConvArgs& a, int layout, const cf* T, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  if (!cfft_rows_inv_mfma_supported(NX, a, layout, pad, sc)) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const uint4* tab = inv_table(dev);
  if (!tab) return false;
  const int ntiles = a.B * a.H * (a.CoutP / 64);
  const int blocks = persistent_grid(reinterpret_cast<const void*>(rows_inv_mfma_kernel), (ntiles + 3) / 4, 256);
  hipLaunchKernelGGL(rows_inv_mfma_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const unsigned*>(T), static_cast<__bf16*>(a.out), a.bias, a.scale, a.shift, a.relu_bn,
                     ntiles, a.H, a.W, a.CoutP, a.Cout, pad, norm, sc, tab);
  return true;
}

}  // namespace cfft
}  // namespace jcm
