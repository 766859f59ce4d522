// The logits layer conv6 (9x9, 512 -> 9, main.py:72) for fp32 handles in "f32_conv" = 2 mode: fp32 operands as two
// fp16 parts, three products (conv_split.hip's fp16x3 scheme) on v_mfma_f32_16x16x32_f16 -- the thin-N dataflow of
// conv_thin_bf16.hip (12 waves x one 32-pixel row, Cout padded 9 -> 16, one ds_read_b128 per operand fragment).
// Activations are split while the halo is written to LDS; weights arrive pre-split (times 2^12) from
// pack_weights_split(ns = 2) with CoutP = 16.
#include "kernels.h"

namespace jcm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace thins {
constexpr int KS = 9, PAD = 4, NW = 12, TH = NW, TW = 32, U = 4, CO = 16, NS = 2;
constexpr int HH = TH + KS - 1, WH = TW + KS - 1, WHP = WH;     // 20 x 40
constexpr int PLANE = HH * WHP;                                   // 800: a multiple of 16 keeps the unit planes bank-aligned
constexpr int HALO_F4 = NS * U * PLANE;                           // [part][unit][slot]
constexpr int TPS = 9, NSTAGE = KS * KS / TPS;                    // one kernel row per stage
constexpr int WSTAGE_F4 = TPS * U * NS * CO;                      // [tap][unit][part][co]: as packed, contiguous per (tap, unit)
constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
constexpr int NT = NW * 64;
constexpr int WREG = (WSTAGE_F4 + NT - 1) / NT;
constexpr int MF = 2;                                             // 16-pixel fragments per wave
constexpr float kScaleInv = 1.0f / 4096.0f;                       // weights are stored times 2^12 (conv_split.hip: kW16Scale)
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace thins

__global__ __launch_bounds__(thins::NT, 3) void conv_thin_split16_kernel(ConvArgs a, int tiles_x, int tiles_y) {
  using namespace thins;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* halo = reinterpret_cast<f32x4*>(smem);
  f32x4* wbuf = halo + HALO_F4;
  const int mt = blockIdx.x;
  const int tx = mt % tiles_x;
  const int ty = (mt / tiles_x) % tiles_y;
  const int b = mt / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const float* __restrict__ xb = static_cast<const float*>(a.x) + (size_t)b * H * W * Cin;
  const _Float16* __restrict__ wp = static_cast<const _Float16*>(a.wp);   // [81][Cin/8][2][16][8]

  const int li = lane & 15, lq = lane >> 4;
  int aslot[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) aslot[f] = lq * PLANE + wid * WHP + f * 16 + li;
  const int bslot = lq * NS * CO + li;                                   // unit lq, part 0, channel li
  f32x4 acc[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cin8 = Cin >> 3;
  f32x4 wreg[WREG];
  auto wload = [&](int chunk, int s) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * NT;
      if (idx < WSTAGE_F4) {
        const int pc = idx % (NS * CO), tu = idx / (NS * CO);            // (part, co) is contiguous in the packed image
        const int u = tu % U, tap = s * TPS + tu / U;
        wreg[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)tap * cin8 + chunk * U + u) * NS * CO + pc) * 8);
      }
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * NT;
      if (idx < WSTAGE_F4) wbuf[buf * WSTAGE_F4 + idx] = wreg[i];
    }
  };

  const int nchunk = Cin >> 5;
  int buf = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
    for (int idx = tid; idx < U * HH * WH; idx += NT) {
      const int u = idx & (U - 1);
      const int pix = idx >> 2;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        const float* src = xb + ((size_t)gy * W + gx) * Cin + chunk * 32 + u * 8;
        lo = *reinterpret_cast<const f32x4*>(src);
        hi = *reinterpret_cast<const f32x4*>(src + 4);
        if (a.in_scale) {
          const float S = a.in_scale[0];
#pragma unroll
          for (int i = 0; i < 4; ++i) { lo[i] *= S; hi[i] *= S; }
        }
      }
      f16x8 p0, p1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = i < 4 ? lo[i] : hi[i - 4];
        const _Float16 h0 = static_cast<_Float16>(v);
        p0[i] = h0;
        p1[i] = static_cast<_Float16>(v - static_cast<float>(h0));
      }
      halo[(0 * U + u) * PLANE + hy * WHP + hx] = __builtin_bit_cast(f32x4, p0);
      halo[(1 * U + u) * PLANE + hy * WHP + hx] = __builtin_bit_cast(f32x4, p1);
    }
    wload(chunk, 0);
    for (int s = 0; s < NSTAGE; ++s) {     // s = kernel row ky
      wstore(buf);
      __syncthreads();
      if (s + 1 < NSTAGE) wload(chunk, s + 1);
      const f32x4* wb = wbuf + buf * WSTAGE_F4;
#pragma unroll
      for (int kx = 0; kx < TPS; ++kx) {
        const int toff = s * WHP + kx;
        const f16x8 b0 = __builtin_bit_cast(f16x8, wb[kx * U * NS * CO + bslot]);
        const f16x8 b1 = __builtin_bit_cast(f16x8, wb[kx * U * NS * CO + bslot + CO]);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          const f16x8 a0 = __builtin_bit_cast(f16x8, halo[aslot[f] + toff]);
          const f16x8 a1 = __builtin_bit_cast(f16x8, halo[U * PLANE + aslot[f] + toff]);
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[f], 0, 0, 0);
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[f], 0, 0, 0);
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[f], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }
  // epilogue: lane (channel li, row group lq): reg r = pixel 4*lq + r of the 16-pixel fragment
  if (li < Cout) {
    const float bi = a.bias[li];
    const float unscale = (a.w_scale ? a.w_scale[1] : kScaleInv) * (a.in_scale ? a.in_scale[1] : 1.0f);
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      const int y = y0 + wid;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + f * 16 + 4 * lq + r;
        if (y < H && x < W) {
          float v = acc[f][r] * unscale + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * a.scale[li] + a.shift[li];
          static_cast<float*>(a.out)[(((size_t)b * H + y) * W + x) * Cout + li] = v;
        }
      }
    }
  }
}

// conv 9x9 stride 1 SAME, fp32 in / out, Cout <= 16, Cin % 32 == 0; weights from pack_weights_split(ns = 2) with CoutP = 16
hipError_t conv_thin_split16(const ConvArgs& a, hipStream_t st) {
  using namespace thins;
  if (a.Cout > CO || a.CoutP != CO || a.Cin % 32) return hipErrorInvalidValue;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  static LdsAttr attr;   // per device, not per process: a second Engine on another GPU needs its own call
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_thin_split16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_thin_split16_kernel, dim3(tiles_x * tiles_y * a.B), dim3(NT), LDS_BYTES, st, a, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace jcm
