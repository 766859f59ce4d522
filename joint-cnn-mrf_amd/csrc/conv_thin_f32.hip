// Thin-N convolution on the fp32 path: the logits layer conv6 (9x9, 512 -> 9, main.py:72).
//
// With Cout = 9 the 32x32 MFMA tile of conv_igemm.hip spends 72 % of its columns on padding
// (33 TFLOP/s algorithmic).  v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 outer
// products per instruction at the same 64 FLOP/clk/SIMD: block b of the wave takes pixels
// 4b..4b+3 (A, lane l = pixel l) and 4 output channels (B, lane l = channel l&3), so 9
// channels pad only to 12 (75 % useful).  Measured layout (tools/mfma4x4_probe.hip):
//   A: lane l -> A[block l>>2][i = l&3];  B: lane l -> B[block l>>2][j = l&3];
//   D: reg r of lane l -> D[block l>>2][i = r][j = l&3].
// Dataflow as conv_igemm.hip: 16-channel chunk halo in LDS, one kernel row (9 taps) of
// packed weights [tap][unit][16 co][4 ch] per stage, double buffered through registers.
// A wave owns 2 rows x 32 pixels; a workgroup (6 waves) a 12x32 patch; fp32 throughout (exact FMA chain).
#include "kernels.h"

namespace jcm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace thin {
// 6 waves x (2 rows x 32 px) = 12x32 patch (tiles 60 rows exactly); NG groups of 4 channels = 12 >= Cout
constexpr int KS = 9, PAD = 4, NW = 6, TH = 2 * NW, TW = 32, U = 4, CO = 16, NG = 3;
constexpr int HH = TH + KS - 1, WH = TW + KS - 1, WHP = 48;
constexpr int PLANE = HH * WHP + 2;
constexpr int HALO_F4 = U * PLANE;
constexpr int TPS = 9, NSTAGE = KS * KS / TPS;
constexpr int WSTAGE_F4 = TPS * U * CO;             // 576 x 16 B = 9 KB
constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
constexpr int NT = NW * 64;
constexpr int WREG = (WSTAGE_F4 + NT - 1) / NT;
}  // namespace thin

__global__ __launch_bounds__(thin::NT, 3) void conv_thin_f32_kernel(ConvArgs a, int tiles_x, int tiles_y) {
  using namespace thin;
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
  const float* __restrict__ wp = static_cast<const float*>(a.wp);   // [81][Cin/4][16][4]

  // lane l = pixel l of the wave's 2x32 strip (A operand); j = l&3 = channel inside a group (B)
  const int prow = 2 * wid + (lane >> 5), pcol = lane & 31;
  const int aslot = prow * WHP + pcol;
  const int j = lane & 3;
  f32x4 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cin4 = Cin >> 2;
  f32x4 wreg[WREG];
  auto wload = [&](int chunk, int s) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * NT;
      if (idx < WSTAGE_F4) {
        const int co = idx % CO, tu = idx / CO;
        const int u = tu % U, tap = s * TPS + tu / U;
        wreg[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)tap * cin4 + chunk * U + u) * CO + co) * 4);
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

  const int nchunk = Cin >> 4;
  int buf = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
    for (int idx = tid; idx < U * HH * WH; idx += NT) {
      const int u = idx & (U - 1);
      const int pix = idx >> 2;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(xb + ((size_t)gy * W + gx) * Cin + chunk * 16 + u * 4);
      halo[u * PLANE + hy * WHP + hx] = v;
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
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const f32x4 af = halo[u * PLANE + aslot + toff];
          f32x4 bf[NG];
#pragma unroll
          for (int g = 0; g < NG; ++g) bf[g] = wb[(kx * U + u) * CO + g * 4 + j];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[t], bf[g][t], acc[g], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }
  // epilogue: reg r of lane l = pixel 4*(l>>2) + r of the strip, channel 4g + (l&3); linear layer (bias only)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int co = g * 4 + j;
    if (co >= Cout) continue;
    const float bi = a.bias[co];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 4 * (lane >> 2) + r;                 // 0..63 inside the strip
      const int y = y0 + 2 * wid + (p >> 5), x = x0 + (p & 31);
      if (y < H && x < W) {
        float v = acc[g][r] + bi;
        if (a.relu_bn) v = fmaxf(v, 0.f) * a.scale[co] + a.shift[co];
        static_cast<float*>(a.out)[(((size_t)b * H + y) * W + x) * Cout + co] = v;
      }
    }
  }
}

// conv 9x9 stride 1 SAME with Cout <= 12, Cin % 16 == 0; weights packed by pack_weights_f32 with CoutP = 16.
hipError_t conv_thin_f32(const ConvArgs& a, hipStream_t st) {
  using namespace thin;
  if (a.Cout > 4 * NG || a.CoutP != CO || a.Cin % 16) return hipErrorInvalidValue;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  static LdsAttr attr;   // per device, not per process: a second Engine on another GPU needs its own call
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_thin_f32_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_thin_f32_kernel, dim3(tiles_x * tiles_y * a.B), dim3(NT), LDS_BYTES, st, a, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace jcm
