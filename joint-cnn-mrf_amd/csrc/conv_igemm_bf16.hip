// Implicit-GEMM SAME convolution (stride 1, 5x5 / 9x9) on bf16 MFMA, fp32 accumulate: the
// roofline path (BASELINE.json configs[2]).  Same dataflow as conv_igemm.hip --
//     for 32-channel chunk:   halo of the pixel patch -> LDS once, reused by all k*k taps
//       for tap stage:        TPS taps of packed weights [unit][BN][8 ch] -> LDS (double buffer)
//         v_mfma_f32_32x32x16_bf16
// -- re-tiled for the 16x higher MFMA rate: 8 waves (2 per SIMD) per workgroup, a 6x32 /
// 12x16 pixel patch (192 pixels: 60x90 maps tile with 6 % padding) x up to 256 output
// channels, so a stage carries 36 MFMAs per wave (TPS=3) between barriers and the weight
// stream is 5 B per kFLOP out of L2.
//
// A 16-byte LDS unit holds 8 consecutive input channels (bf16); MFMA operand k = 8*(lane>>5)+i
// maps half-wave h to unit 2s+h of k16-step s, identically for A (halo) and B (weights).
// Activations are bf16 NHWC; the last layer (conv6) writes fp32 logits.
#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS_, int TH_, int TW_, int BN_, int WM_, int WN_, int TPS_>
struct CfgB {
  static constexpr int KS = KS_, TH = TH_, TW = TW_, BN = BN_, WM = WM_, WN = WN_, TPS = TPS_;
  static constexpr int NT = WM * WN * 64;           // threads
  static constexpr int U = 4;                       // 16-B units per chunk = 32 bf16 channels
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int HH = TH + KS - 1;
  static constexpr int WH = TW + KS - 1;
  static constexpr int WHP = (WH + 15) / 16 * 16;
  static constexpr int PLANE = HH * WHP + 2;
  static constexpr int BM = TH * TW;
  static constexpr int MR = BM / WM / 32;
  static constexpr int NR = BN / WN / 32;
  static constexpr int HALO_F4 = U * PLANE;
  static constexpr int WSTAGE_F4 = TPS * U * BN;
  static constexpr int NSTAGE = KS * KS / TPS;
  static constexpr int WREG = (WSTAGE_F4 + NT - 1) / NT;
  static constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile must split into 32x32 fragments");
  static_assert(KS * KS % TPS == 0, "stages must tile the taps");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C, bool OUT_F32>
__global__ __launch_bounds__(C::NT, (C::NT + 255) / 256) void conv_igemm_bf16_kernel(ConvArgs a, int tiles_x, int tiles_y, int mtiles, int nN) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* halo = reinterpret_cast<f32x4*>(smem);
  f32x4* wbuf = halo + C::HALO_F4;

  const int L = blockIdx.x;
  int mt, nt;
  if ((8 % nN) == 0) {   // an XCD (blocks b, b+8, ...) keeps one channel tile: its L2 streams 1/nN of the weights
    const int xcd = L & 7, q = L >> 3, per = 8 / nN;
    nt = xcd % nN;
    mt = q * per + xcd / nN;
  } else {
    nt = L % nN;
    mt = L / nN;
  }
  if (mt >= mtiles) return;
  const int tx = mt % tiles_x;
  const int ty = (mt / tiles_x) % tiles_y;
  const int b = mt / (tiles_x * tiles_y);
  const int y0 = ty * C::TH, x0 = tx * C::TW, n0 = nt * C::BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid % C::WN;
  const int h = lane >> 5, l31 = lane & 31;

  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CoutP = a.CoutP;
  const __bf16* __restrict__ xb = static_cast<const __bf16*>(a.x) + (size_t)b * H * W * Cin;
  const __bf16* __restrict__ wp = static_cast<const __bf16*>(a.wp);

  int aslot[C::MR], bcol[C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f) {
    const int r = (wm * C::MR + f) * 32 + l31;
    aslot[f] = (r / C::TW) * C::WHP + (r % C::TW);
  }
#pragma unroll
  for (int g = 0; g < C::NR; ++g) bcol[g] = (wn * C::NR + g) * 32 + l31;

  f32x16 acc[C::MR][C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f)
#pragma unroll
    for (int g = 0; g < C::NR; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  const int cin8 = Cin >> 3;
  f32x4 wreg[C::WREG];

  auto wload = [&](int chunk, int s) {
#pragma unroll
    for (int i = 0; i < C::WREG; ++i) {
      const int idx = tid + i * C::NT;
      if ((C::WSTAGE_F4 % C::NT == 0) || idx < C::WSTAGE_F4) {
        const int co = idx % C::BN;
        const int tu = idx / C::BN;
        const int u = tu % C::U, tp = tu / C::U;
        const int tap = s * C::TPS + tp;
        wreg[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)tap * cin8 + chunk * C::U + u) * CoutP + n0 + co) * 8);
      }
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::WREG; ++i) {
      const int idx = tid + i * C::NT;
      if ((C::WSTAGE_F4 % C::NT == 0) || idx < C::WSTAGE_F4) wbuf[buf * C::WSTAGE_F4 + idx] = wreg[i];
    }
  };

  const int nchunk = Cin >> 5;
  int buf = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
    for (int idx = tid; idx < C::U * C::HH * C::WH; idx += C::NT) {
      const int u = idx & (C::U - 1);
      const int pix = idx >> 2;
      const int hy = pix / C::WH, hx = pix - hy * C::WH;
      const int gy = y0 - C::PAD + hy, gx = x0 - C::PAD + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(xb + ((size_t)gy * W + gx) * Cin + chunk * 32 + u * 8);
      halo[u * C::PLANE + hy * C::WHP + hx] = v;
    }
    wload(chunk, 0);
    for (int s = 0; s < C::NSTAGE; ++s) {
      wstore(buf);
      __syncthreads();
      if (s + 1 < C::NSTAGE) wload(chunk, s + 1);
      const f32x4* wb = wbuf + buf * C::WSTAGE_F4;
      const int tap0 = s * C::TPS;
#pragma unroll
      for (int tp = 0; tp < C::TPS; ++tp) {
        const int tap = tap0 + tp;
        const int ky = tap / C::KS, kx = tap - ky * C::KS;
        const int toff = ky * C::WHP + kx;
#pragma unroll
        for (int st = 0; st < C::U / 2; ++st) {
          const int u = st * 2 + h;
          bf16x8 af[C::MR], bf[C::NR];
#pragma unroll
          for (int f = 0; f < C::MR; ++f) af[f] = __builtin_bit_cast(bf16x8, halo[u * C::PLANE + aslot[f] + toff]);
#pragma unroll
          for (int g = 0; g < C::NR; ++g) bf[g] = __builtin_bit_cast(bf16x8, wb[(tp * C::U + u) * C::BN + bcol[g]]);
#pragma unroll
          for (int f = 0; f < C::MR; ++f)
#pragma unroll
            for (int g = 0; g < C::NR; ++g)
              acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[f], bf[g], acc[f][g], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }

  // epilogue: bias (+ ReLU + folded BN) -> bf16 NHWC (fp32 for the logits layer)
#pragma unroll
  for (int g = 0; g < C::NR; ++g) {
    const int co = n0 + bcol[g];
    if (co >= Cout) continue;
    const float bi = a.bias[co];
    float sc = 1.f, sh = 0.f;
    if (a.relu_bn) { sc = a.scale[co]; sh = a.shift[co]; }
#pragma unroll
    for (int f = 0; f < C::MR; ++f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (wm * C::MR + f) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int y = y0 + r / C::TW, x = x0 + r % C::TW;
        if (y < H && x < W) {
          float v = acc[f][g][i] + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc + sh;
          const size_t o = (((size_t)b * H + y) * W + x) * Cout + co;
          if (OUT_F32) static_cast<float*>(a.out)[o] = v;
          else static_cast<__bf16*>(a.out)[o] = static_cast<__bf16>(v);
        }
      }
    }
  }
}

template <class C, bool OUT_F32>
static hipError_t launch_b(const ConvArgs& a, hipStream_t st) {
  const int tiles_x = (a.W + C::TW - 1) / C::TW, tiles_y = (a.H + C::TH - 1) / C::TH;
  const int mtiles = tiles_x * tiles_y * a.B;
  const int nN = a.CoutP / C::BN;
  int blocks;
  if ((8 % nN) == 0) {
    const int per = 8 / nN;
    blocks = (mtiles + per - 1) / per * 8;
  } else {
    blocks = mtiles * nN;
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_bf16_kernel<C, OUT_F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_bf16_kernel<C, OUT_F32>), dim3(blocks), dim3(C::NT), C::LDS_BYTES, st, a, tiles_x, tiles_y, mtiles, nN);
  return hipGetLastError();
}

// N-tile: 256 channels for the 9x9 layers (weight stream 5 B/kFLOP); the 5x5 layers stay at 128 so
// that a 5-tap stage double-buffers inside 160 KB of LDS.
int conv_igemm_bf16_bn(int Cout, int ks) { return (Cout >= 256 && ks == 9) ? 256 : (Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32)); }

hipError_t conv_igemm_bf16(const ConvArgs& a, int ks, bool out_f32, hipStream_t st) {
  const int bn = conv_igemm_bf16_bn(a.Cout, ks);
  const bool wide = a.W >= 64;           // 6x32 patches; narrow maps use 12x16
  if (out_f32) {                         // logits layer (conv6): Cout = 9 -> one 32-channel tile
    if (ks != 9 || bn != 32) return hipErrorInvalidValue;
    return wide ? launch_b<CfgB<9, 6, 32, 32, 6, 1, 9>, true>(a, st) : launch_b<CfgB<9, 12, 16, 32, 6, 1, 9>, true>(a, st);
  }
  if (ks == 9) {
    if (bn == 256) return wide ? launch_b<CfgB<9, 6, 32, 256, 2, 4, 3>, false>(a, st) : launch_b<CfgB<9, 12, 16, 256, 2, 4, 3>, false>(a, st);
    if (bn == 128) return wide ? launch_b<CfgB<9, 6, 32, 128, 2, 4, 3>, false>(a, st) : launch_b<CfgB<9, 12, 16, 128, 2, 4, 3>, false>(a, st);
    if (bn == 64) return wide ? launch_b<CfgB<9, 6, 32, 64, 2, 2, 9>, false>(a, st) : launch_b<CfgB<9, 12, 16, 64, 2, 2, 9>, false>(a, st);
    return wide ? launch_b<CfgB<9, 6, 32, 32, 2, 1, 9>, false>(a, st) : launch_b<CfgB<9, 12, 16, 32, 2, 1, 9>, false>(a, st);
  }
  if (ks == 5) {
    if (bn == 128) return wide ? launch_b<CfgB<5, 6, 32, 128, 2, 4, 5>, false>(a, st) : launch_b<CfgB<5, 12, 16, 128, 2, 4, 5>, false>(a, st);
    if (bn == 64) return wide ? launch_b<CfgB<5, 6, 32, 64, 2, 2, 5>, false>(a, st) : launch_b<CfgB<5, 12, 16, 64, 2, 2, 5>, false>(a, st);
    return wide ? launch_b<CfgB<5, 6, 32, 32, 2, 1, 5>, false>(a, st) : launch_b<CfgB<5, 12, 16, 32, 2, 1, 5>, false>(a, st);
  }
  return hipErrorInvalidValue;
}

// fp32 HWIO [k,k,Cin,Cout] -> bf16 [tap][Cin/8][CoutP][8] (round to nearest even), zero-padded channels.
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int taps, int Cin, int Cout, int CoutP) {
  const size_t n = (size_t)taps * Cin * CoutP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int k8 = i & 7;
    size_t r = i >> 3;
    const int co = r % CoutP; r /= CoutP;
    const int c8 = r % (Cin >> 3);
    const int tap = r / (Cin >> 3);
    const int ci = c8 * 8 + k8;
    wp[i] = static_cast<__bf16>(co < Cout ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f);
  }
}

hipError_t pack_weights_bf16(const float* w_hwio, void* wp, int ks, int Cin, int Cout, int CoutP, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(2048), dim3(256), 0, st, w_hwio, static_cast<__bf16*>(wp), ks * ks, Cin, Cout, CoutP);
  return hipGetLastError();
}

}  // namespace jcm
