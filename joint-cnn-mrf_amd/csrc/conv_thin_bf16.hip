// Thin-N convolution on the bf16 path: the logits layer conv6 (9x9, 512 -> 9, main.py:72) on
// v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 logits out.
//
// A 32x32 MFMA tile wastes 72 % of its columns on Cout = 9; the 16x16x32 shape pads 9 -> 16
// (56 % useful) and consumes a whole 32-channel chunk per instruction: lane l supplies pixel
// (l&15) x channel unit (l>>4) of A and channel (l&15) x unit (l>>4) of B, both one
// ds_read_b128 from the same [unit][slot][8 ch] LDS images the wide kernel uses.  Every A
// fragment feeds exactly one MFMA (N = 16), so the kernel lives on LDS reads and occupancy:
// measured at B=256, 3 waves x 4 rows 3.13 ms, 6 x 2 rows 1.92 ms, 12 waves x 1 row 1.83 ms
// (vs 3.44 ms for the 32x32 tile).  D layout: col = lane&15 (channel), row = 4*(lane>>4) + reg
// (pixel).  Workgroup = 12 waves x (1 row x 32 pixels) = 12x32 patch, 2 workgroups per CU.
#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace thinb {
constexpr int KS = 9, PAD = 4, NW = 12, RW = 1, TH = RW * NW, TW = 32, U = 4, CO = 16;
constexpr int HH = TH + KS - 1, WH = TW + KS - 1, WHP = WH;     // 20 x 40
constexpr int PLANE = HH * WHP;                                   // 800: a multiple of 16 keeps the 4 unit planes bank-aligned
static_assert(PLANE % 16 == 0, "unit planes must be 256-B aligned for conflict-free 16x16x32 A reads");
constexpr int HALO_F4 = U * PLANE;
constexpr int TPS = 9, NSTAGE = KS * KS / TPS;
constexpr int WSTAGE_F4 = TPS * U * CO;                           // 576 x 16 B = 9 KB
constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
constexpr int NT = NW * 64;
constexpr int WREG = (WSTAGE_F4 + NT - 1) / NT;
constexpr int MF = RW * 2;                                        // 16-pixel fragments per wave
}  // namespace thinb

__global__ __launch_bounds__(thinb::NT, 3) void conv_thin_bf16_kernel(ConvArgs a, int tiles_x, int tiles_y) {
  using namespace thinb;
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
  const __bf16* __restrict__ xb = static_cast<const __bf16*>(a.x) + (size_t)b * H * W * Cin;
  const __bf16* __restrict__ wp = static_cast<const __bf16*>(a.wp);   // [81][Cin/8][16][8]

  const int li = lane & 15, lq = lane >> 4;
  int aslot[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) aslot[f] = lq * PLANE + (RW * wid + (f >> 1)) * WHP + (f & 1) * 16 + li;
  const int bslot = lq * CO + li;
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
        const int co = idx % CO, tu = idx / CO;
        const int u = tu % U, tap = s * TPS + tu / U;
        wreg[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)tap * cin8 + chunk * U + u) * CO + co) * 8);
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
      // NHWC: the 4 units of a pixel are 64 contiguous bytes -> unit fastest.  Planar: a unit plane row is contiguous
      // -> pixel fastest (consecutive lanes read consecutive 16-byte slots of one plane).
      const int u = a.in_planar ? idx / (HH * WH) : idx & (U - 1);
      const int pix = a.in_planar ? idx - u * (HH * WH) : idx >> 2;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = a.in_planar ? *reinterpret_cast<const f32x4*>(xb + (((size_t)(chunk * U + u) * H + gy) * W + gx) * 8)     // [C/8][H*W][8] of image b
                        : *reinterpret_cast<const f32x4*>(xb + ((size_t)gy * W + gx) * Cin + chunk * 32 + u * 8);
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
        const bf16x8 bf = __builtin_bit_cast(bf16x8, wb[kx * U * CO + bslot]);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, halo[aslot[f] + toff]);
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[f], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }
  // epilogue: lane (channel li, row group lq): reg r = pixel 4*lq + r of the 16-pixel fragment
  if (li < Cout) {
    const float bi = a.bias[li];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      const int y = y0 + RW * wid + (f >> 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + (f & 1) * 16 + 4 * lq + r;
        if (y < H && x < W) {
          float v = acc[f][r] + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * a.scale[li] + a.shift[li];
          static_cast<float*>(a.out)[(((size_t)b * H + y) * W + x) * Cout + li] = v;
        }
      }
    }
  }
}

// conv 9x9 stride 1 SAME, bf16 in / fp32 out, Cout <= 16, Cin % 32 == 0; weights packed by
// pack_weights_bf16 with CoutP = 16.
hipError_t conv_thin_bf16(const ConvArgs& a, hipStream_t st) {
  using namespace thinb;
  if (a.Cout > CO || a.CoutP != CO || a.Cin % 32) return hipErrorInvalidValue;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  static LdsAttr attr;   // per device, not per process: a second Engine on another GPU needs its own call
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_thin_bf16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_thin_bf16_kernel, dim3(tiles_x * tiles_y * a.B), dim3(NT), LDS_BYTES, st, a, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace jcm
