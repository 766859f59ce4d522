// Implicit-GEMM SAME convolution (stride 1, 5x5 / 9x9) for gfx950 on fp32-input MFMA.
//
// GEMM view: M = output pixels, N = Cout, K = k*k*Cin.  A workgroup owns a
// TH x TW pixel patch (128 pixels) x BN output channels and walks K as
//     for 16-channel chunk c:   stage the (TH+k-1) x (TW+k-1) input halo of chunk c in LDS once
//       for tap stage s:        stage TPS taps of packed weights [unit][BN][4 ch] (double buffer)
//         v_mfma_f32_32x32x2_f32 over the stage, A read from the halo at the tap's offset
// so every input element is fetched from HBM/L2 once per chunk and reused by all k*k taps out
// of LDS; only the weight slab streams.  v_mfma_f32_32x32x2_f32 is an exact-f32 FMA chain
// (this is the parity path: no reduced-precision operand anywhere).
//
// LDS images are [16-byte unit][slot][4 floats]: a lane's ds_read_b128 delivers 4 consecutive
// K values of its row/column, and lanes of one 16-lane read group hit 16 distinct 16-B slots
// (halo rows are padded to a multiple of 16 slots), i.e. conflict-free ds_read_b128.
// The K order inside an 8-channel step is permuted (half-wave h takes
// channels 4h..4h+3) identically for A and B, which MFMA permits because K is a contraction.
#include "kernels.h"

namespace jcm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// STRIP tiling (host side: make_strips): the map is cut into column strips (32 wide, plus the remainder; a narrow map is one
// strip) and every strip is walked in its own row-major pixel order, 128 pixels per tile regardless of row ends.  A 60x90
// map costs 15 + 15 + 13 = 43 tiles instead of the 45 of 4x32 patches (the 26-wide remainder strip wastes 1.6 % instead of
// 19 %), a 30x45 map 11 instead of the 15 of two-row tiles.
struct Strips {
  int n;                 // strips used
  int x_off[8], ws[8];   // first column / width
  int tile0[9];          // first tile of strip i within an image; tile0[n] = tiles per image
};

template <int KS_, int TH_, int TW_, int BN_, int WM_, int WN_, int TPS_, bool FLAT_ = false, bool STRIP_ = false>
struct Cfg {
  static constexpr int KS = KS_, TH = TH_, TW = TW_, BN = BN_, WM = WM_, WN = WN_, TPS = TPS_;
  // FLAT: the 128 pixel slots are R whole rows of a narrow map (R = floor(128 / W), set at launch)
  // instead of a TH x TW patch: a 15x23 map costs 3 tiles x 128 slots instead of 4 x 128.
  static constexpr bool FLAT = FLAT_;
  static constexpr bool STRIP = STRIP_;
  static constexpr int U = 4;                       // 16-B units per chunk = 16 fp32 channels
  static constexpr int PAD = (KS - 1) / 2;          // SAME, stride 1: symmetric
  static constexpr int HH = TH + KS - 1;
  static constexpr int WH = TW + KS - 1;
  static constexpr int WHP = (WH + 15) / 16 * 16;   // halo row pitch in slots (bank-conflict-free reads)
  // +2 slots: spreads the 4 unit planes over banks on the write side.  STRIP keeps the patch tiling's LDS footprint (3 workgroups
  // per CU): strips up to 32 wide need (5 + k-1) x (32 + k-1) slots at most
  static constexpr int PLANE = HH * WHP + 2;
  static constexpr int BM = TH * TW;
  static constexpr int MR = BM / WM / 32;
  static constexpr int NR = BN / WN / 32;
  static constexpr int HALO_F4 = U * PLANE;
  static constexpr int WSTAGE_F4 = TPS * U * BN;
  static constexpr int NSTAGE = KS * KS / TPS;
  static constexpr int WREG = (WSTAGE_F4 + 255) / 256;
  static constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
  static_assert(BM == 128 && WM * WN == 4, "4 waves per 128-pixel patch");
  static_assert(KS * KS % TPS == 0, "stages must tile the taps");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(ConvArgs a, int tiles_x, int tiles_y, int mtiles, int nN, Strips sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* halo = reinterpret_cast<f32x4*>(smem);
  f32x4* wbuf = halo + C::HALO_F4;

  // ---- block -> (pixel tile, channel tile).  Blocks b and b+8 run on the same XCD; give an XCD
  // a single channel tile so its private L2 streams only 1/nN of the packed weights.
  const int L = blockIdx.x;
  int mt, nt;
  if ((8 % nN) == 0) {
    const int xcd = L & 7, q = L >> 3, per = 8 / nN;
    nt = xcd % nN;
    mt = q * per + xcd / nN;
  } else {
    nt = L % nN;
    mt = L / nN;
  }
  if (mt >= mtiles) return;
  const int flat_r = C::FLAT ? tiles_x : 0;        // FLAT: `tiles_x` carries R, one tile across
  const int tx = C::FLAT ? 0 : mt % tiles_x;
  const int ty = C::FLAT ? mt % tiles_y : (mt / tiles_x) % tiles_y;
  int b = C::FLAT ? mt / tiles_y : mt / (tiles_x * tiles_y);
  int y0 = ty * (C::FLAT ? flat_r : C::TH), x0 = tx * C::TW;
  const int n0 = nt * C::BN;
  // STRIP: tile -> (image, strip, first pixel of the tile in the strip's row-major order)
  int s_ws = 1, s_p0 = 0, s_rows = 0;
  if constexpr (C::STRIP) {
    const int tpi = sp.tile0[sp.n];
    b = mt / tpi;
    const int ti = mt - b * tpi;
    int si = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k < sp.n && ti >= sp.tile0[k]) si = k;
    s_ws = sp.ws[si];
    s_p0 = (ti - sp.tile0[si]) * C::BM;
    x0 = sp.x_off[si];
    y0 = s_p0 / s_ws;                                       // first row the tile touches
    const int ylast = min((s_p0 + C::BM - 1) / s_ws, a.H - 1);
    s_rows = ylast - y0 + 1;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid % C::WN;
  const int h = lane >> 5, l31 = lane & 31;

  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CoutP = a.CoutP;
  const float* __restrict__ xb = static_cast<const float*>(a.x) + (size_t)b * H * W * Cin;
  const float* __restrict__ wp = static_cast<const float*>(a.wp);

  const int whp = C::STRIP ? s_ws + C::KS - 1 : (C::FLAT ? W + C::KS - 1 : C::WHP);
  const int wh = C::STRIP ? s_ws + C::KS - 1 : (C::FLAT ? W + C::KS - 1 : C::WH);
  const int hh = C::STRIP ? s_rows + C::KS - 1 : (C::FLAT ? flat_r + C::KS - 1 : C::HH);
  int aslot[C::MR], bcol[C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f) {
    const int r = (wm * C::MR + f) * 32 + l31;
    if constexpr (C::STRIP) {
      const int p = s_p0 + r < H * s_ws ? s_p0 + r : s_p0;   // slots past the strip's end compute on its first pixel and are dropped
      const int yy = p / s_ws;
      aslot[f] = (yy - y0) * whp + (p - yy * s_ws);
    } else if constexpr (C::FLAT) {
      const int rr = r < flat_r * W ? r : 0;          // padding slots compute on pixel 0 and are dropped
      const int yy = rr / W;
      aslot[f] = yy * whp + (rr - yy * W);
    } else {
      aslot[f] = (r / C::TW) * C::WHP + (r % C::TW);
    }
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

  const int cin4 = Cin >> 2;
  f32x4 wreg[C::WREG];

  auto wload = [&](int chunk, int s) {
#pragma unroll
    for (int i = 0; i < C::WREG; ++i) {
      const int idx = tid + i * 256;
      if ((C::WSTAGE_F4 % 256 == 0) || idx < C::WSTAGE_F4) {
        const int co = idx % C::BN;
        const int tu = idx / C::BN;
        const int u = tu % C::U, tp = tu / C::U;
        const int tap = s * C::TPS + tp;
        wreg[i] = *reinterpret_cast<const f32x4*>(wp + (((size_t)tap * cin4 + chunk * C::U + u) * CoutP + n0 + co) * 4);
      }
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::WREG; ++i) {
      const int idx = tid + i * 256;
      if ((C::WSTAGE_F4 % 256 == 0) || idx < C::WSTAGE_F4) wbuf[buf * C::WSTAGE_F4 + idx] = wreg[i];
    }
  };

  const int nchunk = Cin >> 4;
  int buf = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();  // every wave is done reading the previous chunk's halo and weight buffers
    // ---- input halo of this 16-channel chunk -> LDS (zero fill = SAME padding)
    for (int idx = tid; idx < C::U * hh * wh; idx += 256) {
      const int u = idx & (C::U - 1);
      const int pix = idx >> 2;
      const int hy = pix / wh, hx = pix - hy * wh;
      const int gy = y0 - C::PAD + hy, gx = x0 - C::PAD + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(xb + ((size_t)gy * W + gx) * Cin + chunk * 16 + u * 4);
      halo[u * C::PLANE + hy * whp + hx] = v;
    }
    wload(chunk, 0);
    for (int s = 0; s < C::NSTAGE; ++s) {
      wstore(buf);
      __syncthreads();
      if (s + 1 < C::NSTAGE) wload(chunk, s + 1);  // in flight behind this stage's MFMAs
      const f32x4* wb = wbuf + buf * C::WSTAGE_F4;
      const int tap0 = s * C::TPS;
#pragma unroll
      for (int tp = 0; tp < C::TPS; ++tp) {
        const int tap = tap0 + tp;
        const int ky = tap / C::KS, kx = tap - ky * C::KS;
        const int toff = ky * whp + kx;
#pragma unroll
        for (int st = 0; st < C::U / 2; ++st) {
          const int u = st * 2 + h;
          f32x4 af[C::MR], bf[C::NR];
#pragma unroll
          for (int f = 0; f < C::MR; ++f) af[f] = halo[u * C::PLANE + aslot[f] + toff];
#pragma unroll
          for (int g = 0; g < C::NR; ++g) bf[g] = wb[(tp * C::U + u) * C::BN + bcol[g]];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int f = 0; f < C::MR; ++f)
#pragma unroll
              for (int g = 0; g < C::NR; ++g)
                acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[f][t], bf[g][t], acc[f][g], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }

  // ---- epilogue: bias (+ ReLU + folded BatchNorm), NHWC store.  C/D layout of the 32x32 MFMA:
  // col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel).
  float* __restrict__ ob = static_cast<float*>(a.out) + (size_t)b * H * W * Cout;
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
        int y, x;
        bool ok;
        if constexpr (C::STRIP) {
          const int p = s_p0 + r;
          y = p / s_ws; x = x0 + (p - y * s_ws);
          ok = p < H * s_ws;
        } else if constexpr (C::FLAT) {
          const int yy = r / W;
          y = y0 + yy; x = r - yy * W;
          ok = r < flat_r * W && y < H;
        } else {
          y = y0 + r / C::TW; x = x0 + r % C::TW;
          ok = y < H && x < W;
        }
        if (ok) {
          float v = acc[f][g][i] + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc + sh;
          ob[((size_t)y * W + x) * Cout + co] = v;
        }
      }
    }
  }
}

// column strips of an H x W map for the STRIP tiling: the whole map if its halo fits, else 32-wide strips + the remainder
// (a remainder thinner than 8 columns joins the last strip); returns tiles per image, 0 if the map cannot be cut
static int make_strips(int H, int W, int ks, int plane, int bm, Strips* sp) {
  auto fits = [&](int ws) {
    const int rows = (ws - 1 + bm - 1) / ws + 1;               // most rows a run of bm pixels can touch
    return (rows + ks - 1) * (ws + ks - 1) + 2 <= plane;
  };
  int n = 0;
  if (fits(W)) {
    sp->x_off[0] = 0; sp->ws[0] = W; n = 1;
  } else {
    int x = 0;
    while (x < W && n < 8) {
      int ws = W - x >= 32 ? 32 : W - x;
      if (W - x - ws > 0 && W - x - ws < 8) ws = W - x;        // absorb a thin remainder
      if (!fits(ws)) return 0;
      sp->x_off[n] = x; sp->ws[n] = ws; ++n;
      x += ws;
    }
    if (x < W) return 0;
  }
  sp->n = n;
  int t = 0;
  for (int i = 0; i < n; ++i) { sp->tile0[i] = t; t += (H * sp->ws[i] + bm - 1) / bm; }
  sp->tile0[n] = t;
  for (int i = n; i < 8; ++i) { sp->x_off[i] = 0; sp->ws[i] = 1; }
  return t;
}

template <class C>
static hipError_t launch(const ConvArgs& a, hipStream_t st) {
  int tiles_x = (a.W + C::TW - 1) / C::TW, tiles_y = (a.H + C::TH - 1) / C::TH;
  int mtiles = tiles_x * tiles_y * a.B;
  Strips sp = {};
  if constexpr (C::STRIP) {
    const int tpi = make_strips(a.H, a.W, C::KS, C::PLANE, C::BM, &sp);
    if (tpi == 0) return hipErrorInvalidValue;
    mtiles = tpi * a.B;
  }
  if constexpr (C::FLAT) {
    int R = C::BM / a.W;
    if (R > a.H) R = a.H;
    if (R < 1 || (R + C::KS - 1) * (a.W + C::KS - 1) + 2 > C::PLANE) return hipErrorInvalidValue;
    tiles_y = (a.H + R - 1) / R;
    tiles_x = R;
    mtiles = tiles_y * a.B;
  }
  const int nN = a.CoutP / C::BN;
  int blocks;
  if ((8 % nN) == 0) {
    const int per = 8 / nN;
    blocks = (mtiles + per - 1) / per * 8;
  } else {
    blocks = mtiles * nN;
  }
  static LdsAttr attr;   // per device, not per process: a second Engine on another GPU needs its own call
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_igemm_f32_kernel<C>), C::LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_igemm_f32_kernel<C>, dim3(blocks), dim3(256), C::LDS_BYTES, st, a, tiles_x, tiles_y, mtiles, nN, sp);
  return hipGetLastError();
}

int conv_igemm_bn(int Cout) { return Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32); }

// Patch shape: 4x32 tiles a 60x90 map with 6 % padding; 8x16 suits the smaller / odd maps.
// slots an 8x16-patch tiling / a whole-row (FLAT) tiling spends on an H x W map
static inline int slots_patch(int H, int W) { return ((H + 7) / 8) * ((W + 15) / 16) * 128; }
static inline int slots_flat(int H, int W, int ks) {
  int R = 128 / W;
  if (R > H) R = H;
  if (R < 1 || (R + ks - 1) * (W + ks - 1) + 2 > (8 + ks - 1) * 32 + 2) return 1 << 30;
  return ((H + R - 1) / R) * 128;
}

hipError_t conv_igemm_f32(const ConvArgs& a, int ks, hipStream_t st) {
  const int bn = conv_igemm_bn(a.Cout);
  const bool wide = (a.W >= 64) && (a.H % 4 == 0);
  if (bn == 128) {      // strip tiling whenever it needs fewer 128-pixel tiles than the patch / whole-row choices
    Strips sp;
    const int t_strip = make_strips(a.H, a.W, ks, (4 + ks - 1) * 48 + 2, 128, &sp);
    const int t_patch = wide ? ((a.H + 3) / 4) * ((a.W + 31) / 32) : slots_patch(a.H, a.W) / 128;
    const int t_flat = slots_flat(a.H, a.W, ks) / 128;
    if (t_strip > 0 && t_strip < t_patch && t_strip < t_flat)
      return ks == 9 ? launch<Cfg<9, 4, 32, 128, 2, 2, 1, false, true>>(a, st) : launch<Cfg<5, 4, 32, 128, 2, 2, 1, false, true>>(a, st);
  }
  if (!wide && bn == 128 && slots_flat(a.H, a.W, ks) < slots_patch(a.H, a.W)) {   // e.g. 15x23: 3 tiles instead of 4
    if (ks == 9) return launch<Cfg<9, 8, 16, 128, 2, 2, 1, true>>(a, st);
    if (ks == 5) return launch<Cfg<5, 8, 16, 128, 2, 2, 1, true>>(a, st);
  }
  if (ks == 9) {
    if (bn == 128) return wide ? launch<Cfg<9, 4, 32, 128, 2, 2, 1>>(a, st) : launch<Cfg<9, 8, 16, 128, 2, 2, 1>>(a, st);
    if (bn == 64) return wide ? launch<Cfg<9, 4, 32, 64, 2, 2, 3>>(a, st) : launch<Cfg<9, 8, 16, 64, 2, 2, 3>>(a, st);
    return wide ? launch<Cfg<9, 4, 32, 32, 4, 1, 9>>(a, st) : launch<Cfg<9, 8, 16, 32, 4, 1, 9>>(a, st);
  }
  if (ks == 5) {
    if (bn == 128) return wide ? launch<Cfg<5, 4, 32, 128, 2, 2, 1>>(a, st) : launch<Cfg<5, 8, 16, 128, 2, 2, 1>>(a, st);
    if (bn == 64) return wide ? launch<Cfg<5, 4, 32, 64, 2, 2, 5>>(a, st) : launch<Cfg<5, 8, 16, 64, 2, 2, 5>>(a, st);
    return wide ? launch<Cfg<5, 4, 32, 32, 4, 1, 5>>(a, st) : launch<Cfg<5, 8, 16, 32, 4, 1, 5>>(a, st);
  }
  return hipErrorInvalidValue;
}

// HWIO [k,k,Cin,Cout] -> [tap][Cin/4][CoutP][4], zero-padded channels.
__global__ void pack_weights_f32_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int Cin, int Cout, int CoutP) {
  const size_t n = (size_t)taps * Cin * CoutP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int k4 = i & 3;
    size_t r = i >> 2;
    const int co = r % CoutP; r /= CoutP;
    const int c4 = r % (Cin >> 2);
    const int tap = r / (Cin >> 2);
    const int ci = c4 * 4 + k4;
    wp[i] = co < Cout ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
  }
}

hipError_t pack_weights_f32(const float* w_hwio, float* wp, int ks, int Cin, int Cout, int CoutP, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_f32_kernel, dim3(2048), dim3(256), 0, st, w_hwio, wp, ks * ks, Cin, Cout, CoutP);
  return hipGetLastError();
}

}  // namespace jcm
