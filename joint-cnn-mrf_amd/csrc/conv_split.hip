// fp32 convolution on the 16-bit matrix cores with split operands (the direct A/B arm of the frequency-domain route, "f32_conv" = 2).
//
// A 16-bit x 16-bit product is exact in the MFMA's fp32 accumulator, so an fp32 operand can be fed as a sum of 16-bit parts and the
// products of the parts accumulated: the error class of an fp32 FMA chain.  (Rounds 1-4 also carried a three-bf16-part / six-product
// form, "bf16x6"; it was dominated by the form below on every axis and retired in round 5.  The template keeps NS general.)
//
// fp16x3 variant (NS = 2): an fp32 value is a0 + a1 with two fp16 parts to 22 bits (a1 may be an fp16 subnormal: absolute
// error < 3e-8, harmless for O(1) activations; weights are stored times 2^12 so theirs is < 1e-11), fp16 x fp16 products are
// exact in fp32, and a0*b1 + a0*b0 + a1*b0 -- three v_mfma_f32_32x32x16_f16 -- drops only a1*b1 (2^-22): on the same
// K = 41472 dot product 2.6e-7 of max|result| with fp32 accumulation, i.e. the accumulate rounding, not the split, is what
// is left.  5.3x the fp32 MFMA peak.  Range: callers pass power-of-two scales (ConvArgs::in_scale / w_scale, pow2_scale_of)
// that lift each operand tensor's largest magnitude into [2^13, 2^14); the epilogue undoes them exactly.
//
// Dataflow = conv_igemm_bf16.hip's big-tile kernel (12x32 pixels x 256 channels per workgroup, 8
// waves of 3x4 fragments, rotating-B schedule, weights by LDS-DMA) with the K axis extended by the
// (p,q) sub-steps: activations stay fp32 in HBM and are split once per 16-channel chunk while the
// halo is written to LDS (one plane per part); weights are split at pack time.
#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// KS x KS taps, TH x TW pixel patch (or, FLAT, TH*TW pixel slots filled with whole rows of a narrow map), BN channels
// NS_ = operand parts: 3 = bf16 parts, six products; 2 = fp16 parts, three products (header comment).  TPS_ = taps per
// weight stage (the taps of a stage lie side by side in one kernel row).
template <int KS_, int BN_, bool FLAT_, int NS_ = 3, int TPS_ = 1>
struct CfgS {
  static constexpr int KS = KS_, TH = 12, TW = 32, BN = BN_, WM = 4, WN = 2;
  static constexpr bool FLAT = FLAT_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int U = 2;                       // 16-B units per chunk = 16 channels = one k16 step
  static constexpr int NSPLIT = NS_, NSUB = NS_ == 3 ? 6 : 3, TPS = TPS_;
  static constexpr int NSTEP = TPS * NSUB;          // MFMA groups between two barriers
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int HH = TH + KS - 1, WH = TW + KS - 1, WHP = WH;
  // FLAT: (R + KS-1) x (W + KS-1) slots with R = floor(384 / W) rows: 16x53 (30x45 maps) / 23x31 (15x23) for KS = 9
  static constexpr int PLANE = FLAT ? 850 : HH * WHP + 2;
  static constexpr int BM = TH * TW, MR = BM / WM / 32, NR = BN / WN / 32;
  static constexpr int HALO_F4 = NSPLIT * U * PLANE;            // [split][unit][slot]
  static constexpr int WSTAGE_F4 = TPS * NSPLIT * U * BN;       // one stage: [tap][split][unit][co]
  static constexpr int NSTAGE = KS * KS / TPS;
  static constexpr int LDS_BYTES = (HALO_F4 + 2 * WSTAGE_F4) * 16;
  static_assert(KS % TPS == 0, "the taps of a stage share a kernel row");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// sub-step -> (activation part, weight part).  Weight parts are visited in runs (bf16x6: 0 0 0 1 1 2, fp16x3: 1 0 0) so
// that a B fragment is re-read only when its part changes, and the last two steps of bf16x6 share a0: 27 instead of 42
// fragment reads per 72 MFMAs.  (All products land in the same accumulator, whose magnitude is the running sum: their
// order is immaterial.)
template <int NS> __host__ __device__ constexpr int sub_b(int s) { return NS == 3 ? (s < 3 ? 0 : (s < 5 ? 1 : 2)) : (s == 0 ? 1 : 0); }
template <int NS> __host__ __device__ constexpr int sub_a(int s) { return NS == 3 ? (s < 3 ? 2 - s : (s < 5 ? 4 - s : 0)) : (s == 2 ? 1 : 0); }
// step of a stage -> (tap of the stage, sub-step)
template <class C> __host__ __device__ constexpr int key_a(int st) { return (st / C::NSUB) * 8 + sub_a<C::NSPLIT>(st % C::NSUB); }
template <class C> __host__ __device__ constexpr int key_b(int st) { return (st / C::NSUB) * 8 + sub_b<C::NSPLIT>(st % C::NSUB); }
template <class C> __host__ __device__ constexpr bool reload_b(int st) { return st == 0 || (st < C::NSTEP && key_b<C>(st) != key_b<C>(st - 1)); }
template <class C> __host__ __device__ constexpr bool load_a(int st) { return st == 0 || (st < C::NSTEP && key_a<C>(st) != key_a<C>(st - 1)); }
template <class C> __host__ __device__ constexpr int a_buf(int st) {          // which of the two A register sets step st reads
  int n = 0;
  for (int k = 1; k <= st; ++k) n += load_a<C>(k) ? 1 : 0;
  return n & 1;
}

template <class C, int STEP>
__device__ __forceinline__ void a_load(f32x4 (&fa)[C::MR], const unsigned (&aaddr)[C::MR]) {
  constexpr int aoff = (sub_a<C::NSPLIT>(STEP % C::NSUB) * C::U * C::PLANE + STEP / C::NSUB) * 16;     // part plane + tap shift
  static_assert(aoff < 65536, "ds_read offset field is 16 bits");
#pragma unroll
  for (int f = 0; f < C::MR; ++f)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(aoff) : "memory");
}
template <class C, int STEP>
__device__ __forceinline__ void b_load(f32x4& fb, unsigned baddr) {
  constexpr int boff = ((STEP / C::NSUB) * C::NSPLIT + sub_b<C::NSPLIT>(STEP % C::NSUB)) * C::U * C::BN * 16;
  static_assert(boff < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb) : "v"(baddr), "i"(boff) : "memory");
}
// rotating-B schedule (see conv_igemm_bf16.hip).  LDS queue in issue order: A(0) B0..B3(0) | A(1) [B0..B3(1)] | A(2) ...
// with the bracketed reloads present only where the weight part changes.  When the MFMAs of (step k, group g)
// issue they need A(k) and the latest B[g]; younger reads still allowed in flight:
//   the rest of this step's B reloads (NR-1-g, if B was reloaded for k) + A(k+1) (MR) + next step's reloads so far (g)
template <class C, int STEP, int G>
__device__ __forceinline__ void rot_g(f32x4 (&fa)[2][C::MR], f32x4 (&fb)[C::NR], const unsigned (&baddr)[C::NR], f32x16 (&acc)[C::MR][C::NR]) {
  if constexpr (G < C::NR) {
    constexpr bool more = STEP + 1 < C::NSTEP;
    constexpr int cur = a_buf<C>(STEP);
    constexpr int inflight = (reload_b<C>(STEP) ? C::NR - 1 - G : 0) + ((more && load_a<C>(STEP + 1)) ? C::MR : 0) +
                             ((more && reload_b<C>(STEP + 1)) ? G : 0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(inflight) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < C::MR; ++f) {
      if constexpr (C::NSPLIT == 2)
        acc[f][G] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[cur][f]), __builtin_bit_cast(f16x8, fb[G]), acc[f][G], 0, 0, 0);
      else
        acc[f][G] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][f]), __builtin_bit_cast(bf16x8, fb[G]), acc[f][G], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (more && reload_b<C>(STEP + 1)) b_load<C, STEP + 1>(fb[G], baddr[G]);
    rot_g<C, STEP, G + 1>(fa, fb, baddr, acc);
  }
}
template <class C, int STEP>
__device__ __forceinline__ void stage_steps(f32x4 (&fa)[2][C::MR], f32x4 (&fb)[C::NR], const unsigned (&aaddr)[C::MR],
                                            const unsigned (&baddr)[C::NR], f32x16 (&acc)[C::MR][C::NR]) {
  if constexpr (STEP < C::NSTEP) {
    if constexpr (STEP + 1 < C::NSTEP && load_a<C>(STEP + 1)) a_load<C, STEP + 1>(fa[a_buf<C>(STEP + 1)], aaddr);
    rot_g<C, STEP, 0>(fa, fb, baddr, acc);
    stage_steps<C, STEP + 1>(fa, fb, aaddr, baddr, acc);
  }
}

// two fp16 parts: a = a0 + a1 to 22 bits (fp16 subnormals keep the absolute error of a1 below 3e-8)
__device__ __forceinline__ void split8h(const f32x4& lo, const f32x4& hi, f16x8& p0, f16x8& p1) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = i < 4 ? lo[i] : hi[i - 4];
    const _Float16 h0 = static_cast<_Float16>(v);
    p0[i] = h0;
    p1[i] = static_cast<_Float16>(v - static_cast<float>(h0));
  }
}

constexpr float kW16Scale = 4096.0f;     // fp16 weight parts are stored times 2^12 (|w| < 16), undone in the epilogue

template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv_split_kernel(ConvArgs a, int tiles_x, int tiles_y, int mtiles, int nN) {
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
  // FLAT tiles: `tiles_x` carries R (rows per tile), there is one tile across
  const int flat_r = C::FLAT ? tiles_x : 0;
  const int tx = C::FLAT ? 0 : mt % tiles_x;
  const int ty = C::FLAT ? mt % tiles_y : (mt / tiles_x) % tiles_y;
  const int b = C::FLAT ? mt / tiles_y : mt / (tiles_x * tiles_y);
  const int y0 = ty * (C::FLAT ? flat_r : C::TH), x0 = tx * C::TW, n0 = nt * C::BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid % C::WN;
  const int h = lane >> 5, l31 = lane & 31;

  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CoutP = a.CoutP;
  const float* __restrict__ xb = static_cast<const float*>(a.x) + (size_t)b * H * W * Cin;
  const __bf16* __restrict__ wp = static_cast<const __bf16*>(a.wp);

  // halo geometry: compile-time for patches, per-launch for FLAT tiles
  const int whp = C::FLAT ? W + C::KS - 1 : C::WHP;
  const int wh = C::FLAT ? W + C::KS - 1 : C::WH;
  const int hh = C::FLAT ? flat_r + C::KS - 1 : C::HH;
  int aslot[C::MR], bcol[C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f) {
    const int r = (wm * C::MR + f) * 32 + l31;
    if constexpr (C::FLAT) {
      const int rr = r < flat_r * W ? r : 0;          // padding slots compute on pixel 0 and are dropped
      const int yy = rr / W;
      aslot[f] = yy * whp + (rr - yy * W);
    } else {
      aslot[f] = (r / C::TW) * C::WHP + (r % C::TW);
    }
  }
#pragma unroll
  for (int g = 0; g < C::NR; ++g) bcol[g] = (wn * C::NR + g) * 32 + l31;

  // half-wave h reads unit h of the chunk (channels 8h..8h+7): MFMA operand k = 8*(lane>>5) + i
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned abase[C::MR], bbase[C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f) abase[f] = lds0 + (unsigned)(h * C::PLANE + aslot[f]) * 16u;
#pragma unroll
  for (int g = 0; g < C::NR; ++g) bbase[g] = lds0 + (unsigned)(h * C::BN + bcol[g]) * 16u;

  f32x16 acc[C::MR][C::NR];
#pragma unroll
  for (int f = 0; f < C::MR; ++f)
#pragma unroll
    for (int g = 0; g < C::NR; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  const int cin8 = Cin >> 3;
  // one stage = one tap of one 16-channel chunk: 3 parts x 2 units x BN channels x 16 B (24 KB at BN = 256) in 1-KB LDS-DMA pieces
  constexpr int NWAVE = C::NT / 64;
  constexpr int NPIECE = C::WSTAGE_F4 / 64;
  constexpr int QPU = C::BN / 64;
  static_assert(C::BN % 64 == 0, "a (part, unit) row is a whole number of pieces");
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wp), 0,
                                                       (int)((size_t)C::KS * C::KS * Cin * CoutP * 2 * C::NSPLIT), 0x00020000);   // 2-byte parts either way
  const unsigned wvoff = (unsigned)(n0 + lane) * 16u;
  auto wdma = [&](int g, int bufsel) {
    const int chunk = g / C::NSTAGE, st = g - chunk * C::NSTAGE;
#pragma unroll
    for (int i = 0; i < (NPIECE + NWAVE - 1) / NWAVE; ++i) {
      const int piece = wid + i * NWAVE;              // wave-uniform; LDS image [part][unit][co]
      if ((NPIECE % NWAVE != 0) && piece >= NPIECE) break;
      f32x4* dst = wbuf + bufsel * C::WSTAGE_F4 + piece * 64;
      const int q = piece % QPU, pu = piece / QPU;
      const int u = pu % C::U, pt = pu / C::U;
      const int part = pt % C::NSPLIT, tap = st * C::TPS + pt / C::NSPLIT;
      const unsigned soff = (unsigned)(((((tap * cin8 + chunk * C::U + u) * C::NSPLIT) + part) * CoutP + q * 64) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff + soff, 0, 0, 0);
    }
  };

  const int nchunk = Cin >> 4;
  const int nstage_total = nchunk * C::NSTAGE;
  int buf = 0;
  wdma(0, 0);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();   // every wave is done reading the previous chunk's halo
    for (int idx = tid; idx < C::U * hh * wh; idx += C::NT) {
      const int u = idx & (C::U - 1);
      const int pix = idx >> 1;
      const int hy = pix / wh, hx = pix - hy * wh;
      const int gy = y0 - C::PAD + hy, gx = x0 - C::PAD + hx;
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        const float* src = xb + ((size_t)gy * W + gx) * Cin + chunk * 16 + u * 8;
        lo = *reinterpret_cast<const f32x4*>(src);
        hi = *reinterpret_cast<const f32x4*>(src + 4);
      }
      const int slot = hy * whp + hx;
      static_assert(C::NSPLIT == 2, "two fp16 parts per operand");
      if (a.in_scale) {                 // gradients: lift into the fp16 range by the tensor's power-of-two scale
        const float S = a.in_scale[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] *= S; hi[i] *= S; }
      }
      f16x8 p0, p1;
      split8h(lo, hi, p0, p1);
      halo[(0 * C::U + u) * C::PLANE + slot] = __builtin_bit_cast(f32x4, p0);
      halo[(1 * C::U + u) * C::PLANE + slot] = __builtin_bit_cast(f32x4, p1);
    }
    __syncthreads();   // halo visible before any wave's (pre-barrier) step-0 A reads
    for (int s = 0; s < C::NSTAGE; ++s) {
      const int g = chunk * C::NSTAGE + s;
      const int tap0 = s * C::TPS;
      const int ky = tap0 / C::KS, kx = tap0 - ky * C::KS;          // TPS divides KS: one kernel row per stage
      const unsigned tbytes = (unsigned)(ky * whp + kx) * 16u;
      const unsigned wbytes = (unsigned)(C::HALO_F4 + buf * C::WSTAGE_F4) * 16u;
      unsigned aaddr[C::MR], baddr[C::NR];
#pragma unroll
      for (int f = 0; f < C::MR; ++f) aaddr[f] = abase[f] + tbytes;
#pragma unroll
      for (int gq = 0; gq < C::NR; ++gq) baddr[gq] = bbase[gq] + wbytes;
      f32x4 fa[2][C::MR], fb[C::NR];
      a_load<C, 0>(fa[0], aaddr);                              // from the halo: stable for the whole chunk
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage g have landed
      __builtin_amdgcn_s_barrier();                         // ... and everyone's; the other buffer (stage g-1) is free
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gq = 0; gq < C::NR; ++gq) b_load<C, 0>(fb[gq], baddr[gq]);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < nstage_total) wdma(g + 1, buf ^ 1);       // in flight behind this stage's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      stage_steps<C, 0>(fa, fb, aaddr, baddr, acc);
      buf ^= 1;
    }
  }

  // epilogue: bias (+ ReLU + folded BatchNorm) -> fp32 NHWC
  float* __restrict__ ob = static_cast<float*>(a.out) + (size_t)b * H * W * Cout;
  const float unscale = (C::NSPLIT == 2 ? (a.w_scale ? a.w_scale[1] : 1.0f / kW16Scale) : 1.0f) * (a.in_scale ? a.in_scale[1] : 1.0f);
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
        if constexpr (C::FLAT) {
          const int yy = r / W;
          y = y0 + yy; x = r - yy * W;
          ok = r < flat_r * W && y < H;
        } else {
          y = y0 + r / C::TW; x = x0 + r % C::TW;
          ok = y < H && x < W;
        }
        if (ok) {
          float v = (C::NSPLIT == 2 ? acc[f][g][i] * unscale : acc[f][g][i]) + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc + sh;
          ob[((size_t)y * W + x) * Cout + co] = v;
        }
      }
    }
  }
}

// fp32 HWIO [k,k,Cin,Cout] -> bf16 [tap][Cin/8][part][CoutP][8], part = 0 (high) .. 2 (low); zero-padded channels
__global__ void pack_weights_split16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int taps, int Cin, int Cout, int CoutP,
                                            const float* __restrict__ w_scale) {
  const float sw = w_scale ? w_scale[0] : kW16Scale;
  const size_t n = (size_t)taps * Cin * CoutP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int k8 = i & 7;
    size_t r = i >> 3;
    const int co = r % CoutP; r /= CoutP;
    const int c8 = r % (Cin >> 3);
    const int tap = r / (Cin >> 3);
    const float v = (co < Cout ? w[((size_t)tap * Cin + c8 * 8 + k8) * Cout + co] : 0.f) * sw;
    const _Float16 h0 = static_cast<_Float16>(v);
    const size_t base = (((size_t)tap * (Cin >> 3) + c8) * 2) * CoutP * 8 + (size_t)co * 8 + k8;
    wp[base] = h0;
    wp[base + (size_t)CoutP * 8] = static_cast<_Float16>(v - static_cast<float>(h0));
  }
}

}  // namespace

namespace {
template <class C>
hipError_t launch_s(const ConvArgs& a, hipStream_t st) {
  int tiles_x = (a.W + C::TW - 1) / C::TW, tiles_y = (a.H + C::TH - 1) / C::TH;
  int mtiles = tiles_x * tiles_y * a.B;
  if constexpr (C::FLAT) {
    int R = C::BM / a.W;
    if (R > a.H) R = a.H;
    if (R < 1 || (R + C::KS - 1) * (a.W + C::KS - 1) + 2 > C::PLANE) return hipErrorInvalidValue;
    tiles_y = (a.H + R - 1) / R;
    tiles_x = R;                      // the kernel reads R from this slot
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
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_split_kernel<C>), C::LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_split_kernel<C>, dim3(blocks), dim3(C::NT), C::LDS_BYTES, st, a, tiles_x, tiles_y, mtiles, nN);
  return hipGetLastError();
}
bool flat_fits(int ks, int H, int W) {
  int R = 384 / W;
  if (R > H) R = H;
  return W <= 48 && R >= 1 && (R + ks - 1) * (W + ks - 1) + 2 <= 850;
}
}  // namespace

// channel tile the kernel will use for a layer (its packed CoutP must be a multiple of it)
int conv_split_bn(int Cout) { return Cout % 256 == 0 ? 256 : 128; }

namespace {
// pixel tiles of one launch (12x32 patches, or whole-row tiles on the narrow maps)
int split_tiles(int B, int H, int W) {
  if (W >= 64 && H % 12 == 0) return B * (H / 12) * ((W + 31) / 32);
  int R = 384 / W;
  if (R > H) R = H;
  return B * ((H + R - 1) / R);
}
// 256-channel tiles unless that leaves most of the 256 CUs without a workgroup (small maps at small batch)
// ... and 64-channel tiles when even 128-channel ones leave more than half the chip idle (15x23 maps at 16 images)
int split_bn(int CoutP, int tiles) {
  if (CoutP % 256 == 0 && tiles * (CoutP / 256) >= 256) return 256;
  if (tiles * (CoutP / 128) >= 128) return 128;
  return 64;
}
}  // namespace

// min_wgs: smallest grid worth a launch (default 128: below half a chip of 384-pixel tiles the 128-pixel exact kernel wins; tests
// pass 0 to force the split kernels on small batches)
bool conv_split_supported(int ks, int Cin, int CoutP, int B, int H, int W, int min_wgs) {
  if (!(ks == 9 || ks == 5) || Cin % 16 || CoutP % 128) return false;
  if (!((W >= 64 && H % 12 == 0) || flat_fits(ks, H, W))) return false;
  const int tiles = split_tiles(B, H, W);
  return tiles * (CoutP / split_bn(CoutP, tiles)) >= min_wgs;
}

// ns = operand parts: 2 (fp16 parts, three products)
size_t conv_split_weight_bytes(int ks, int Cin, int CoutP, int ns) { return (size_t)ks * ks * Cin * CoutP * 2 * ns; }

hipError_t pack_weights_split(const float* w_hwio, void* wp, int ks, int Cin, int Cout, int CoutP, int ns, hipStream_t st, const float* w_scale) {
  if (ns != 2) return hipErrorInvalidValue;      // (ns = 3, three bf16 parts / six products, was retired in round 5)
  hipLaunchKernelGGL(pack_weights_split16_kernel, dim3(2048), dim3(256), 0, st, w_hwio, static_cast<_Float16*>(wp), ks * ks, Cin, Cout, CoutP, w_scale);
  return hipGetLastError();
}

namespace {
template <int NS, int TPS9>
hipError_t dispatch_s(const ConvArgs& a, int ks, hipStream_t st) {
  const bool wide = a.W >= 64 && a.H % 12 == 0;
  const int bn = split_bn(a.CoutP, split_tiles(a.B, a.H, a.W));
  if (ks == 9) {
    if (bn == 256) return wide ? launch_s<CfgS<9, 256, false, NS, TPS9>>(a, st) : launch_s<CfgS<9, 256, true, NS, TPS9>>(a, st);
    if (bn == 128) return wide ? launch_s<CfgS<9, 128, false, NS, TPS9>>(a, st) : launch_s<CfgS<9, 128, true, NS, TPS9>>(a, st);
    return wide ? launch_s<CfgS<9, 64, false, NS, TPS9>>(a, st) : launch_s<CfgS<9, 64, true, NS, TPS9>>(a, st);
  }
  if (bn == 256) return wide ? launch_s<CfgS<5, 256, false, NS, 1>>(a, st) : launch_s<CfgS<5, 256, true, NS, 1>>(a, st);
  if (bn == 128) return wide ? launch_s<CfgS<5, 128, false, NS, 1>>(a, st) : launch_s<CfgS<5, 128, true, NS, 1>>(a, st);
  return wide ? launch_s<CfgS<5, 64, false, NS, 1>>(a, st) : launch_s<CfgS<5, 64, true, NS, 1>>(a, st);
}
}  // namespace

// a.x fp32 NHWC, a.wp from pack_weights_split with the same ns (CoutP a multiple of 128), a.out fp32
hipError_t conv_split_f32(const ConvArgs& a, int ks, int ns, hipStream_t st) {
  if (!conv_split_supported(ks, a.Cin, a.CoutP, a.B, a.H, a.W, 0)) return hipErrorInvalidValue;
  if (ns != 2) return hipErrorInvalidValue;
  // fp16x3: three taps of weights per stage (108 MFMAs between barriers)
  return dispatch_s<2, 3>(a, ks, st);
}

}  // namespace jcm
