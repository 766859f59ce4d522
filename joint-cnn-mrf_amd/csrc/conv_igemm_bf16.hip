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

template <int KS_, int TH_, int TW_, int BN_, int WM_, int WN_, int TPS_, bool ROT_ = false, bool FLAT_ = false>
struct CfgB {
  static constexpr int KS = KS_, TH = TH_, TW = TW_, BN = BN_, WM = WM_, WN = WN_, TPS = TPS_;
  static constexpr bool ROT = ROT_;                 // rotating-B fragment schedule (big tiles, see stage_steps_rot)
  // FLAT: the M tile is R whole rows of a NARROW map (30x45, 15x23) flattened to TH*TW pixel slots
  // (R = floor(TH*TW / W), chosen at launch) instead of a TH x TW patch: a 12x16 patch wastes
  // 28 % / 120 % of its slots on those maps, whole rows waste 14 % / 11 %.
  static constexpr bool FLAT = FLAT_;
  static constexpr int NT = WM * WN * 64;           // threads
  static constexpr int U = 4;                       // 16-B units per chunk = 32 bf16 channels
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int HH = TH + KS - 1;
  static constexpr int WH = TW + KS - 1;
  // halo row pitch in 16-B slots: a 32-pixel fragment lies in one row (any pitch is conflict-free);
  // a 16-pixel-wide patch puts 2 rows in a fragment and needs the pitch to be a multiple of 16.
  static constexpr int WHP = TW == 32 ? WH : (WH + 15) / 16 * 16;
  static constexpr int PLANE = FLAT ? 850 : HH * WHP + 2;   // FLAT: (R+k-1) x (W+k-1) <= 16x53 / 23x31 slots
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

// Issue the MR + NR ds_read_b128 of k16-step STEP of the current stage (immediate offsets).
template <class C, int STEP>
__device__ __forceinline__ void frag_load(f32x4 (&fa)[C::MR], f32x4 (&fb)[C::NR], const unsigned (&aaddr)[C::MR],
                                          const unsigned (&baddr)[C::NR]) {
  constexpr int tp = STEP / (C::U / 2), st = STEP % (C::U / 2);
  constexpr int aoff = (2 * st * C::PLANE + tp) * 16;
  constexpr int boff = ((tp * C::U + 2 * st) * C::BN) * 16;
  static_assert(aoff < 65536 && boff < 65536, "ds_read offset field is 16 bits");
#pragma unroll
  for (int f = 0; f < C::MR; ++f)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(aoff) : "memory");
#pragma unroll
  for (int g = 0; g < C::NR; ++g)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[g]) : "v"(baddr[g]), "i"(boff) : "memory");
}

// Steps STEP.. of a stage: [reads of STEP+1] -> wait until STEP's own reads landed (the MR+NR newer
// ones stay in flight) -> MR*NR MFMAs.
template <class C, int STEP>
__device__ __forceinline__ void stage_steps(f32x4 (&fa)[2][C::MR], f32x4 (&fb)[2][C::NR], const unsigned (&aaddr)[C::MR],
                                            const unsigned (&baddr)[C::NR], f32x16 (&acc)[C::MR][C::NR]) {
  constexpr int NSTEP = C::TPS * (C::U / 2);
  if constexpr (STEP < NSTEP) {
    constexpr int cur = STEP & 1;
    if constexpr (STEP + 1 < NSTEP) {
      frag_load<C, STEP + 1>(fa[cur ^ 1], fb[cur ^ 1], aaddr, baddr);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(C::MR + C::NR) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < C::MR; ++f)
#pragma unroll
      for (int g = 0; g < C::NR; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][f]),
                                                            __builtin_bit_cast(bf16x8, fb[cur][g]), acc[f][g], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage_steps<C, STEP + 1>(fa, fb, aaddr, baddr, acc);
  }
}

// ---- rotating-B schedule for big tiles (MR x NR = 3 x 4 fragments per wave) -----------------------
// A fragments are double buffered by step parity; the NR B fragments live in ONE register set:
// as soon as the MR MFMAs that use B[g] of step k have issued, B[g] of step k+1 is read into
// it.  In issue order the LDS queue is  An(k)[MR] B0(k) .. B{NR-1}(k) An(k+1)[MR] B0(k+1) ...,
// so when the MFMAs of (k, g) need B[g](k) exactly MR+NR-1 younger reads are in flight --
// lgkmcnt(MR+NR-1) at every (k, g), and NR-1-g in the last step of a stage.
template <class C, int STEP>
__device__ __forceinline__ void a_load(f32x4 (&fa)[C::MR], const unsigned (&aaddr)[C::MR]) {
  constexpr int tp = STEP / (C::U / 2), st = STEP % (C::U / 2);
  constexpr int aoff = (2 * st * C::PLANE + tp) * 16;
  static_assert(aoff < 65536, "ds_read offset field is 16 bits");
#pragma unroll
  for (int f = 0; f < C::MR; ++f)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(aoff) : "memory");
}
template <class C, int STEP>
__device__ __forceinline__ void b_load(f32x4& fb, unsigned baddr) {
  constexpr int tp = STEP / (C::U / 2), st = STEP % (C::U / 2);
  constexpr int boff = ((tp * C::U + 2 * st) * C::BN) * 16;
  static_assert(boff < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb) : "v"(baddr), "i"(boff) : "memory");
}
template <class C, int STEP, int G>
__device__ __forceinline__ void rot_g(f32x4 (&fa)[2][C::MR], f32x4 (&fb)[C::NR], const unsigned (&baddr)[C::NR],
                                      f32x16 (&acc)[C::MR][C::NR]) {
  constexpr int NSTEP = C::TPS * (C::U / 2);
  if constexpr (G < C::NR) {
    constexpr bool more = STEP + 1 < NSTEP;
    constexpr int cur = STEP & 1;
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(more ? C::MR + C::NR - 1 : C::NR - 1 - G) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < C::MR; ++f)
      acc[f][G] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][f]),
                                                          __builtin_bit_cast(bf16x8, fb[G]), acc[f][G], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (more) b_load<C, STEP + 1>(fb[G], baddr[G]);
    rot_g<C, STEP, G + 1>(fa, fb, baddr, acc);
  }
}
template <class C, int STEP>
__device__ __forceinline__ void stage_steps_rot(f32x4 (&fa)[2][C::MR], f32x4 (&fb)[C::NR], const unsigned (&aaddr)[C::MR],
                                                const unsigned (&baddr)[C::NR], f32x16 (&acc)[C::MR][C::NR]) {
  constexpr int NSTEP = C::TPS * (C::U / 2);
  if constexpr (STEP < NSTEP) {
    if constexpr (STEP + 1 < NSTEP) a_load<C, STEP + 1>(fa[(STEP & 1) ^ 1], aaddr);
    rot_g<C, STEP, 0>(fa, fb, baddr, acc);
    stage_steps_rot<C, STEP + 1>(fa, fb, aaddr, baddr, acc);
  }
}

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
  const __bf16* __restrict__ xb = static_cast<const __bf16*>(a.x) + (size_t)b * H * W * Cin;
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

  // per-lane LDS byte addresses of this lane's fragment rows (half-wave h reads unit 2*st+h)
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

  // Weight stage -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write
  // pass).  A stage is NPIECE 1-KB pieces (64 output channels x 16 B of one (tap, unit)); the
  // LDS image is lane-linear, which is exactly the [unit][co][8 ch] layout the fragments read.
  constexpr int NWAVE = C::NT / 64;
  constexpr int NPIECE = C::WSTAGE_F4 / 64;
  static_assert(C::WSTAGE_F4 % 64 == 0, "a stage is a whole number of 1-KB LDS-DMA pieces");
  // Source addressing through a buffer descriptor: the per-lane part (output channel) is ONE VGPR
  // for the whole kernel, the stage-dependent part (tap, chunk, unit) is a scalar offset and the
  // 64-channel quarter an immediate -- no per-stage vector address arithmetic, nothing to spill.
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wp), 0,
                                                       (int)((size_t)C::KS * C::KS * Cin * CoutP * 2), 0x00020000);
  const unsigned wvoff = (unsigned)(n0 + lane) * 16u;
  auto wdma = [&](int g, int bufsel) {
    const int chunk = g / C::NSTAGE, s = g - chunk * C::NSTAGE;
#pragma unroll
    for (int i = 0; i < (NPIECE + NWAVE - 1) / NWAVE; ++i) {
      const int piece = wid + i * NWAVE;              // wave-uniform
      if ((NPIECE % NWAVE == 0) || piece < NPIECE) {
        f32x4* dst = wbuf + bufsel * C::WSTAGE_F4 + piece * 64;
        if constexpr (C::BN % 64 == 0) {
          constexpr int QPU = C::BN / 64;             // 1-KB pieces per (tap, unit) row
          const int q = piece % QPU, tu = piece / QPU;
          const int u = tu % C::U, tap = s * C::TPS + tu / C::U;
          const unsigned soff = (unsigned)(((tap * cin8 + chunk * C::U + u) * CoutP + q * 64) * 16);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, wvoff + soff, 0, 0, 0);
        } else {
          const int idx = piece * 64 + lane;          // f32x4 index inside the stage image
          const int co = idx % C::BN;
          const int tu = idx / C::BN;
          const int u = tu % C::U, tap = s * C::TPS + tu / C::U;
          const __bf16* src = wp + (((size_t)tap * cin8 + chunk * C::U + u) * CoutP + n0 + co) * 8;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    }
  };

  const int nchunk = Cin >> 5;
  const int nstage_total = nchunk * C::NSTAGE;
  int buf = 0;
  wdma(0, 0);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();   // every wave is done reading the previous chunk's halo
    for (int idx = tid; idx < C::U * hh * wh; idx += C::NT) {
      const int u = idx & (C::U - 1);
      const int pix = idx >> 2;
      const int hy = pix / wh, hx = pix - hy * wh;
      const int gy = y0 - C::PAD + hy, gx = x0 - C::PAD + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(xb + ((size_t)gy * W + gx) * Cin + chunk * 32 + u * 8);
      halo[u * C::PLANE + hy * whp + hx] = v;
    }
    __syncthreads();   // halo visible before any wave's (pre-barrier) step-0 A reads
    for (int s = 0; s < C::NSTAGE; ++s) {
      const int g = chunk * C::NSTAGE + s;
      // ---- one stage = TPS taps x 2 k16-steps.  Fragment reads are software pipelined one step
      // ahead with hand-counted waits: hipcc sinks prefetched ds_reads back next to their MFMAs
      // and only ever waits lgkmcnt(0) here, which exposes a full LDS round trip per step, so
      // the reads are inline asm (the compiler then tracks none of them) and every wait is ours.
      const int tap0 = s * C::TPS;
      const int ky = tap0 / C::KS, kx0 = tap0 - ky * C::KS;      // TPS divides KS: one kernel row per stage
      const unsigned tbytes = (unsigned)(ky * whp + kx0) * 16u;
      const unsigned wbytes = (unsigned)(C::HALO_F4 + buf * C::WSTAGE_F4) * 16u;
      unsigned aaddr[C::MR], baddr[C::NR];
#pragma unroll
      for (int f = 0; f < C::MR; ++f) aaddr[f] = abase[f] + tbytes;
#pragma unroll
      for (int gq = 0; gq < C::NR; ++gq) baddr[gq] = bbase[gq] + wbytes;
      if constexpr (C::ROT) {
        f32x4 fa[2][C::MR], fb[C::NR];
        // Stage boundary, ordered so that nothing waits idle: the A fragments of step 0 come from
        // the halo (stable for the whole chunk) and are requested BEFORE the barrier; the weight
        // DMA of the next stage is issued while the step-0 reads are in flight.
        a_load<C, 0>(fa[0], aaddr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage g have landed
        __builtin_amdgcn_s_barrier();                        // ... and everyone's; buf^1 (stage g-1) is free
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gq = 0; gq < C::NR; ++gq) b_load<C, 0>(fb[gq], baddr[gq]);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < nstage_total) wdma(g + 1, buf ^ 1);      // in flight behind this stage's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        stage_steps_rot<C, 0>(fa, fb, aaddr, baddr, acc);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g + 1 < nstage_total) wdma(g + 1, buf ^ 1);
        f32x4 fa[2][C::MR], fb[2][C::NR];
        frag_load<C, 0>(fa[0], fb[0], aaddr, baddr);
        stage_steps<C, 0>(fa, fb, aaddr, baddr, acc);
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
          float v = acc[f][g][i] + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc + sh;
          const size_t o = a.out_planar ? (((size_t)b * (Cout >> 3) + (co >> 3)) * H * W + (size_t)y * W + x) * 8 + (co & 7)
                                        : (((size_t)b * H + y) * W + x) * Cout + co;
          if (OUT_F32) static_cast<float*>(a.out)[o] = v;
          else static_cast<__bf16*>(a.out)[o] = static_cast<__bf16>(v);
        }
      }
    }
  }
}

template <class C, bool OUT_F32>
static hipError_t launch_b(const ConvArgs& a, hipStream_t st) {
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
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_igemm_bf16_kernel<C, OUT_F32>), C::LDS_BYTES); e != hipSuccess) return e;
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
  if (ks == 9 && bn == 256 && conv_strip_bf16_supported(a, ks)) return conv_strip_bf16(a, st);   // flattened strips: no padded slots
  if (ks == 5 && bn == 128 && conv5_strip_bf16_supported(a, ks)) return conv5_strip_bf16(a, st);  // 768-pixel strips for the 5x5 layers
  if (a.in_planar || (a.out_planar && (out_f32 || a.Cout % 8))) return hipErrorInvalidValue;       // the patch kernels read NHWC
  if (ks == 9) {
    // 60x90 / 120x180 maps: 12x32 patch x 256 channels, rotating-B schedule (half the weight stream
    // per FLOP of the 6x32 tile, 7 instead of 10 fragment reads per 12 MFMAs)
    if (bn == 256 && wide && a.H % 12 == 0) return launch_b<CfgB<9, 12, 32, 256, 4, 2, 3, true>, false>(a, st);
    // narrow maps (30x45, 15x23): whole-row tiles, same rotating-B schedule
    if (bn == 256 && !wide && (384 / a.W + 8) * (a.W + 8) + 2 <= 850 && a.W <= 48)
      return launch_b<CfgB<9, 12, 32, 256, 4, 2, 3, true, true>, false>(a, st);
    if (bn == 256) return wide ? launch_b<CfgB<9, 6, 32, 256, 2, 4, 3>, false>(a, st) : launch_b<CfgB<9, 12, 16, 256, 2, 4, 3>, false>(a, st);
    if (bn == 128) return wide ? launch_b<CfgB<9, 6, 32, 128, 2, 4, 3>, false>(a, st) : launch_b<CfgB<9, 12, 16, 128, 2, 4, 3>, false>(a, st);
    if (bn == 64) return wide ? launch_b<CfgB<9, 6, 32, 64, 2, 2, 9>, false>(a, st) : launch_b<CfgB<9, 12, 16, 64, 2, 2, 9>, false>(a, st);
    return wide ? launch_b<CfgB<9, 6, 32, 32, 2, 1, 9>, false>(a, st) : launch_b<CfgB<9, 12, 16, 32, 2, 1, 9>, false>(a, st);
  }
  if (ks == 5) {
    // K is short here (2-4 chunks): one-tap stages keep LDS at 42 KB so 3 workgroups share a CU and
    // cover each other's halo loads and barriers (measured 10 % faster than 5-tap stages)
    if (bn == 128) return wide ? launch_b<CfgB<5, 6, 32, 128, 2, 4, 1>, false>(a, st) : launch_b<CfgB<5, 12, 16, 128, 2, 4, 1>, false>(a, st);
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
