// 9x9 SAME convolution on bf16 MFMA, M flattened across the whole batch ("strip" tiles).
//
// conv_igemm_bf16.hip tiles every image with 12x32 pixel patches: 60x90 maps need 15 x 384 = 5760 slots for
// 5400 pixels (6.25 % of the MFMA work is padding), and the halo of each 32-channel chunk is reloaded
// synchronously through VGPRs between two barriers.  This kernel removes both:
//
//   * M tile = 384 CONSECUTIVE pixels of the flattened [B*H*W] pixel axis (256 x 5400 = 3600 x 384 exactly):
//     no padded slots; a tile may run over an image boundary.  A fragment row is an arbitrary halo slot (the
//     A operand is read with per-lane LDS addresses), so a strip costs nothing over a patch in the MFMA loop.
//   * halo = every image row the strip touches, +-4 rows, whole rows with a 4-slot zero gap between rows
//     (pitch W+4: the gap is the right pad of one row and the left pad of the next) and 4 zero rows between
//     the two images of a boundary tile.  It lives in LDS as two 16-byte-unit planes of a 16-CHANNEL chunk
//     and is refilled by LDS-DMA in 1-KB pieces (64 consecutive slots; pad / out-of-image slots are
//     out-of-range lanes of a buffer load and arrive as zeros).  A piece of the NEXT chunk is loaded as soon as
//     the kernel rows of the current chunk that read it are done (schedule computed on the host: one piece
//     per plane per stage, issued by waves 0 and 1), so the halo never stops the MFMA stream.
//   * weights: 3 taps x 16 channels x 256 output channels per stage (24.6 KB) in a 4-deep LDS-DMA ring;
//     at the barrier that opens stage g the weights of stage g+1 have ALREADY landed, so the fragments of the
//     next stage's first k-step are requested before the barrier and the barrier has no load behind it.
//
// Stage = 3 taps of one kernel row x one k16 step = 36 MFMAs per wave; 8 waves (4 x 2), 3 x 4 fragments of
// 32x32 per wave, rotating-B fragment schedule as in conv_igemm_bf16.hip.
// Reference semantics: conv2d SAME stride 1 + bias + ReLU + BatchNorm (main.py:133-135,156-169).
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace strip {
constexpr int KS = 9, BM = 384, BN = 256, MR = 3, NR = 4, NT = 512;   // 8 waves as 4 (M) x 2 (N), 3 x 4 fragments each
constexpr int TPS = 3, NSTAGE = 27, NB = 4;          // taps per stage, stages per 16-channel chunk, weight ring depth
constexpr int PLANE = 1728;                            // 16-B slots per halo plane (27 pieces of 64)
constexpr int WST = TPS * 2 * BN;                      // slots per weight stage: [tap][unit][BN]
constexpr int WB0 = 2 * PLANE;                         // first weight slot
constexpr int DUMMY = WB0 + NB * WST;                  // 64 slots that absorb padding DMA (tail stages, unscheduled halo slots)
constexpr int LDS_BYTES = (DUMMY + 64) * 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct Geom {
  int H, W, HW, pitch, rows, npieces, magic;   // magic: s / pitch == (s * magic) >> 22 for s < PLANE
  int Mtotal, mtiles, nN, async_ok;
  unsigned sched[7];                           // 27 bytes: bit 7 valid, bit 6 "next chunk", bits 0-4 piece
};
}  // namespace strip

using namespace strip;

// one k16 step = one tap: A at slot offset `tp`, B at [tp][unit h][BN]
template <int TP>
__device__ __forceinline__ void s_a_load(f32x4 (&fa)[MR], const unsigned (&aaddr)[MR]) {
#pragma unroll
  for (int f = 0; f < MR; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(TP * 16) : "memory");
}
template <int TP>
__device__ __forceinline__ void s_b_load(f32x4& fb, unsigned baddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb) : "v"(baddr), "i"(TP * 2 * BN * 16) : "memory");
}

// MFMAs of (step, column G) and the read of B[G] for the following step.  Queue invariant (see conv_igemm_bf16.hip):
// exactly MR+NR-1 younger ds_reads are in flight when B[G] of the current step is needed.
template <int PAR, int STEP, int G, class After>
__device__ __forceinline__ void s_rot_g(f32x4 (&fa)[2][MR], f32x4 (&fb)[NR], const unsigned (&baddr)[NR], f32x16 (&acc)[MR][NR], After&& after) {
  if constexpr (G < NR) {
    constexpr int cur = (STEP + PAR) & 1;          // a stage has 3 steps: the A double buffer flips parity every stage
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MR + NR - 1) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < MR; ++f)
      acc[f][G] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][f]), __builtin_bit_cast(bf16x8, fb[G]), acc[f][G], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    s_b_load<(STEP + 1) % TPS>(fb[G], baddr[G]);     // STEP == TPS-1: baddr already points at the next stage's buffer
    after(G);
    __builtin_amdgcn_sched_barrier(0);
    s_rot_g<PAR, STEP, G + 1>(fa, fb, baddr, acc, after);
  }
}

__global__ __launch_bounds__(NT, 2) void conv_strip_bf16_kernel(ConvArgs a, Geom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds = reinterpret_cast<f32x4*>(smem);

  const int L = blockIdx.x, nN = gm.nN;
  int mt, nt;
  if ((8 % nN) == 0) {   // an XCD (blocks b, b+8, ...) keeps one channel tile: its L2 streams 1/nN of the weights
    const int xcd = L & 7, q = L >> 3, per = 8 / nN;
    nt = xcd % nN;
    mt = q * per + xcd / nN;
  } else {
    nt = L % nN;
    mt = L / nN;
  }
  if (mt >= gm.mtiles) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int H = gm.H, W = gm.W, HW = gm.HW, pitch = gm.pitch;
  const int Cin = a.Cin, Cout = a.Cout, CoutP = a.CoutP;
  const int n0 = nt * BN;

  // ---- tile geometry: pixels P0 .. P0+383 of the flattened batch
  const int P0 = mt * BM;
  const int b0 = P0 / HW;
  const int y0 = (P0 - b0 * HW) / W;
  const int Plast = min(P0 + BM - 1, gm.Mtotal - 1);
  const bool crossing = Plast / HW != b0;             // the strip runs into the next image
  const bool async_halo = gm.async_ok && !crossing;   // boundary tiles (1 in 15) reload their halo synchronously

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned aaddr[MR], baddr[NR];
#pragma unroll
  for (int f = 0; f < MR; ++f) {
    const int P = min(P0 + (wm * MR + f) * 32 + l31, gm.Mtotal - 1);      // slots past the end recompute the last pixel and are dropped
    const int b = P / HW, rem = P - b * HW;
    const int y = rem / W, x = rem - y * W;
    const int dv = (b - b0) * (H + 4) + y - y0;                             // halo row of the pixel at tap row 0 ... +ky
    aaddr[f] = lds0 + (unsigned)(h * PLANE + dv * pitch + x) * 16u;
  }
#pragma unroll
  for (int g = 0; g < NR; ++g) baddr[g] = lds0 + (unsigned)(WB0 + h * BN + (wn * NR + g) * 32 + l31) * 16u;

  f32x16 acc[MR][NR];
#pragma unroll
  for (int f = 0; f < MR; ++f)
#pragma unroll
    for (int g = 0; g < NR; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  // ---- weights: stage (chunk, s) = taps 3s..3s+2, units 2*chunk, 2*chunk+1.  Wave w moves, for each tap, unit w/4,
  // output-channel quarter w%4 (a 1-KB piece); the LDS image [tap][unit][BN] is lane-linear.  (Giving the DMA of a stage
  // to one wave of each SIMD while its partner computes was measured 6 % SLOWER: the barrier waits for the loaders.)
  const int cin8 = Cin >> 3;
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.wp)), 0,
                                                       (int)((size_t)KS * KS * Cin * CoutP * 2), 0x00020000);
  const unsigned wvoff = (unsigned)(n0 + lane) * 16u;
  const int wu = wid >> 2, wq = wid & 3;
  const unsigned wtap_stride = (unsigned)(cin8 * CoutP * 16);
  auto w_piece = [&](int chunk, int s, int i, int buf, bool real) {      // tap i of the stage; !real: a padding DMA (zeros into the spare slots)
    const unsigned soff = real ? (unsigned)(3 * s + i) * wtap_stride + (unsigned)(((chunk * 2 + wu) * CoutP + wq * 64) * 16) : 0u;
    f32x4* dst = real ? lds + WB0 + buf * WST + (i * 2 + wu) * BN + wq * 64 : lds + DUMMY;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, real ? wvoff : 0x80000000u, soff, 0, 0);
  };

  // ---- halo: piece q = slots 64q..64q+63 of a plane; slot s -> (halo row j, position in row); position < 4 is the
  // zero gap; halo row j is virtual row y0-4+j of the two-image column [image b0 | 4 zero rows | image b0+1].
  const int nimg = min(2, a.B - b0);
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.x)) + (size_t)b0 * HW * Cin, 0,
                                                       (int)((size_t)nimg * HW * Cin * 2), 0x00020000);
  auto halo_piece = [&](int q, int plane, int chunk, bool real) {
    const int s = q * 64 + lane;
    const int j = (int)(((unsigned)s * (unsigned)gm.magic) >> 22);
    const int pos = s - j * pitch;
    int y = y0 - 4 + j, img = 0;
    if (y >= H + 4) { y -= H + 4; img = 1; }
    const bool ok = real && pos >= 4 && j < gm.rows && y >= 0 && y < H && img < nimg;
    const unsigned voff = ok ? (unsigned)(((img * H + y) * W + pos - 4) * Cin * 2) : 0x80000000u;
    f32x4* dst = real ? lds + plane * PLANE + q * 64 : lds + DUMMY;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, (unsigned)(chunk * 32 + plane * 16), 0, 0);
  };
  auto bulk_halo = [&](int chunk) {
    for (int q = wid; q < gm.npieces; q += NT / 64) {
      halo_piece(q, 0, chunk, true);
      halo_piece(q, 1, chunk, true);
    }
  };

  const int nchunk = Cin >> 4;
  const int G = nchunk * NSTAGE;

  // ---- prologue: three weight stages + the whole halo of chunk 0, everything landed before the first read
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < TPS; ++i) w_piece(0, s, i, s, true);
  bulk_halo(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4 fa[2][MR], fb[NR];
  s_a_load<0>(fa[0], aaddr);
#pragma unroll
  for (int g = 0; g < NR; ++g) s_b_load<0>(fb[g], baddr[g]);

  // One stage; PAR = which A buffer holds its step 0.  A stage has 3 steps, so PAR flips every stage: the loop body is a
  // PAIR of stages in straight-line code (G is even: Cin % 32 == 0), never a branch on the parity.
  int g = 0, buf = 0, chunk = 0, st = 0;
  auto one_stage = [&](auto par) {
    constexpr int PAR = decltype(par)::value;
    // ---- barrier that opens stage g: this wave's pieces of stage g+1 have landed (the newest batch stays in flight);
    // afterwards everybody's have, and ring slot (g-1)%4 and the halo pieces that died with stage g-1 are free.
    // Every wave issues the same number of DMAs at every stage (padding ones at the tail), so the count is static.
    if (wid < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // the DMA batch of this stage, spread behind the MFMA groups of step 0: weights of stage g+3, one halo piece
    const bool wreal = g + 3 < G;
    int c3 = chunk, s3 = st + 3;
    if (s3 >= NSTAGE) { s3 -= NSTAGE; ++c3; }
    const int buf3 = (buf + 3) & 3;
    unsigned sw = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) sw = (st >> 2) == i ? gm.sched[i] : sw;
    const unsigned sb = (sw >> ((st & 3) * 8)) & 0xffu;
    const int hchunk = chunk + ((sb >> 6) & 1);
    const bool hreal = async_halo && (sb & 0x80u) && hchunk < nchunk && !(chunk == 0 && !(sb & 0x40u));
    auto dma = [&](int Gc) {
      if (Gc < TPS) w_piece(c3, s3, Gc, buf3, wreal);
      else if (wid < 2) halo_piece((int)(sb & 31u), wid, hchunk, hreal);
    };
    auto nothing = [](int) {};

    // step 0 (tap 3s): reads of step 1 go out, DMA issue interleaved
    s_a_load<1>(fa[PAR ^ 1], aaddr);
    s_rot_g<PAR, 0, 0>(fa, fb, baddr, acc, dma);
    // step 1
    s_a_load<2>(fa[PAR], aaddr);
    s_rot_g<PAR, 1, 0>(fa, fb, baddr, acc, nothing);
    // addresses of the next stage: 3 taps on, next kernel row, or back to tap 0 of the next chunk; next ring slot
    {
      const int dslots = (st % 3 != 2) ? 3 : (st != NSTAGE - 1 ? pitch - 6 : -(8 * pitch + 6));
      const unsigned da = (unsigned)(dslots * 16);
#pragma unroll
      for (int f = 0; f < MR; ++f) aaddr[f] += da;
      const unsigned db = buf == 3 ? (unsigned)(-3 * WST * 16) : (unsigned)(WST * 16);
#pragma unroll
      for (int gq = 0; gq < NR; ++gq) baddr[gq] += db;
      buf = (buf + 1) & 3;
    }
    // step 2: its B reads and these A reads belong to step 0 of stage g+1 (landed at this stage's barrier)
    s_a_load<0>(fa[PAR ^ 1], aaddr);
    s_rot_g<PAR, 2, 0>(fa, fb, baddr, acc, nothing);

    if (!async_halo && st == NSTAGE - 1 && chunk + 1 < nchunk) {
      // boundary tile / no schedule: reload the whole halo for the next chunk between two barriers
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      bulk_halo(chunk + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      s_a_load<0>(fa[PAR ^ 1], aaddr);
#pragma unroll
      for (int gq = 0; gq < NR; ++gq) s_b_load<0>(fb[gq], baddr[gq]);
    }
    ++g;
    if (++st == NSTAGE) { st = 0; ++chunk; }
  };
  for (int gp = 0; gp < G; gp += 2) {
    one_stage(std::integral_constant<int, 0>{});
    one_stage(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- epilogue: bias, ReLU, folded BatchNorm -> bf16 NHWC at the flat pixel index
#pragma unroll
  for (int gq = 0; gq < NR; ++gq) {
    const int co = n0 + (wn * NR + gq) * 32 + l31;
    if (co >= Cout) continue;
    const float bi = a.bias[co];
    float sc = 1.f, sh = 0.f;
    if (a.relu_bn) { sc = a.scale[co]; sh = a.shift[co]; }
#pragma unroll
    for (int f = 0; f < MR; ++f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int P = P0 + (wm * MR + f) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (P < gm.Mtotal) {
          float v = acc[f][gq][i] + bi;
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc + sh;
          static_cast<__bf16*>(a.out)[(size_t)P * Cout + co] = static_cast<__bf16>(v);
        }
      }
    }
  }
}

// ---- host side: geometry + the halo refill schedule -------------------------------------------------------------
namespace {

// Piece q of the next chunk may be issued once the last stage that reads it (in the current chunk) is over, and must
// be issued 3 stages before the first stage that reads it (a batch issued behind barrier g is waited for at barrier
// g+2, and the first fragments of a stage are requested one stage early).  Kernel row ky = stages 3ky..3ky+2 reads
// halo rows ky .. ky+Dn-1 of a non-boundary tile.  Times are stages from the start of the current chunk; slot = time
// mod 27; a slot holds one piece.  Returns false if no assignment exists.
bool make_schedule(int pitch, int Dn, int npieces, unsigned (&sched)[7]) {
  int lo[32], hi[32], ids[32], n = 0;
  for (int q = 0; q < npieces; ++q) {
    const int r1 = (64 * q) / pitch, r2 = (64 * q + 63) / pitch;
    if (r1 > Dn + 7) continue;                         // rows no non-boundary tile reads
    const int death = 3 * (r2 < 8 ? r2 : 8) + 2;
    const int need = NSTAGE + 3 * (r1 - (Dn - 1) > 0 ? r1 - (Dn - 1) : 0);
    lo[n] = death + 1;
    hi[n] = need - 3;
    ids[n] = q;
    if (lo[n] > hi[n] || n >= NSTAGE) return false;
    ++n;
  }
  // bipartite matching pieces -> slots (augmenting paths); a slot t in [lo,hi] maps to slot t % 27
  int owner[NSTAGE], when[32];
  for (int s = 0; s < NSTAGE; ++s) owner[s] = -1;
  struct M {
    static bool aug(int p, const int* lo, const int* hi, int* owner, int* when, bool* seen) {
      for (int t = lo[p]; t <= hi[p] && t < lo[p] + NSTAGE; ++t) {
        const int s = t % NSTAGE;
        if (seen[s]) continue;
        seen[s] = true;
        if (owner[s] < 0 || aug(owner[s], lo, hi, owner, when, seen)) {
          owner[s] = p;
          when[p] = t;
          return true;
        }
      }
      return false;
    }
  };
  for (int p = 0; p < n; ++p) {
    bool seen[NSTAGE] = {};
    if (!M::aug(p, lo, hi, owner, when, seen)) return false;
  }
  unsigned char bytes[28] = {};
  for (int p = 0; p < n; ++p) {
    const int t = when[p];
    bytes[t % NSTAGE] = (unsigned char)(0x80 | (t < NSTAGE ? 0x40 : 0) | ids[p]);
  }
  for (int i = 0; i < 7; ++i) sched[i] = bytes[4 * i] | (bytes[4 * i + 1] << 8) | (bytes[4 * i + 2] << 16) | ((unsigned)bytes[4 * i + 3] << 24);
  return true;
}

bool make_geom(const ConvArgs& a, Geom& gm) {
  if (a.Cin % 32 || a.CoutP % BN || a.W < 8 || a.H < 1) return false;      // Cin % 32: stages are processed in pairs
  const long long HW = (long long)a.H * a.W, M = HW * a.B;
  if (HW < BM || M >= (1ll << 30) || 2 * HW * a.Cin * 2 >= (1ll << 31)) return false;   // a strip touches at most two images
  gm.H = a.H; gm.W = a.W; gm.HW = (int)HW; gm.pitch = a.W + 4;
  const int Dn = (BM - 1 + a.W - 1) / a.W + 1;          // image rows a strip can touch
  gm.rows = Dn + 12;                                     // + 4 above, 4 below, 4 zero rows between two images
  const int slots = gm.rows * gm.pitch + 4;
  if (slots > PLANE) return false;
  gm.npieces = (slots + 63) / 64;
  gm.magic = ((1 << 22) + gm.pitch - 1) / gm.pitch;
  for (int s = 0; s < PLANE; ++s)
    if ((int)(((unsigned)s * (unsigned)gm.magic) >> 22) != s / gm.pitch) return false;
  gm.Mtotal = (int)M;
  gm.mtiles = (int)((M + BM - 1) / BM);
  gm.nN = a.CoutP / BN;
  for (unsigned& w : gm.sched) w = 0;
  gm.async_ok = make_schedule(gm.pitch, Dn, gm.npieces, gm.sched) ? 1 : 0;
  return true;
}

}  // namespace

bool conv_strip_bf16_supported(const ConvArgs& a, int ks) {
  static const bool enabled = [] { const char* e = std::getenv("JCM_STRIP"); return !(e && e[0] == '0'); }();   // JCM_STRIP=0: A/B against the patch kernel
  Geom gm;
  return enabled && ks == KS && make_geom(a, gm);
}

hipError_t conv_strip_bf16(const ConvArgs& a, hipStream_t st) {
  Geom gm;
  if (!make_geom(a, gm)) return hipErrorInvalidValue;
  int blocks;
  if ((8 % gm.nN) == 0) {
    const int per = 8 / gm.nN;
    blocks = (gm.mtiles + per - 1) / per * 8;
  } else {
    blocks = gm.mtiles * gm.nN;
  }
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_strip_bf16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_strip_bf16_kernel, dim3(blocks), dim3(NT), LDS_BYTES, st, a, gm);
  return hipGetLastError();
}

}  // namespace jcm
