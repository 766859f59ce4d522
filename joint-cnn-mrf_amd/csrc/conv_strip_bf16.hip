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
//     the two images of a boundary tile.  It lives in LDS as two 16-byte-unit planes of a 16-CHANNEL chunk,
//     zeroed once, and is (re)filled by LDS-DMA one row part at a time (64 pixels of one plane; the lanes past
//     the row end are masked off, rows outside the image are never loaded).  Row j of the NEXT chunk is loaded
//     while kernel row j+1 of the current chunk runs -- kernel row ky reads rows ky..ky+5, so row j is dead by
//     then -- and the rows that stay live to the end (8..13) right after the chunk boundary, long before kernel
//     row 3 first reads them: the halo never stops the MFMA stream.  All addressing is scalar.
//   * weights: 3 taps x 16 channels x 256 output channels per stage (24.6 KB) in a 4-deep LDS-DMA ring;
//     at the barrier that opens stage g the weights of stage g+1 have ALREADY landed, so the fragments of the
//     next stage's first k-step are requested before the barrier and the barrier has no load behind it.  The
//     two waves of a SIMD (w, w+4) issue their DMA in different k-steps: while one issues, the other computes.
//   * the MFMA is issued as D^T = W^T X^T (operands swapped): a lane then owns one PIXEL and 4 consecutive output
//     channels per accumulator quad; the weight fragments are fed with their 32 channels permuted so that the 16
//     accumulators of a lane are 16 consecutive channels: the epilogue stores 16 bytes at a time instead of 2.
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
constexpr int ROWPAD = 4;                              // zero slots between two halo rows (the taps reach 4 pixels past a row end).  16 would keep
                                                       // fragments that wrap to the next image row bank-conflict free (11 % of the LDS cycles are such
                                                       // conflicts); measured: no change in time or clock, so the smaller halo stays
constexpr int WST = TPS * 2 * BN;                      // slots per weight stage: [tap][unit][BN]
constexpr int WB0 = 2 * PLANE;                         // first weight slot
constexpr int LDS_BYTES = (WB0 + NB * WST) * 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct Geom {
  int H, W, HW, pitch, rows, nparts, Dn;       // rows: halo rows allocated; nparts: 64-pixel parts per row; Dn: rows a strip can touch
  int Mtotal, mtiles, nN, async_ok;
  int nfull, nrem;                             // item list: nfull tile-units as full tiles, then nrem tile-units as 4 quarter tiles each
};
}  // namespace strip

using namespace strip;

// one k16 step = one tap: A at slot offset `tp`, B at [tp][unit h][BN]
template <int TP>
__device__ __forceinline__ void s_a_load(f32x4 (&fa)[MR], const unsigned (&aaddr)[MR]) {
#pragma unroll
  for (int f = 0; f < MR; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(TP * 16) : "memory");
}
template <int TP, int G>      // column G of the wave's 4 B fragments: 32 channels = 512 bytes further on
__device__ __forceinline__ void s_b_load(f32x4& fb, unsigned baddr) {
  static_assert(TP * 2 * BN * 16 + G * 512 < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb) : "v"(baddr), "i"(TP * 2 * BN * 16 + G * 512) : "memory");
}

// MFMAs of (step, column G) and the read of B[G] for the following step.  Queue invariant (see conv_igemm_bf16.hip):
// exactly MR+NR-1 younger ds_reads are in flight when B[G] of the current step is needed.
template <int NC, int PAR, int STEP, int G, class After>      // NC: B fragments (32-channel columns) per wave
__device__ __forceinline__ void s_rot_g(f32x4 (&fa)[2][MR], f32x4 (&fb)[NC], const unsigned baddr, f32x16 (&acc)[MR][NC], After&& after) {
  if constexpr (G < NC) {
    constexpr int cur = (STEP + PAR) & 1;          // a stage has 3 steps: the A double buffer flips parity every stage
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MR + NC - 1) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < MR; ++f)
      acc[f][G] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[G]), __builtin_bit_cast(bf16x8, fa[cur][f]), acc[f][G], 0, 0, 0);   // D^T: rows = channels, columns = pixels
    __builtin_amdgcn_sched_barrier(0);
    s_b_load<(STEP + 1) % TPS, G>(fb[G], baddr);     // STEP == TPS-1: baddr already points at the next stage's buffer
    after(STEP * NC + G);
    __builtin_amdgcn_sched_barrier(0);
    s_rot_g<NC, PAR, STEP, G + 1>(fa, fb, baddr, acc, after);
  }
}

// One work item: the 384-pixel strip `mt` x NC 32-channel columns per wave.  NC = 4: the full 256-channel tile `nt`.
// NC = 1: a quarter of it (64 channels, `sub`), used for the last items of a launch so that the remainder of
// tiles / CUs does not cost a whole extra round of full tiles.
template <int NC>
__device__ __forceinline__ void strip_tile(const ConvArgs& a, const Geom& gm, char* smem, int mt, int nt, int sub) {
  f32x4* lds = reinterpret_cast<f32x4*>(smem);
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
  unsigned aaddr[MR], baddr;
#pragma unroll
  for (int f = 0; f < MR; ++f) {
    const int P = min(P0 + (wm * MR + f) * 32 + l31, gm.Mtotal - 1);      // slots past the end recompute the last pixel and are dropped
    const int b = P / HW, rem = P - b * HW;
    const int y = rem / W, x = rem - y * W;
    const int dv = (b - b0) * (H + 4) + y - y0;                             // halo row of the pixel at tap row 0 ... +ky
    aaddr[f] = lds0 + (unsigned)(h * PLANE + dv * pitch + x) * 16u;
  }
  // Row m of a D^T fragment comes out in lane half h = (m>>2)&1, register 4*(m>>3) + (m&3).  Feeding channel
  // 16h + 4(m>>3) + (m&3) as row m makes a lane's 16 registers 16 CONSECUTIVE channels: two 16-byte stores per fragment.
  const int bperm = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const int col0 = NC == NR ? wn * NR * 32 : sub * 64 + wn * 32;      // first channel of the wave inside the 256-channel tile
  baddr = lds0 + (unsigned)(WB0 + h * BN + col0 + bperm) * 16u;

  f32x16 acc[MR][NC];
#pragma unroll
  for (int f = 0; f < MR; ++f)
#pragma unroll
    for (int g = 0; g < NC; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  // ---- weights: stage (chunk, s) = taps 3s..3s+2, units 2*chunk, 2*chunk+1.  Wave w moves, for each tap, unit w/4,
  // output-channel quarter w%4 (a 1-KB piece); the LDS image [tap][unit][BN] is lane-linear.  (Giving the DMA of a stage
  // to one wave of each SIMD while its partner computes was measured 6 % SLOWER: the barrier waits for the loaders.)
  const int cin8 = Cin >> 3;
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.wp)), 0,
                                                       (int)((size_t)KS * KS * Cin * CoutP * 2), 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;        // the only per-lane DMA offset: n0 goes into the scalar offsets
  // NC = 4: wave w moves unit w/4, quarter w%4 of each of the 3 taps.  NC = 1: only quarter `sub` is needed: 6 pieces,
  // one for each of waves 0..5 (tap w/2, unit w%2).
  const int wu = NC == NR ? wid >> 2 : wid & 1, wq = NC == NR ? wid & 3 : sub;
  const int nw = NC == NR ? TPS : (wid < 6 ? 1 : 0);                 // weight pieces this wave issues per stage
  const int wt0 = NC == NR ? 0 : wid >> 1;                            // its (first) tap inside the stage
  const unsigned wtap_stride = (unsigned)(cin8 * CoutP * 16);
  auto w_piece = [&](int chunk, int s, int i, int buf) __attribute__((always_inline)) {      // tap i of stage (chunk, s) into ring slot buf (prologue; the loop advances scalars)
    const unsigned soff = (unsigned)(3 * s + i) * wtap_stride + (unsigned)(((chunk * 2 + wu) * CoutP + n0 + wq * 64) * 16);
    f32x4* dst = lds + WB0 + buf * WST + (i * 2 + wu) * BN + wq * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, lane16 + soff, 0, 0, 0);
  };

  // ---- halo: row j of a plane = slots j*pitch .. +pitch-1 = [4 zero slots | W pixels]; it holds virtual row y0-4+j of
  // the two-image column [image b0 | 4 zero rows | image b0+1].  One DMA moves one 64-pixel part of one row of one plane;
  // everything but the lane's pixel offset is wave-uniform.
  const int nimg = min(2, a.B - b0);
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.x)) + (size_t)b0 * HW * Cin, 0,
                                                       (int)((size_t)nimg * HW * Cin * 2), 0x00020000);
  // per-lane source offset of a halo part: planar = 16 bytes per pixel (64 pixels of a unit plane are 1 KB contiguous);
  // NHWC = Cin*2 bytes per pixel, computed where it is used (no register kept for it)
  auto hvoff = [&]() __attribute__((always_inline)) { return a.in_planar ? lane16 : lane16 * (unsigned)(Cin >> 3); };
  auto halo_part = [&](int j, int plane, int part, int chunk) __attribute__((always_inline)) -> bool {      // bulk (re)load of one row part; returns whether a DMA was issued
    int y = y0 - 4 + j, img = 0;
    if (y >= H + 4) { y -= H + 4; img = 1; }
    const bool ok = j < gm.rows && y >= 0 && y < H && img < nimg;
    if (ok) {
      const unsigned soff = a.in_planar ? (unsigned)((((img * cin8 + chunk * 2 + plane) * H + y) * W + 64 * part) * 16)
                                        : (unsigned)(((img * H + y) * W + 64 * part) * Cin * 2 + chunk * 32 + plane * 16);
      f32x4* dst = lds + plane * PLANE + j * pitch + 4 + 64 * part;
      if (lane < W - 64 * part)      // lanes past the row end stay off: the next row's slots are not touched
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)dst, 16, hvoff(), soff, 0, 0);
    }
    return ok;
  };
  const int upr = 2 * gm.nparts;                      // DMAs per halo row
  auto bulk_halo = [&](int chunk) __attribute__((always_inline)) {
    for (int u = wid; u < gm.rows * upr; u += NT / 64) {
      const int j = u / upr, r = u - j * upr;
      halo_part(j, r & 1, r >> 1, chunk);
    }
  };
  // zero the two planes once per item: gaps, rows outside the image and the 4 rows between two images are never written
  // again.  (The barrier in front: every wave has finished the LDS reads of the work group's previous item.)
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < 2 * PLANE; i += NT) lds[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int nchunk = Cin >> 4;
  const int G = nchunk * NSTAGE;

  // ---- prologue: three weight stages + the whole halo of chunk 0, everything landed before the first read
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if constexpr (NC == NR) {
#pragma unroll
      for (int i = 0; i < TPS; ++i) w_piece(0, s, i, s);
    } else if (nw) {
      w_piece(0, s, wt0, s);
    }
  }
  bulk_halo(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4 fa[2][MR], fb[NC];
  auto b_load0 = [&]() __attribute__((always_inline)) {      // the B fragments of a stage's step 0
    s_b_load<0, 0>(fb[0], baddr);
    if constexpr (NC == NR) { s_b_load<0, 1>(fb[1], baddr); s_b_load<0, 2>(fb[2], baddr); s_b_load<0, 3>(fb[3], baddr); }
  };
  s_a_load<0>(fa[0], aaddr);
  b_load0();

  // ---- per-stage control kept to a handful of scalar instructions (every SALU op sits between MFMAs of an in-order wave:
  // 200 of them per stage cost 4 % of the kernel).
  // Weights of stage g+3: source offset and LDS address advance by constants; past the last stage the offset runs out
  // of the buffer and the DMA delivers zeros into a free ring slot -- no tail logic.  (The offset rides in the VECTOR operand:
  // a scalar offset is not bounds-checked, and reads behind the buffer leave stale lines of other streams' data in the L2s.)
  const unsigned TS = wtap_stride, CS = (unsigned)(2 * CoutP * 16);
  unsigned wsoff = (unsigned)(9 + wt0) * TS + (unsigned)((wu * CoutP + n0 + wq * 64) * 16);  // stage 3 = taps 9..11 of chunk 0
  unsigned wm0 = lds0 + (unsigned)((WB0 + 3 * WST + (wt0 * 2 + wu) * BN + wq * 64) * 16);
  // Halo parts of an async tile: what THIS wave issues at stage st is fixed for the whole tile, so it is tabulated once,
  // lane st of two VGPRs (read back with v_readlane): ha = source offset without the chunk term,
  // hb = LDS slot | lanes << 12 | next-chunk << 19 | valid << 20.
  //   kernel row ky = st/3 >= 1, its first stage: row ky-1 died with kernel row ky-1 -> its 2*nparts parts, for the NEXT
  //   chunk, go to waves 0..2*nparts-1;  ky == 0: the rows that were live to the end of the previous chunk (8 .. Dn+7),
  //   8 parts per stage, for THIS chunk.
  unsigned htab = 0;                                    // lanes 0..26: hb, lanes 32..58: ha
  if (async_halo && (lane & 31) < NSTAGE) {
    const int stl = lane & 31;
    unsigned ha = 0, hb = 0;
    const int ky = stl / 3, si = stl - 3 * ky;
    int hj = -1, hr = 0, nxt = 0;
    if (ky == 0) {
      const int u = stl * (NT / 64) + wid;
      if (u < gm.Dn * upr) { hj = 8 + u / upr; hr = u % upr; }
    } else if (si == 0 && wid < upr) {
      hj = ky - 1; hr = wid; nxt = 1;
    }
    if (hj >= 0) {
      const int plane = hr & 1, part = hr >> 1;
      int y = y0 - 4 + hj, img = 0;
      if (y >= H + 4) { y -= H + 4; img = 1; }
      if (hj < gm.rows && y >= 0 && y < H && img < nimg) {
        ha = a.in_planar ? (unsigned)((((img * cin8 + plane) * H + y) * W + 64 * part) * 16)
                         : (unsigned)(((img * H + y) * W + 64 * part) * Cin * 2 + plane * 16);
        hb = (unsigned)(plane * PLANE + hj * pitch + 4 + 64 * part) | ((unsigned)min(64, W - 64 * part) << 12) | ((unsigned)nxt << 19) | (1u << 20);
      }
    }
    htab = lane < 32 ? hb : ha;
  }
  const unsigned hcs = a.in_planar ? (unsigned)(2 * HW * 16) : 32u;         // source step per 16-channel chunk

  // One stage; PAR = which A buffer holds its step 0.  A stage has 3 steps, so PAR flips every stage: the loop body is a
  // PAIR of stages in straight-line code (G is even: Cin % 32 == 0), never a branch on the parity.
  int buf = 0, chunk = 0, st = 0, si = 0;
  bool prev_extra = false;
  auto one_stage = [&](auto par) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    // ---- barrier that opens stage g: this wave's pieces of stage g+1 have landed (the newest batch stays in flight);
    // afterwards everybody's have, and ring slot (g-1)%4 and the halo rows that died with stage g-1 are free.
    // Every wave issues 3 weight pieces per stage + at most one halo part.
    {   // the previous batch (nw weight pieces + maybe a halo part) may stay in flight
      const int keep = nw + (prev_extra ? 1 : 0);
      if (keep == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (keep == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (keep == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // halo parts exist only in the first stage of a kernel row and in the first three stages of a chunk: the table is not even
    // looked at elsewhere (two thirds of the stages)
    unsigned eb = 0, ea = 0;
    if (si == 0 || st < 3)
    {
      eb = __builtin_amdgcn_readlane(htab, st);
      ea = __builtin_amdgcn_readlane(htab, st + 32);
    }
    const int hchunk = chunk + (int)((eb >> 19) & 1u);
    const bool extra = (eb >> 20) && hchunk < nchunk && (chunk | (int)((eb >> 19) & 1u)) != 0;   // this-chunk parts: chunk 0 came with the prologue
    // idx = 4 * step + MFMA group: the three weight pieces behind groups 0..2 of step 0, the halo part behind group 3
    auto dma = [&](int idx) __attribute__((always_inline)) {
      if (idx >= 0 && idx < nw) {
        auto dst = (__attribute__((address_space(3))) char*)(size_t)(wm0 + (unsigned)(idx * 2 * BN * 16));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, lane16 + wsoff + (unsigned)idx * TS, 0, 0, 0);
      }
      else if (idx == (NC == NR ? TPS : 1) && extra) {
        auto dst = (__attribute__((address_space(3))) char*)(size_t)(lds0 + (eb & 0xfffu) * 16u);
        if (lane16 < ((eb >> 8) & 0x7f0u))       // lanes past the row end stay off: the next row's slots are not touched
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)dst, 16, hvoff(), ea + (unsigned)hchunk * hcs, 0, 0);
      }
    };
    auto nodma = [](int) __attribute__((always_inline)) {};
    auto dma1 = [&](int idx) __attribute__((always_inline)) { if constexpr (NC != NR) dma(idx); };   // NC = 1: one MFMA group per step, the DMA slots 1.. follow in step 1

    // step 0 (tap 3s): reads of step 1 go out, DMA issue interleaved
    s_a_load<1>(fa[PAR ^ 1], aaddr);
    s_rot_g<NC, PAR, 0, 0>(fa, fb, baddr, acc, dma);
    // step 1
    s_a_load<2>(fa[PAR], aaddr);
    s_rot_g<NC, PAR, 1, 0>(fa, fb, baddr, acc, dma1);
    // scalars and addresses of the next stage: 3 taps on, next kernel row, or back to tap 0 of the next chunk; next ring slots
    const bool last = st == NSTAGE - 1;
    {
      wsoff += st == NSTAGE - 4 ? 3u * TS + CS - 81u * TS : 3u * TS;      // stage g+4 opens a chunk when st == 23
      wm0 = buf == 0 ? wm0 - (unsigned)(3 * WST * 16) : wm0 + (unsigned)(WST * 16);   // its ring slot (buf+3)%4 wraps when buf == 0
      const int dslots = si != 2 ? 3 : (!last ? pitch - 6 : -(8 * pitch + 6));
      const unsigned da = (unsigned)(dslots * 16);
#pragma unroll
      for (int f = 0; f < MR; ++f) aaddr[f] += da;
      const unsigned db = buf == 3 ? (unsigned)(-3 * WST * 16) : (unsigned)(WST * 16);
      baddr += db;
      buf = (buf + 1) & 3;
      si = si == 2 ? 0 : si + 1;
    }
    // step 2: its B reads and these A reads belong to step 0 of stage g+1 (landed at this stage's barrier)
    s_a_load<0>(fa[PAR ^ 1], aaddr);
    s_rot_g<NC, PAR, 2, 0>(fa, fb, baddr, acc, nodma);

    prev_extra = extra;
    if (last) {
      if (!async_halo && chunk + 1 < nchunk) {
        // boundary tile / no schedule: reload the whole halo for the next chunk between two barriers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bulk_halo(chunk + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prev_extra = false;
        s_a_load<0>(fa[PAR ^ 1], aaddr);
        b_load0();
      }
    }
    chunk += last ? 1 : 0;       // (not "if (last) ++chunk; else ++st;": LLVM turns that into an increment through a selected
    st = last ? 0 : st + 1;      //  pointer, which keeps both counters in scratch memory and every use of them in VGPRs)
  };
  for (int gp = 0; gp < G; gp += 2) {
    one_stage(std::integral_constant<int, 0>{});
    one_stage(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead DMAs of the last stages (zeros into free ring slots)

  // ---- epilogue: bias, ReLU, folded BatchNorm -> bf16.  acc[f][g][i] = pixel (wm*3+f)*32 + l31, channel
  // (wn*4+g)*32 + 16h + i (see bperm): two 8-channel units, one 16-byte store each.
  int pb[MR], pp[MR];
#pragma unroll
  for (int f = 0; f < MR; ++f) {
    const int P = P0 + (wm * MR + f) * 32 + l31;
    pb[f] = P / HW;
    pp[f] = P < gm.Mtotal ? P - pb[f] * HW : -1;
  }
#pragma unroll
  for (int gq = 0; gq < NC; ++gq) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int co = n0 + col0 + gq * 32 + 16 * h + 8 * u;
      if (co >= Cout) continue;                        // Cout % 8 == 0 (checked on the host)
      float bi[8], sc[8], sh[8];
      *reinterpret_cast<float4*>(bi) = *reinterpret_cast<const float4*>(a.bias + co);
      *reinterpret_cast<float4*>(bi + 4) = *reinterpret_cast<const float4*>(a.bias + co + 4);
      if (a.relu_bn) {
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(a.scale + co);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(a.scale + co + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(a.shift + co);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(a.shift + co + 4);
      }
#pragma unroll
      for (int f = 0; f < MR; ++f) {
        bf16x8 ov;
        bool special = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = acc[f][gq][8 * u + k] + bi[k];
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc[k] + sh[k];
          ov[k] = static_cast<__bf16>(v);
        }
        (void)special;
        if (pp[f] >= 0) {
          const size_t o = a.out_planar ? (((size_t)pb[f] * (Cout >> 3) + (co >> 3)) * HW + pp[f]) * 8      // [B][Cout/8][H*W][8]: 32 pixels x 16 B contiguous
                                        : ((size_t)pb[f] * HW + pp[f]) * Cout + co;
          *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(a.out) + o) = ov;
        }
      }
    }
  }
}

// Persistent launch: one work group per CU walks the item list.  Items 0 .. nfull-1 are full tiles in the XCD-aware order
// (work group w takes w, w + grid, ...: grid is a multiple of 8, so a work group -- and with it an XCD's L2 -- keeps one
// channel tile's weights); the remaining `nrem` tile-units follow as 4 quarter-tile items each, one per work group.
__global__ __launch_bounds__(NT, 2) void conv_strip_bf16_kernel(ConvArgs a, Geom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nN = gm.nN;
  auto unit_of = [&](int L, int& mt, int& nt) __attribute__((always_inline)) {
    if ((8 % nN) == 0) {   // an XCD (items b, b+8, ...) keeps one channel tile: its L2 streams 1/nN of the weights
      const int xcd = L & 7, q = L >> 3, per = 8 / nN;
      nt = xcd % nN;
      mt = q * per + xcd / nN;
    } else {
      nt = L % nN;
      mt = L / nN;
    }
  };
  for (int L = blockIdx.x; L < gm.nfull; L += gridDim.x) {
    int mt, nt;
    unit_of(L, mt, nt);
    if (mt < gm.mtiles) strip_tile<NR>(a, gm, smem, mt, nt, 0);
  }
  // (running the quarter items FIRST, to stagger the output bursts of the work groups, measured 0.5 % slower)
  for (int q = blockIdx.x; q < 4 * gm.nrem; q += gridDim.x) {
    int mt, nt;
    unit_of(gm.nfull + (q >> 2), mt, nt);
    if (mt < gm.mtiles) strip_tile<1>(a, gm, smem, mt, nt, q & 3);
  }
}

// ---- host side: geometry ---------------------------------------------------------------------------------------
namespace {

bool make_geom(const ConvArgs& a, Geom& gm) {
  if (a.Cin % 32 || a.CoutP % BN || a.Cout % 8 || a.W < 8 || a.H < 1) return false;      // Cin % 32: stages are processed in pairs
  const long long HW = (long long)a.H * a.W, M = HW * a.B;
  if (HW < BM || M >= (1ll << 30) || 2 * HW * a.Cin * 2 >= (1ll << 31)) return false;   // a strip touches at most two images
  gm.H = a.H; gm.W = a.W; gm.HW = (int)HW; gm.pitch = a.W + ROWPAD;
  gm.Dn = (BM - 1 + a.W - 1) / a.W + 1;               // image rows a strip can touch
  gm.rows = gm.Dn + 12;                                // + 4 above, 4 below, 4 zero rows between two images
  if (gm.rows * gm.pitch + 4 > PLANE) return false;
  gm.nparts = (a.W + 63) / 64;
  gm.Mtotal = (int)M;
  gm.mtiles = (int)((M + BM - 1) / BM);
  gm.nN = a.CoutP / BN;
  // The in-loop halo refill (see the kernel): row j <= 7 is reloaded during kernel row j+1 by waves 0..2*nparts-1; rows
  // 8..Dn+7 right after the chunk boundary, 8 parts per stage; a part issued behind barrier s is usable from stage s+3 on;
  // kernel row ky (stages 3ky..) first reads row ky+Dn-1.  All of that must hold for the worst strip.
  bool ok = 2 * gm.nparts <= NT / 64;
  for (int j = 8; j < gm.Dn + 8 && ok; ++j) {
    const int issued = ((j - 8) * 2 * gm.nparts) / (NT / 64);          // stage of the chunk in which row j is requested
    const int need = 3 * (j - (gm.Dn - 1) > 0 ? j - (gm.Dn - 1) : 0);
    ok = issued <= 2 && issued + 3 <= need;
  }
  gm.async_ok = ok ? 1 : 0;
  return true;
}

}  // namespace

bool conv_strip_bf16_supported(const ConvArgs& a, int ks) {
  Geom gm;
  return ks == KS && make_geom(a, gm);
}

hipError_t conv_strip_bf16(const ConvArgs& a, hipStream_t st) {
  Geom gm;
  if (!make_geom(a, gm)) return hipErrorInvalidValue;
  int units;                                           // tile-units in the XCD-aware numbering (padded to whole groups of 8)
  if ((8 % gm.nN) == 0) {
    const int per = 8 / gm.nN;
    units = (gm.mtiles + per - 1) / per * 8;
  } else {
    units = gm.mtiles * gm.nN;
  }
  // one work group per CU (153 KB of LDS each).  Whole rounds of full tiles; what is left over, if it fits the chip as
  // quarter tiles, runs as quarter tiles: 7200 units on 256 CUs = 28 rounds + 128 quarter items instead of 29 rounds.
  static std::atomic<int> ncu_cache[64];
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  int ncu = ncu_cache[dev & 63].load();
  if (!ncu) {
    if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
    ncu = ncu / 8 * 8;
    if (ncu < 8) ncu = 8;
    ncu_cache[dev & 63].store(ncu);
  }
  gm.nfull = units / ncu * ncu;
  gm.nrem = units - gm.nfull;
  if (4 * gm.nrem > ncu) { gm.nfull = units; gm.nrem = 0; }
  const int blocks = units < ncu && gm.nrem == 0 ? units : ncu;
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_strip_bf16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_strip_bf16_kernel, dim3(blocks), dim3(NT), LDS_BYTES, st, a, gm);
  return hipGetLastError();
}

}  // namespace jcm
