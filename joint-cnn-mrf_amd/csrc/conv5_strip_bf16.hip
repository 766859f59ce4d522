// 5x5 SAME convolution with 128-channel output tiles (conv2_*: 64 -> 128, conv3_*: 128 -> 256 of the part detector, main.py:45-47)
// on bf16 MFMA with 768-pixel strips.
//
// The patch kernel of conv_igemm_bf16.hip gives these layers a 6x32 pixel x 128 channel tile: every tile streams all
// 25 x Cin x 128 weights (0.8 MB at Cin = 128) from L2 for 192 pixels -- 190 FLOP per byte, 6.3 GB per launch -- and a wave
// owns 3 x 1 fragments (4 LDS reads per 3 MFMAs).  Measured 20-37 % of the MFMA peak.  Here:
//
//   * M tile = 768 consecutive positions of one image's padded-flattened pixel axis s = r*P + 2 + x (P = W + 2: the two
//     zero slots in front of a row are also the right pad of the row above), so a tap is ONE uniform shift
//     (ky-2)*P + (kx-2) of the A operand and a tile streams the weights once per 768 pixels (4x less L2 traffic);
//     8 waves x (3 x 4 fragments): 96 pixels x all 128 channels per wave, 7 LDS reads per 12 MFMAs as in conv_strip_bf16.hip.
//     The last tile of an image runs with 2 or 1 fragment rows per wave.
//   * 16-channel chunks.  The window of the input the tile reads (768 + 4P + 4 slots x 2 unit planes) is double-buffered
//     and filled by LDS-DMA one row part at a time while the previous chunk computes (table in two VGPRs, v_readlane);
//     weights: one kernel row = 5 taps x 16 channels x 128 columns (20 KB) per stage in a 3-deep ring, requested two
//     stages ahead.  One barrier per stage (60 MFMAs per wave).
//   * D^T = W^T X^T with the channel permutation of conv_strip_bf16.hip: a lane owns a pixel and stores 8 channels at a time.
//   * A tile is short (77-154 k MFMA cycles per SIMD), so what surrounds the MFMA loop counts: the work groups are persistent, and
//     the NEXT item's table, first window and first weight stages are issued before the CURRENT item's epilogue (which
//     works from registers); bias / scale / shift of the item's 128 channels wait in LDS.  Measured on conv2_fullres: prologue
//     17 k + epilogue 26 k cycles per tile beside 94 k of loop before this overlap.
// Reference semantics: conv2d SAME stride 1 + bias + ReLU + BatchNorm (main.py:133-135,156-169).
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace c5 {
constexpr int KS = 5, BN = 128, NR = 4, NW = 8, NT = NW * 64, MRMAX = 3;
constexpr int NB = 3;                                  // weight ring depth (stages)
constexpr int WST = KS * 2 * BN;                       // slots per weight stage: [kx][unit][BN]
constexpr int WB0 = 0, XB0 = NB * WST;                 // weights first (a masked-off low lane of a row part may point below its buffer)
constexpr int WINMAX = 1544;                           // window slots per unit plane: 768 + 4P + 4, W <= 191
constexpr int XBUF = 2 * WINMAX;
constexpr int EC0 = XB0 + 2 * XBUF;                    // epilogue constants, staged once per work group: [bias, scale, shift][CMAX] floats
constexpr int CMAX = 256;                              // output channels (CoutP) the staging area holds
constexpr int LDS_BYTES = (EC0 + 3 * CMAX / 4) * 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int XS = 3, EPS = 4, NXI = XS * EPS;         // window-part table: stages of a chunk that issue parts (0 .. XS-1), entries per wave and stage, entries per wave

struct Geom {
  int H, W, HW, P, S, tpi, nN, items, nparts;         // S = H*P padded positions per image; tpi tiles per image; nN 128-channel tiles
};
struct Item { int b, s0, n0, mr; };                    // image, first position, first output channel, fragment rows per wave
struct Tab { unsigned td, ts, tw, nxs; };              // the wave's window-part table (lanes 0..NXI-1: LDS address, source offset, pixel lanes) and its per-stage counts
constexpr unsigned kZeroSrc = 0x40000000u;             // a source offset beyond any tensor this kernel accepts (< 2^30 bytes per image): the DMA delivers zeros
}  // namespace c5

using namespace c5;

#define C5_T(i) do { } while (0)

template <int TP>
__device__ __forceinline__ void c5_a_load(f32x4 (&fa)[MRMAX], const unsigned (&aaddr)[MRMAX], int mr) {
#pragma unroll
  for (int f = 0; f < MRMAX; ++f)
    if (f < mr) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[f]) : "v"(aaddr[f]), "i"(TP * 16) : "memory");
}
template <int TP, int G>
__device__ __forceinline__ void c5_b_load(f32x4& fb, unsigned baddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb) : "v"(baddr), "i"(TP * 2 * BN * 16 + G * 512) : "memory");
}

__device__ __forceinline__ Item c5_item(const ConvArgs& a, const Geom& gm, int L) {
  Item it;
  it.b = L % a.B;
  const int r = L / a.B;
  const int nt = r % gm.nN, ti = r / gm.nN;             // the channel tiles of one strip follow each other on the same CU: its window is an L2 hit
  it.n0 = nt * BN;
  it.s0 = ti * (NW * MRMAX * 32);
  const int left = gm.S - it.s0;
  it.mr = left > NW * 2 * 32 ? 3 : left > NW * 32 ? 2 : 1;
  return it;
}
// The image's descriptor starts TWO PIXEL STEPS in front of the image: lane L of a row's first part is slot L of the padded row = pixel L - 2, so with
// this base one lane offset (L pixel steps) serves every part and all that differs between parts is a scalar.  The lanes of the pad slots are never
// enabled for a pixel DMA, so nothing in front of the image is read.
__device__ __forceinline__ unsigned c5_xbias(const ConvArgs& a) { return a.in_planar ? 32u : (unsigned)(a.Cin * 4); }
__device__ __forceinline__ auto c5_xrsrc(const ConvArgs& a, const Geom& gm, int b) {
  const unsigned bias = c5_xbias(a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<__bf16*>(static_cast<const __bf16*>(a.x)) + (size_t)b * gm.HW * a.Cin) - bias, 0,
                                           (int)((size_t)gm.HW * a.Cin * 2 + bias), 0x00020000);
}
__device__ __forceinline__ auto c5_wrsrc(const ConvArgs& a) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.wp)), 0, (int)((size_t)KS * KS * a.Cin * a.CoutP * 2), 0x00020000);
}
// window part i of this wave (table lane i) for `chunk` into window `bufsel`: the pixels of one row part.  Everything entry-specific is scalar (three
// v_readlane): the LDS address, the source offset, the enabled lanes; xvoff = the lane's pixel step, the same for every part (c5_xrsrc).  (Round 5: the
// MFMA loop is bound by its issue slots -- this call was 40 instructions, 12 per chunk and wave.)
template <class R>
__device__ __forceinline__ void c5_xdma(R xrsrc, const Tab& t, int i, unsigned chunk_off, unsigned buf_off, unsigned xvoff, int lane) {
  const unsigned w = __builtin_amdgcn_readlane(t.tw, i);
  if (!(w >> 16)) return;
  const unsigned d = __builtin_amdgcn_readlane(t.td, i) + buf_off, so = __builtin_amdgcn_readlane(t.ts, i) + chunk_off;
  if ((unsigned)lane - (w & 0xffu) < ((w >> 8) & 0xffu))
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(size_t)d, 16, xvoff, so, 0, 0);
}
// the zero slots of entry i -- the two pad slots in front of a row, whole rows above / below the image -- in both windows, once per item: no pixel DMA ever
// writes them, so they stay zero through the item's chunks.  Zeros come from the per-LANE offset kZeroSrc (the lane offset is what the buffer's range check sees).
template <class R>
__device__ __forceinline__ void c5_zdma(R xrsrc, const Tab& t, unsigned tz, int i, int lane) {
  const unsigned z = __builtin_amdgcn_readlane(tz, i);
  if (!(z >> 16)) return;
  const unsigned d = __builtin_amdgcn_readlane(t.td, i);
  if ((unsigned)lane - (z & 0xffu) < ((z >> 8) & 0xffu)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(size_t)d, 16, kZeroSrc, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(size_t)(d + XBUF * 16), 16, kZeroSrc, 0, 0, 0);
  }
}
// weights of stage (chunk, ky) into ring slot `slot`: 20 pieces of 1 KB = (kx, unit, 64-column half); wave w moves pieces w, w+8, w+16.
// Past the last stage the source offset runs past the item's data (or out of the buffer: zeros) into a free slot: no tail logic.
// What depends on the wave only -- a piece's offset inside a stage of the source and inside a ring slot -- is formed once per item (WPieces); a call
// adds the stage's scalar terms: the DMA's vector offset is the lane's 16 bytes alone (the loop is bound by its issue slots: timing build, round 5).
struct WPieces { unsigned src[3], dst[3]; };      // bytes: source offset without the stage term, LDS address inside ring slot 0
__device__ __forceinline__ WPieces c5_wpieces(const ConvArgs& a, unsigned lds0, int wid, int n0) {
  WPieces w;
  const unsigned wtap = (unsigned)((a.Cin >> 3) * a.CoutP * 16);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = wid + NW * i;
    const int kx = piece >> 2, unit = (piece >> 1) & 1, q = piece & 1;
    w.src[i] = (unsigned)kx * wtap + (unsigned)((unit * a.CoutP + n0 + q * 64) * 16);
    w.dst[i] = lds0 + (unsigned)(WB0 + (kx * 2 + unit) * BN + q * 64) * 16u;
  }
  return w;
}
template <class R>
__device__ __forceinline__ void c5_wdma(const ConvArgs& a, R wrsrc, const WPieces& w, int wid, unsigned lane16, int chunk, int ky, int slot) {
  const unsigned wtap = (unsigned)((a.Cin >> 3) * a.CoutP * 16);
  const unsigned stage_src = (unsigned)(ky * KS) * wtap + (unsigned)(chunk * 2 * a.CoutP * 16), stage_dst = (unsigned)(slot * WST * 16);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (wid + NW * i < KS * 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(size_t)(w.dst[i] + stage_dst), 16, lane16, w.src[i] + stage_src, 0, 0);
}

// Everything of an item that does not need the accumulators: the window-part table, chunk 0 and the first two weight stages requested
// (NOT waited for).
// Caller: every wave has finished its LDS reads of the previous item and drained its own DMAs.
__device__ __forceinline__ Tab c5_setup(const ConvArgs& a, const Geom& gm, char* smem, const Item& it) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = gm.H, W = gm.W, P = gm.P, Cin = a.Cin;
  const int WIN = NW * it.mr * 32 + 4 * P + 4;          // window: s0 - 2P - 2 ... + WIN
  const int sB = it.s0 - 2 * P - 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // Entry u = wid + 8 i (i = EPS * stage + j, stages 0..XS-1) lives in lane i: one 64-slot part of one window row of one unit plane.  A row
  // is [2 pad slots | W pixels]; the parts tile it from the pad on.  td = LDS byte address of the part's lane 0 in window 0; ts = source offset of lane 0
  // (c5_xrsrc's base, without the chunk term); tw = the lanes that fetch pixels, lo | count << 8 | valid << 16; tz = the lanes whose slots are zero
  // (pad slots, rows above / below the image), same packing.  Every slot of the window is covered by one of the two.
  Tab t{0u, 0u, 0u, 0u};
  unsigned tz = 0u;
  if (lane < NXI) {
    const int u = wid + NW * lane;
    const int upr = 2 * gm.nparts;
    const int r_lo = (sB >= 0 ? sB / P : -((-sB + P - 1) / P));
    const int jr = u / upr, rem = u - jr * upr;
    const int r = r_lo + jr, plane = rem & 1, part = rem >> 1;
    const int rowbase = r * P + 64 * part - sB;                                     // window slot of lane 0
    const int lo = max(0, -rowbase), hi = min(min(64, P - 64 * part), WIN - rowbase);
    if (lo < hi) {
      const bool inside = r >= 0 && r < H;
      const int px0 = r * W + 64 * part;                                            // pixel of lane 2 (lane L is slot 64 part + L of the padded row = pixel 64 part + L - 2)
      t.td = lds0 + (unsigned)(XB0 + plane * WINMAX + rowbase) * 16u;
      t.ts = !inside ? 0u : a.in_planar ? (unsigned)((plane * H * W + px0) * 16) : (unsigned)(px0 * Cin * 2 + plane * 16);
      const int plo = part == 0 ? max(lo, 2) : lo;                                  // pixel lanes [plo, hi), zero lanes [lo, zhi)
      const int zhi = !inside ? hi : part == 0 ? min(hi, 2) : lo;
      if (inside && plo < hi) t.tw = (unsigned)plo | ((unsigned)(hi - plo) << 8) | (1u << 16);
      if (lo < zhi) tz = (unsigned)lo | ((unsigned)(zhi - lo) << 8) | (1u << 16);
    }
  }
#pragma unroll
  for (int i = 0; i < NXI; ++i) t.nxs += (__builtin_amdgcn_readlane(t.tw, i) >> 16) << (4 * (i / EPS));   // pixel DMAs per stage, 4 bits each
  __builtin_amdgcn_s_barrier();                           // every wave has drained its DMAs: the ring and the windows are free
  const auto xrsrc = c5_xrsrc(a, gm, it.b);
  const auto wrsrc = c5_wrsrc(a);
  const unsigned xvoff = a.in_planar ? (unsigned)lane * 16u : (unsigned)lane * 16u * (unsigned)(a.Cin >> 3);
#pragma unroll
  for (int i = 0; i < NXI; ++i) c5_zdma(xrsrc, t, tz, i, lane);
#pragma unroll
  for (int i = 0; i < NXI; ++i) c5_xdma(xrsrc, t, i, 0u, 0u, xvoff, lane);
  const WPieces wp = c5_wpieces(a, lds0, wid, it.n0);
  c5_wdma(a, wrsrc, wp, wid, (unsigned)lane * 16u, 0, 0, 0);
  c5_wdma(a, wrsrc, wp, wid, (unsigned)lane * 16u, 0, 1, 1);
  return t;
}

// One item: the MFMA loop, then the NEXT item's setup, then this item's epilogue.
template <int MR>
__device__ __forceinline__ void c5_tile(const ConvArgs& a, const Geom& gm, char* smem, const Item& it, const Tab& t, bool has_next, const Item& nxt, Tab& tn) {
  f32x4* lds = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int H = gm.H, W = gm.W, HW = gm.HW, P = gm.P;
  const int Cin = a.Cin, Cout = a.Cout;
  const int HWo = a.hpool ? H * (W >> 1) : HW;      // pixels per image of the OUTPUT map
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  unsigned aaddr[MRMAX], baddr;
#pragma unroll
  for (int f = 0; f < MR; ++f) aaddr[f] = lds0 + (unsigned)(XB0 + h * WINMAX + (wid * MR + f) * 32 + l31) * 16u;
  // row m of a D^T fragment comes out in lane half (m>>2)&1, register 4*(m>>3) + (m&3): feeding channel 16h' + 4(m>>3) + (m&3) as row m
  // makes a lane's 16 registers 16 consecutive channels (conv_strip_bf16.hip)
  const int bperm = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  baddr = lds0 + (unsigned)(WB0 + h * BN + bperm) * 16u;

  f32x16 acc[MR][NR];
#pragma unroll
  for (int f = 0; f < MR; ++f)
#pragma unroll
    for (int g = 0; g < NR; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  const auto wrsrc = c5_wrsrc(a);
  const auto xrsrc = c5_xrsrc(a, gm, it.b);
  const int nw = wid < KS * 4 - 2 * NW ? 3 : 2;        // weight pieces this wave issues per stage
  const WPieces wp = c5_wpieces(a, lds0, wid, it.n0);
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned xvoff = a.in_planar ? lane16 : lane16 * (unsigned)(Cin >> 3);          // the lane's pixel step (c5_xdma)
  const unsigned xcs = a.in_planar ? (unsigned)(2 * gm.HW * 16) : 32u;                   // source step per 16-channel chunk

  // the setup's requests (chunk 0, two weight stages) and its LDS writes have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  C5_T(0);

  const int nchunk = Cin >> 4;
  const int G = nchunk * KS;                            // stages
  const unsigned P16 = (unsigned)P * 16u;
  f32x4 fa[2][MRMAX], fb[NR];
  int chunk = 0, ky = 0, slot = 0;
  int wchunk = 0, wky = 2;                              // weights of stage g + 2

  // One stage = kernel row ky of one 16-channel chunk = 5 taps.  PAR = the A buffer of its first tap; 5 is odd, so PAR flips every
  // stage: the loop body is a pair of stages (G is even: Cin % 32 == 0).
  auto one_stage = [&](auto parc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    const int bufsel = chunk & 1;
    // the next chunk's window parts go out in this chunk's first XS stages, into the other window
    const bool xgo = ky < XS && chunk + 1 < nchunk;
    const unsigned xchunk_off = (unsigned)(chunk + 1) * xcs, xbuf_off = bufsel ? 0u : (unsigned)(XBUF * 16);
    // aaddr / baddr are running addresses: this stage's kernel row of this chunk's window, this stage's ring slot
    // fragments of tap 0 (the barrier in front of this stage made the window and this stage's weights visible)
    c5_a_load<0>(fa[PAR], aaddr, MR);
    c5_b_load<0, 0>(fb[0], baddr); c5_b_load<0, 1>(fb[1], baddr); c5_b_load<0, 2>(fb[2], baddr); c5_b_load<0, 3>(fb[3], baddr);
    auto step = [&](auto tpc) __attribute__((always_inline)) {
      constexpr int TP = decltype(tpc)::value;
      constexpr int cur = (TP + PAR) & 1;
      constexpr bool more = TP + 1 < KS;
      if constexpr (more) c5_a_load<TP + 1>(fa[cur ^ 1], aaddr, MR);
#pragma unroll
      for (int g = 0; g < NR; ++g) {
        // queue invariant (conv_igemm_bf16.hip): MR + NR - 1 younger reads in flight when B[g] of this tap is needed
        if (more) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MR + NR - 1) : "memory");
        else if (g == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (g == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (g == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < MR; ++f)
          acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[g]), __builtin_bit_cast(bf16x8, fa[cur][f]), acc[f][g], 0, 0, 0);   // D^T: rows = channels, columns = pixels
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {
          if (g == 0) c5_b_load<TP + 1, 0>(fb[0], baddr);
          if (g == 1) c5_b_load<TP + 1, 1>(fb[1], baddr);
          if (g == 2) c5_b_load<TP + 1, 2>(fb[2], baddr);
          if (g == 3) c5_b_load<TP + 1, 3>(fb[3], baddr);
        }
        // DMA of this stage: the weights of stage g + 2 first, then (stages 0..3 of a chunk) two parts of the next chunk's window
        if (TP == 0 && g == 1) c5_wdma(a, wrsrc, wp, wid, lane16, wchunk, wky, slot == 0 ? NB - 1 : slot - 1);
        // (the parts go out in the chunk's first XS = 3 stages: with a fourth, the last parts had one stage -- 3 us -- to arrive before the chunk's last barrier
        // and the loop waited for them: timing build, 30 % of the loop)
        if (TP >= 1 && g == 1 && xgo) c5_xdma(xrsrc, t, EPS * ky + TP - 1, xchunk_off, xbuf_off, xvoff, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
    // next stage: one kernel row down, or back to row 0 of the other window; next ring slot
    const bool last = ky == KS - 1;
    const int oky = ky, oc = chunk;
    {
      const unsigned da = !last ? P16 : (unsigned)(bufsel ? -XBUF * 16 : XBUF * 16) - (unsigned)(KS - 1) * P16;
#pragma unroll
      for (int f = 0; f < MR; ++f) aaddr[f] += da;
      baddr += slot == NB - 1 ? (unsigned)(-(NB - 1) * WST * 16) : (unsigned)(WST * 16);
    }
    chunk += last ? 1 : 0;
    ky = last ? 0 : ky + 1;
    const bool wlast = wky == KS - 1;
    wchunk += wlast ? 1 : 0;
    wky = wlast ? 0 : wky + 1;
    slot = slot == NB - 1 ? 0 : slot + 1;
    // This wave's pieces of the next stage's weights (requested one stage ago) have landed; everything younger may stay in flight: the
    // previous stage's window parts, this stage's weights and window parts.  The chunk's last stage waits for all window parts.
    {
      const bool more_x = oc + 1 < nchunk;
      const int nx1 = (last || !more_x) ? 0 : (int)((t.nxs >> (4 * oky)) & 15u);
      const int nx0 = (oky == 0 || last || !more_x) ? 0 : (int)((t.nxs >> (4 * (oky - 1))) & 15u);
      const int keep = nw + nx0 + nx1;
      if (keep >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      else if (keep == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (keep == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (keep == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (keep == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if (keep == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (keep == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (keep == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (keep == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int g = 0; g < G; g += 2) {
    one_stage(std::integral_constant<int, 0>{});
    one_stage(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead weight DMAs of the last stages (into free ring slots)
  C5_T(1);

  // ---- the next item's setup goes out now: its first window and weights land while this item's results are written
  if (has_next) tn = c5_setup(a, gm, smem, nxt);
  C5_T(2);

  // ---- epilogue: bias, ReLU, folded BatchNorm -> bf16.  acc[f][g][i] = position s0 + (wid*MR+f)*32 + l31, channel n0 + 32 g + 16 h + i
  // (the lane index goes through an empty asm: the addresses below are then computed HERE instead of being hoisted out of the
  // persistent item loop and spilled around the MFMA loop)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int eh = le >> 5, el = le & 31;
  int pp[MRMAX];
  {
    const float rP = 1.0f / (float)P;
#pragma unroll
    for (int f = 0; f < MR; ++f) {
      const int s = it.s0 + (wid * MR + f) * 32 + el;
      const int r = (int)(((float)s + 0.5f) * rP), c = s - r * P;      // exact: s < 2^22
      pp[f] = (c >= 2 && r < H) ? r * W + c - 2 : -1;
      // hpool: the lane of the even pixel x stores max(x, x + 1) at pixel x / 2 of the half-width map (P even: the parity of x is the lane's; its right
      // neighbour is the next lane of the same fragment)
      if (a.hpool) pp[f] = (c >= 2 && r < H && !(c & 1)) ? r * (W >> 1) + ((c - 2) >> 1) : -1;
    }
  }
  const float* ec = reinterpret_cast<const float*>(lds + EC0) + it.n0;
#pragma unroll
  for (int g = 0; g < NR; ++g) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cl = g * 32 + 16 * eh + 8 * u, co = it.n0 + cl;
      if (co >= Cout) continue;                        // Cout % 8 == 0 (checked on the host)
      float bi[8], sc[8], sh[8];
      *reinterpret_cast<f32x4*>(bi) = *reinterpret_cast<const f32x4*>(ec + cl);
      *reinterpret_cast<f32x4*>(bi + 4) = *reinterpret_cast<const f32x4*>(ec + cl + 4);
      *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(ec + CMAX + cl);
      *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(ec + CMAX + cl + 4);
      *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(ec + 2 * CMAX + cl);
      *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(ec + 2 * CMAX + cl + 4);
#pragma unroll
      for (int f = 0; f < MR; ++f) {
        bf16x8 ov;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = acc[f][g][8 * u + k] + bi[k];
          if (a.relu_bn) v = fmaxf(v, 0.f) * sc[k] + sh[k];
          if (a.hpool) v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xF, 0xF, true)));      // the lane pair's other pixel
          ov[k] = static_cast<__bf16>(v);
        }
        if (pp[f] >= 0) {
          const size_t o = a.out_planar ? (((size_t)it.b * (Cout >> 3) + (co >> 3)) * HWo + pp[f]) * 8      // [B][Cout/8][H*W][8]
                                        : ((size_t)it.b * HWo + pp[f]) * Cout + co;
          *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(a.out) + o) = ov;
        }
      }
    }
  }
  C5_T(3);
}

__global__ __launch_bounds__(NT, 2) void conv5_strip_bf16_kernel(ConvArgs a, Geom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int L = blockIdx.x;
  if (L >= gm.items) return;
  {   // bias / scale / shift of all output channels, once (channels past Cout: 0 / 1 / 0)
    float* ec = reinterpret_cast<float*>(reinterpret_cast<f32x4*>(smem) + EC0);
    for (int i = threadIdx.x; i < 3 * CMAX; i += NT) {
      const int k = i / CMAX, co = i - k * CMAX;
      float v = k == 1 ? 1.f : 0.f;
      if (co < a.Cout) v = k == 0 ? a.bias[co] : !a.relu_bn ? v : k == 1 ? a.scale[co] : a.shift[co];
      ec[i] = v;
    }
  }
  Item it = c5_item(a, gm, L);
  Tab t = c5_setup(a, gm, smem, it);
  for (;;) {
    const int Ln = L + gridDim.x;
    const bool has_next = Ln < gm.items;
    const Item nxt = c5_item(a, gm, has_next ? Ln : L);
    Tab tn{};
    if (it.mr == 3) c5_tile<3>(a, gm, smem, it, t, has_next, nxt, tn);
    else if (it.mr == 2) c5_tile<2>(a, gm, smem, it, t, has_next, nxt, tn);
    else c5_tile<1>(a, gm, smem, it, t, has_next, nxt, tn);
    if (!has_next) break;
    L = Ln; it = nxt; t = tn;
  }
}

namespace {
bool make_geom(const ConvArgs& a, Geom& gm) {
  if (a.CoutP % BN || a.CoutP < BN || a.CoutP > CMAX || a.Cout % 8 || a.Cout > a.CoutP || a.Cin % 32 || a.W < 8 || a.H < 1 || a.B < 1) return false;
  if (a.hpool && (a.W & 1)) return false;      // the pooled pair is a lane pair: even width
  const long long HW = (long long)a.H * a.W;
  if (HW * a.Cin * 2 >= (1ll << 30) || (long long)a.H * (a.W + 2) >= (1 << 22) || (long long)KS * KS * a.Cin * a.CoutP * 2 >= (1ll << 31)) return false;
  gm.H = a.H; gm.W = a.W; gm.HW = (int)HW; gm.P = a.W + 2;
  gm.S = a.H * gm.P;
  const int bm = NW * MRMAX * 32;
  if (bm + 4 * gm.P + 4 > WINMAX) return false;
  gm.tpi = (gm.S + bm - 1) / bm;
  gm.nN = a.CoutP / BN;
  if ((long long)gm.tpi * gm.nN * a.B >= (1ll << 30)) return false;
  gm.items = gm.tpi * gm.nN * a.B;
  gm.nparts = (gm.P + 63) / 64;                          // 64-slot parts of a window row (2 pad slots + W pixels)
  // every row a window can touch needs its 2 * nparts entries in the table
  const int rows = (bm + 4 * gm.P + 4 + gm.P - 1) / gm.P + 1;
  if (rows * 2 * gm.nparts > NW * NXI) return false;
  return true;
}
}  // namespace

bool conv5_strip_bf16_supported(const ConvArgs& a, int ks) {
  Geom gm;
  return ks == KS && make_geom(a, gm);
}

hipError_t conv5_strip_bf16(const ConvArgs& a, hipStream_t st) {
  Geom gm;
  if (!make_geom(a, gm)) return hipErrorInvalidValue;
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
  const int blocks = gm.items < ncu ? gm.items : ncu;
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv5_strip_bf16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv5_strip_bf16_kernel, dim3(blocks), dim3(NT), LDS_BYTES, st, a, gm);
  return hipGetLastError();
}

}  // namespace jcm
