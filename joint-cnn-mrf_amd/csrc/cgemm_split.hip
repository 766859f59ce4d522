// The channel GEMM of the frequency-domain convolutions (conv_fft.hip) on the bf16 matrix cores.
//
// For every frequency f of the transformed maps a stride-1 SAME convolution (main.py:133-135; the 9x9 calls at :49,57,66,71,72) is one
// complex matrix product over the channels,  Y[f][b][co] = sum_ci X[f][b][ci] * W[f][ci][co]  (M = images, K = Cin, N = Cout), and
// there are NY * (NX/2+1) of them per layer (3136 for the 60x90 maps).  Round 2 ran them through a library fp32 GEMM on the fp32 matrix
// pipe (1/16 of the bf16 rate).  This kernel runs them on v_mfma_f32_32x32x16_bf16 with SPLIT operands:
//
//   * a 16-bit x 16-bit product is exact in the fp32 accumulator, so an operand can be fed as a sum of 16-bit parts.  fp32 handles: two FP16
//     parts of spectra scaled by powers of two (conv_fft_common.h: Fp16Scale), products x0w1 + x1w0 + x0w0 -- 22 significant bits; bf16
//     handles: ONE scaled fp16 part (default) or two bf16 parts / three products (the strict arm, "fft_single" = 0).
//   * the complex product is four real ones: Yr += Xr Wr - Xi Wi, Yi += Xr Wi + Xi Wr; the minus is a sign flip of the Xi fragment in
//     registers (one v_xor per VGPR).  Both accumulators of an output element stay in the same lane.
//   * both operands arrive ALREADY SPLIT in the exact image the LDS wants -- the activation spectra from the column pass of the forward
//     transform (conv_fft.hip: cols_fwd_split_kernel), the filter spectra once per handle (weight_spectra_split_kernel) -- tile-major, so
//     that a work group streams two contiguous arrays with LDS-DMA (buffer_load ... lds, 1 KB per wave instruction) and nothing else:
//         Xs[f][m-tile][k16][re|im][part][k-half][MT rows][8 bf16]       Ws[f][n-tile][k16][re|im][part][k-half][128 cols][8 bf16]
//     (an MFMA operand lane holds 8 consecutive k of one row / column = one 16-byte unit; units of consecutive rows are consecutive, so
//     every ds_read_b128 of a fragment is bank-conflict free.)  With two parts the split spectra are exactly as large as fp32 complex
//     ones: no extra HBM traffic on bf16 handles.
//   * two 16-bit parts are exactly as large as the fp32 number: no form inflates HBM traffic.  The 9-channel logits layer gets a 32-column tile.
//   * a stage = one k16 step of all parts; R-deep ring, counted vmcnt + raw s_barrier (the DMA of stage g+R-1 is issued right after
//     the barrier that opens stage g, so R-1 stages are in flight while one computes); the fragments of a product are read while the
//     MFMAs of the previous one run.
//   * work groups b, b+8, ... share an XCD: they are given the tiles of the same frequencies, so the N-tiles that re-read an X slab
//     (and the M-tiles that re-read a W slab) find it in that XCD's L2.
//
// Output: Y[f][b][co] complex fp32 (what the inverse column pass reads), or complex fp16 (Y16 below).  gfx950 only.
#include <atomic>
#include <cmath>
#include <type_traits>

#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace cg {

// NP parts per operand; WM x WN waves, FM x FN 32x32 fragments per wave; R stages in LDS
// HALF: the parts are fp16 (11 significant bits each; operands scaled into the fp16 range by the producers) instead of bf16
// K32: ONE 16-bit part per operand; the two "part" planes of the NP = 2 layout hold the two 16-channel halves of a 32-channel stage instead
// (product s = half s of X times half s of W): the same LDS image, the same DMA pieces, two products per stage that both advance K.
template <int NP_, int WM_, int WN_, int FM_, int FN_, int R_, bool HALF_ = false, bool K32_ = false>
struct Cfg {
  static constexpr int NP = NP_, WM = WM_, WN = WN_, FM = FM_, FN = FN_, R = R_;
  static constexpr bool HALF = HALF_, K32 = K32_;
  static constexpr int NW = WM * WN, NT = 64 * NW;
  static constexpr int MT = 32 * WM * FM, NTL = 32 * WN * FN;
  static constexpr int XST = 4 * NP * MT, WST = 4 * NP * NTL;      // 16-byte units per stage: [re|im][part][k-half][rows]
  static constexpr int STAGE = XST + WST;
  static constexpr int XPW = XST / 64 / NW, WPW = WST / 64 / NW;    // 1-KB DMA pieces per wave per stage
  static constexpr int PW = XPW + WPW;
  static constexpr int LDS_BYTES = R * STAGE * 16;
  static constexpr int NPROD = K32 ? 2 : NP == 2 ? 3 : 1;
  // a ring that fills more than half of the LDS leaves ONE work group per CU: nothing else covers its prologue (the first stages' DMA latency) and its
  // epilogue.  Such tiles run as persistent work groups that request the first stages of their NEXT tile before they store the current one.
  static constexpr bool PERSIST = LDS_BYTES > 80 * 1024;
  // cache policy of the filter-spectra DMA: the fp32 handles' tiles (two fp16 parts, <= 128 rows) are bound by that stream, which is read once per launch --
  // the nontemporal hint (bit 1) keeps it from pushing the re-read activation slabs out of the L2: -1.5 % per fp32 step in a same-box A/B; the one-part
  // bf16 form measured +0.8 % with it and keeps the default
  static constexpr int WAUX = (HALF && !K32) ? 2 : 0;
  static_assert(!K32 || NP == 2, "K32 reuses the two-plane layout");
  static_assert(NTL == 128 || NTL == 32, "column tiles the filter spectra are laid out for");
  static_assert(XST % (64 * NW) == 0 && WST % (64 * NW) == 0, "whole DMA pieces per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert((R - 1) * PW < 64, "vmcnt is a 6-bit counter");
};

// (x part, w part) of product s.  Any order gives the same sum up to fp32 rounding of the accumulator, which carries the whole K sum.
template <class C> __device__ __forceinline__ constexpr int cprod_x(int s);
template <class C> __device__ __forceinline__ constexpr int cprod_w(int s);
template <int NP> __device__ __forceinline__ constexpr int prod_x(int s) { return NP == 2 ? (s == 0 ? 0 : s == 1 ? 1 : 0) : 0; }
template <int NP> __device__ __forceinline__ constexpr int prod_w(int s) { return NP == 2 ? (s == 0 ? 1 : s == 1 ? 0 : 0) : 0; }

template <class C> __device__ __forceinline__ constexpr int cprod_x(int s) { return C::K32 ? s : prod_x<C::NP>(s); }
template <class C> __device__ __forceinline__ constexpr int cprod_w(int s) { return C::K32 ? s : prod_w<C::NP>(s); }

template <int OFF> __device__ __forceinline__ void lds_read(f32x4& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
}

// the fragments of one product: X part PX (re, im) for the wave's FM row blocks, W part PWT (re, im) for its FN column blocks.
// Unit of (plane c, part p, k-half, row) inside a stage: ((c * NP + p) * 2 + k-half) * rows + row; the k-half and the row are in the lane address.
template <class C, int PX, int PWT>
__device__ __forceinline__ void frag_load(f32x4 (&xr)[C::FM], f32x4 (&xi)[C::FM], f32x4 (&wr)[C::FN], f32x4 (&wi)[C::FN], unsigned xaddr, unsigned waddr) {
  constexpr int XR = (0 * C::NP + PX) * 2 * C::MT * 16, XI = (1 * C::NP + PX) * 2 * C::MT * 16;
  constexpr int WR = (0 * C::NP + PWT) * 2 * C::NTL * 16, WI = (1 * C::NP + PWT) * 2 * C::NTL * 16;
  lds_read<XR>(xr[0], xaddr);
  lds_read<XI>(xi[0], xaddr);
  if constexpr (C::FM > 1) {
    lds_read<XR + 512>(xr[1], xaddr);
    lds_read<XI + 512>(xi[1], xaddr);
  }
  lds_read<WR>(wr[0], waddr);
  lds_read<WI>(wi[0], waddr);
  if constexpr (C::FN > 1) {
    lds_read<WR + 512>(wr[1], waddr);
    lds_read<WI + 512>(wi[1], waddr);
  }
}

template <bool HALF>
__device__ __forceinline__ f32x16 mfma16(const f32x4& a, const f32x4& b, const f32x16& c) {
  if constexpr (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Args {
  const char* xs;       // split activation spectra
  const char* ws;       // split filter spectra
  float2* y;            // [F][B][ldy]
  int F, B, ldy, KC;    // ldy: complex numbers per row of y; KC = Cin / 16
  int mtiles, ntiles;
  float yshift;         // Y16: the power of two the products are multiplied by on their way to fp16
};

// Y16 (one-part route of bf16 handles with 16-bit intermediates): Y is written as complex FP16, 4 bytes per product instead of 8 -- half of this
// kernel's stores and half of what the inverse column pass reads.  No data-dependent scale is needed: the operands arrive scaled into known ranges
// (|Xs| < 2^15.5 per image, |Ws| < 2^14, conv_fft_common.h / conv_fft.hip), so |Y| < Cin 2^29.5 and the constant shift 2^-(ceil(log2 Cin) + 14) keeps
// every component below 2^15.5 < 65504; typical entries land near 1, fourteen binades above fp16's smallest normal number.
template <class C, bool Y16>
__global__ __launch_bounds__(C::NT) void cgemm_split_kernel(Args a) {
  static_assert(C::FM <= 2 && C::FN <= 2, "frag_load covers 1 or 2 fragments per side");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FM = C::FM, FN = C::FN, MT = C::MT, NTL = C::NTL, R = C::R, NW = C::NW;
  constexpr int XST = C::XST, WST = C::WST, STAGE = C::STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid % C::WN;
  const int h = lane >> 5, l31 = lane & 31;

  // work groups b, b+8, ... run on one XCD: they walk the tiles of frequencies xcd, xcd + 8, ... one frequency after the other.
  // Persistent form: work group (xcd, s) of gridDim.x / 8 per XCD takes tiles s, s + gridDim.x / 8, ... of that list.
  const int T = a.mtiles * a.ntiles;
  const int bid = blockIdx.x, xcd = bid & 7;
  const int jstep = C::PERSIST ? (int)(gridDim.x >> 3) : 0;
  int j = bid >> 3;
  const int KC = a.KC;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int WBYTES = WST * 16;      // bytes of one stage of the W stream in HBM
  const unsigned lane16 = (unsigned)lane * 16u;
  if ((j / T) * 8 + xcd >= a.F) return;

  // stage g -> ring slot: the LDS image of a stage is its HBM image, X pieces first; wave w moves pieces w, w + NW, ...
  // A DMA is issued for EXISTING stages only (g < KC; the callers guard): rounds 2-3 let the look-ahead run past the last stage and relied on
  // the bounds check of the buffer descriptor ("delivers zeros into a free slot"); nothing is requested that is not used any more.  The stage
  // offset rides in the (bounds-checked) vector offset.
  auto issue = [&](const char* xg, const char* wg, int g, int slot) __attribute__((always_inline)) {
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xg), 0, KC * XST * 16, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wg), 0, KC * WBYTES, 0x00020000);
    const unsigned sbase = (unsigned)(slot * STAGE * 16);
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const unsigned q = (unsigned)(wid + i * NW) * 1024u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(smem + sbase + q), 16, lane16 + (unsigned)g * (unsigned)(XST * 16) + q, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < C::WPW; ++i) {
      const unsigned q = (unsigned)(wid + i * NW) * 1024u;
      // (the cache policy must be a literal: a dependent constant in that operand silently drops the kernel from the host object)
      if constexpr (C::WAUX == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + sbase + XST * 16 + q), 16, lane16 + (unsigned)g * (unsigned)(WST * 16) + q, 0, 0, 2);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + sbase + XST * 16 + q), 16, lane16 + (unsigned)g * (unsigned)(WST * 16) + q, 0, 0, 0);
    }
  };
  auto x_of = [&](int f, int mt) __attribute__((always_inline)) { return a.xs + ((size_t)(f * a.mtiles + mt) * KC) * (size_t)(XST * 16); };
  auto w_of = [&](int f, int nt) __attribute__((always_inline)) { return a.ws + ((size_t)(f * a.ntiles + nt) * KC) * (size_t)WBYTES; };
  bool prefetched = false;      // the first R - 1 stages of this tile were requested behind the previous tile's loop
  for (;;) {
  const int f = (j / T) * 8 + xcd, t = j % T;
  const int mt = t / a.ntiles, nt = t % a.ntiles;
  const char* xrsrc = x_of(f, mt);
  const char* wrsrc = w_of(f, nt);

  f32x16 accr[FM][FN], acci[FM][FN];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int i = 0; i < 16; ++i) { accr[fm][fn][i] = 0.f; acci[fm][fn][i] = 0.f; }

  if (!prefetched) {
#pragma unroll
    for (int s = 0; s < R - 1; ++s)
      if (s < KC) issue(xrsrc, wrsrc, s, s);
  }

  // per-lane fragment addresses inside slot 0: unit (k-half h, row) of plane (re, part 0)
  unsigned xaddr = lds0 + (unsigned)(h * MT + wm * FM * 32 + l31) * 16u;
  unsigned waddr = lds0 + (unsigned)(XST + h * NTL + wn * FN * 32 + l31) * 16u;

  int slot = 0;
  for (int g = 0; g < KC; ++g) {
    // this wave's DMA pieces of stage g have landed (R-2 younger stages may stay in flight); after the barrier everybody's have, and every
    // wave is done reading slot (g-1) % R, which the DMA of stage g+R-1 now refills
    // Nothing is ever requested for a stage past the last one (see issue()): behind the last stage nothing younger is in flight.
    static_assert(R == 2 || R == 3, "the wait below counts at most one younger stage");
    // (a prefetched tile: the stores of the previous tile's epilogue are younger than its first stages' DMAs and count in vmcnt too -- everything is waited for)
    const bool last = g + 1 >= KC || (C::PERSIST && g == 0 && prefetched);
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((R - 2) * C::PW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int ns = slot == 0 ? R - 1 : slot - 1;
    if (g + R - 1 < KC) issue(xrsrc, wrsrc, g + R - 1, ns);
    f32x4 xr[2][FM], xi[2][FM], wr[2][FN], wi[2][FN];
    frag_load<C, cprod_x<C>(0), cprod_w<C>(0)>(xr[0], xi[0], wr[0], wi[0], xaddr, waddr);
    auto product = [&](auto sc) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value, cur = S & 1;
      if constexpr (S + 1 < C::NPROD) {
        frag_load<C, cprod_x<C>(S + 1), cprod_w<C>(S + 1)>(xr[cur ^ 1], xi[cur ^ 1], wr[cur ^ 1], wi[cur ^ 1], xaddr, waddr);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(2 * FM + 2 * FN) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 nxi[FM];
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const u32x4 v = __builtin_bit_cast(u32x4, xi[cur][fm]) ^ u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        nxi[fm] = __builtin_bit_cast(f32x4, v);
      }
      // re += Xr Wr ; im += Xr Wi ; re += (-Xi) Wi ; im += Xi Wr  -- consecutive MFMAs never share an accumulator
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          accr[fm][fn] = mfma16<C::HALF>(xr[cur][fm], wr[cur][fn], accr[fm][fn]);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          acci[fm][fn] = mfma16<C::HALF>(xr[cur][fm], wi[cur][fn], acci[fm][fn]);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          accr[fm][fn] = mfma16<C::HALF>(nxi[fm], wi[cur][fn], accr[fm][fn]);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          acci[fm][fn] = mfma16<C::HALF>(xi[cur][fm], wr[cur][fn], acci[fm][fn]);
      __builtin_amdgcn_sched_barrier(0);
    };
    product(std::integral_constant<int, 0>{});
    if constexpr (C::NPROD > 1) product(std::integral_constant<int, 1>{});
    if constexpr (C::NPROD > 2) product(std::integral_constant<int, 2>{});
    // next ring slot
    const bool wrap = slot == R - 1;
    const unsigned d = wrap ? (unsigned)(-(R - 1) * STAGE * 16) : (unsigned)(STAGE * 16);
    xaddr += d;
    waddr += d;
    slot = wrap ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nothing is in flight here any more; kept as the guard of the LDS hand-back)
  // persistent work groups: the next tile's first stages go out now and land while this tile is stored
  const int jn = j + jstep, fn_ = (jn / T) * 8 + xcd;
  const bool more = C::PERSIST && fn_ < a.F;
  if (more) {
    __builtin_amdgcn_s_barrier();      // every wave has read its last fragments: the ring is free
    const int tn = jn % T;
    const char* xn = x_of(fn_, tn / a.ntiles);
    const char* wn_ = w_of(fn_, tn % a.ntiles);
#pragma unroll
    for (int s = 0; s < R - 1; ++s)
      if (s < KC) issue(xn, wn_, s, s);
  }

  // ---- Y[f][b][co]: accumulator register i of a fragment is row (i&3) + 8 (i>>2) + 4 h, column l31: a half wave stores 32 complex
  // numbers = 256 contiguous bytes per instruction
  const int n0 = nt * NTL + wn * FN * 32 + l31;
  float2* yf = a.y + (size_t)f * a.B * a.ldy;
  unsigned* yh = reinterpret_cast<unsigned*>(a.y) + (size_t)f * a.B * a.ldy;
  const float ys = a.yshift;
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int r0 = mt * MT + (wm * FM + fm) * 32 + 4 * h;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + (i & 3) + 8 * (i >> 2);
      if (row < a.B) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            if constexpr (Y16) {      // a half wave stores 32 complex fp16 numbers = 128 contiguous bytes
              typedef _Float16 h2 __attribute__((ext_vector_type(2)));
              const h2 v{static_cast<_Float16>(accr[fm][fn][i] * ys), static_cast<_Float16>(acci[fm][fn][i] * ys)};
              yh[(size_t)row * a.ldy + n0 + fn * 32] = __builtin_bit_cast(unsigned, v);
              continue;
            }
            float2* q = yf + (size_t)row * a.ldy + n0 + fn * 32;
            // fp32 handles: streaming store (the 64-row tiles are bound by the filter spectra they read; keeping Y out of the L2 measured 366 -> 361 us, the
            // logits layer's thin tile 264 -> 243 us).  The one-part bf16 form writes as much as it reads and measured 832 -> 931 us with it: plain stores.
            constexpr bool kStream = !C::K32;
            if constexpr (kStream) {
              typedef float f2n __attribute__((ext_vector_type(2)));
              __builtin_nontemporal_store(f2n{accr[fm][fn][i], acci[fm][fn][i]}, reinterpret_cast<f2n*>(q));
            } else {
              *q = make_float2(accr[fm][fn][i], acci[fm][fn][i]);
            }
          }
      }
    }
  }
  if (!more) break;
  j = jn;
  prefetched = true;
  }      // tiles of this work group
}

// tile shapes.  128 columns (32 for the thin logits layer); what varies is the M tile, the ring depth and where the W operand is split.
using CfgB256 = Cfg<2, 4, 2, 2, 2, 3>;            // bf16 handles, 129 .. 256 images: the whole micro-batch in one 256 x 128 tile, W read once
using CfgB128 = Cfg<2, 2, 2, 2, 2, 2>;            // bf16 handles, 65 .. 128 images (two work groups per CU)
using CfgB64 = Cfg<2, 1, 4, 2, 1, 2>;             // bf16 handles, <= 64 images
// fp32 handles, np = 4: TWO FP16 parts per operand, three products (22 significant bits; the producers scale the spectra by powers of two so that
// they fit fp16's range, conv_fft.hip) -- half the matrix-core work of the six-product bf16 form, a third less activation-spectra traffic.
// Two fp16 parts are exactly as large as the fp32 number, so the filter spectra are stored split and both operands arrive by LDS-DMA.
using CfgH64 = Cfg<2, 1, 4, 2, 1, 2, true>;                // <= 64 images (ring depth 2: depth 3 measured no faster in round 5)
using CfgH128 = Cfg<2, 2, 2, 2, 2, 2, true>;               // 65 .. 128 images per tile
using CfgH64T = Cfg<2, 2, 1, 1, 1, 3, true>;               // Cout <= 32 (the logits layer): 64 x 32 tile, two waves

// bf16 handles, np = 5: ONE scaled fp16 part per operand (11 significant bits -- eight times finer than the bf16 tensors the layer reads and
// writes), 32 channels per stage: a third of the matrix-core work of the two-part form and half of its operand bytes.
using CfgS256 = Cfg<2, 4, 2, 2, 2, 3, true, true>;
using CfgS128 = Cfg<2, 2, 2, 2, 2, 2, true, true>;
using CfgS64 = Cfg<2, 1, 4, 2, 1, 2, true, true>;

template <class C, bool Y16 = false> hipError_t launch(const Args& a, hipStream_t st) {
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(cgemm_split_kernel<C, Y16>), C::LDS_BYTES); e != hipSuccess) return e;
  const int T = a.mtiles * a.ntiles;
  int blocks = (a.F + 7) / 8 * 8 * T;
  if constexpr (C::PERSIST) {      // one work group per CU, whole groups of 8 (one per XCD)
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
    if (blocks > ncu) blocks = ncu;
  }
  hipLaunchKernelGGL((cgemm_split_kernel<C, Y16>), dim3(blocks), dim3(C::NT), C::LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace cg

// np: 2 = bf16 handles, two bf16 parts; 5 = bf16 handles, ONE scaled fp16 part, 32 channels per stage; 4 = fp32 handles, two fp16 parts (scaled).
// (np = 3, three bf16 parts / six products on fp32 handles, was retired in round 5: dominated by np = 4 on every axis.)
int cgemm_split_ntile(int np, int Cout) { return np == 4 && Cout <= 32 ? 32 : 128; }
int cgemm_split_parts(int np) { return np == 5 ? 1 : 2; }
int cgemm_split_mtile(int np, int B, int Cout) {
  if (cgemm_split_ntile(np, Cout) == 32) return 64;
  if (np == 2 || np == 5) return B > 128 ? 256 : B > 64 ? 128 : 64;
  return B > 64 ? 128 : 64;
}
size_t cgemm_split_w_bytes(int np, int F, int Cin, int Cout) {
  const int ntl = cgemm_split_ntile(np, Cout);
  const size_t coutp = (size_t)(Cout + ntl - 1) / ntl * ntl;
  return (size_t)F * Cin * coutp * 4 * cgemm_split_parts(np);
}

// the shift of the fp16 product spectra (cgemm_split_kernel<C, Y16 = true>) and its inverse, both exact powers of two
float cgemm_split_y16_shift(int Cin) {
  int lg = 0;
  while ((1 << lg) < Cin) ++lg;
  return std::ldexp(1.0f, -(lg + 14));
}

// y16_shift: 0 = Y complex fp32; otherwise (np = 5 only) Y complex fp16 = products * y16_shift, y16_shift = cgemm_split_y16_shift(Cin)
hipError_t cgemm_split(const void* xs, const void* ws, void* y, int np, int F, int B, int Cin, int Cout, int ldy, hipStream_t st, float y16_shift) {
  const int ntl = cgemm_split_ntile(np, Cout);
  const int ntiles = (Cout + ntl - 1) / ntl;
  const int kstep = np == 5 ? 32 : 16;      // channels per stage
  if ((np != 2 && np != 4 && np != 5) || Cin % kstep || ldy < ntiles * ntl || F < 1 || B < 1) return hipErrorInvalidValue;
  const int MT = cgemm_split_mtile(np, B, Cout);
  if (y16_shift != 0.f && np != 5) return hipErrorInvalidValue;
  cg::Args a{static_cast<const char*>(xs), static_cast<const char*>(ws), static_cast<float2*>(y), F, B, ldy, Cin / kstep, (B + MT - 1) / MT, ntiles, y16_shift};
  if ((long long)a.KC * 8 * (MT > ntl ? MT : ntl) * 16 >= (1ll << 31)) return hipErrorInvalidValue;      // buffer descriptor range (at most 8 units per row and stage)
  if (np == 2) return MT == 256 ? cg::launch<cg::CfgB256>(a, st) : MT == 128 ? cg::launch<cg::CfgB128>(a, st) : cg::launch<cg::CfgB64>(a, st);
  if (np == 5 && y16_shift != 0.f) return MT == 256 ? cg::launch<cg::CfgS256, true>(a, st) : MT == 128 ? cg::launch<cg::CfgS128, true>(a, st) : cg::launch<cg::CfgS64, true>(a, st);
  if (np == 5) return MT == 256 ? cg::launch<cg::CfgS256>(a, st) : MT == 128 ? cg::launch<cg::CfgS128>(a, st) : cg::launch<cg::CfgS64>(a, st);
  return ntl == 32 ? cg::launch<cg::CfgH64T>(a, st) : MT == 128 ? cg::launch<cg::CfgH128>(a, st) : cg::launch<cg::CfgH64>(a, st);
}

}  // namespace jcm
