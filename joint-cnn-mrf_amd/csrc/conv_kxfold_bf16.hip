// The logits layer conv6 (9x9, 512 -> 9, main.py:72) with the kernel COLUMNS folded into the GEMM's N axis.
//
// As an implicit GEMM the layer has N = 9: every activation fragment read from LDS feeds one narrow MFMA, and the
// kernel lives on LDS reads (conv_thin_bf16.hip: 16x16x32 tiles, 39 % of the MFMA rate at best).  Here one MFMA column
// is a (kx, joint) pair: for every kernel ROW ky the kernel computes the 1x1 convolution
//     Y[s][kx*9 + k] += sum_c X[s + (ky-4)*P][c] * W[ky][kx][c][k]          (N = 81 -> 96 = 3 fragments of 32)
// over the padded-flattened pixel axis s = r*P + 4 + x (P = W + 4: the 4 zero slots in front of a row are the right pad
// of the row above), so an activation fragment is read ONCE per kernel row and feeds 3 MFMAs x 32 columns, and the
// logits are the 9-term shifted sum
//     out[s][k] = bias[k] + sum_kx Y[s + kx - 4][kx*9 + k]
// taken once per tile through LDS, in kx order (deterministic).  MFMA work 96/144 of the 16-column kernel's, LDS reads
// per MFMA cycle 4.5x lower.
//
//   * tile = 768 consecutive s of one image (8 waves x 3 fragments x 32) -> 760 outputs (the shifted sum needs +-4);
//     the last tile of an image runs with fewer fragments per wave.  Work group w walks items w, w + grid, ...: item L is
//     tile L / B of image L % B, so with B a multiple of the grid one CU computes one image, tile after tile, and the
//     +-4-row halo a tile shares with its predecessor is an L2 hit.
//   * 16-channel chunks: the window of X the tile reads (768 + 8 P slots x 2 unit planes) and the chunk's weights
//     ([9 ky][2 units][96 columns] x 16 B = 27 KB) are double-buffered in LDS and filled by LDS-DMA while the previous chunk
//     computes: row parts of 64 pixels (lanes outside the row / the window masked off; rows outside the image are never
//     loaded and stay zero), 1-KB weight pieces.  What a wave issues is tabulated once per tile in two VGPRs (v_readlane).
//   * D^T = W^T X^T as in conv_strip_bf16.hip: a lane owns a pixel, so Y goes to LDS column-major with conflict-free
//     32-bit writes and comes back with conflict-free reads along s.
// Reference semantics: conv2d SAME stride 1 + bias, no activation (main.py:133-135, 72).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace kxf {
constexpr int KS = 9, CO = 9, NCOL = 96, NF = 3;      // output channels, MFMA columns (81 used), 32-column fragments
constexpr int NW = 8, NT = NW * 64, MRMAX = 3;
constexpr int WINMAX = 768 + 8 * 94;                  // window slots per unit plane (W <= 90)
constexpr int XBUF = 2 * WINMAX;                      // slots per X buffer (two unit planes)
constexpr int WCH = KS * 2 * NCOL;                    // weight slots per chunk: [ky][unit][column]
constexpr int WB0 = 0, XB0 = 2 * WCH;                 // weights first: a masked-off low lane of an X row part may point below its buffer
constexpr int LDS_BYTES = (XB0 + 2 * XBUF) * 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int YS = 768;                               // epilogue: Ybuf[32 columns][YS] floats, obuf[760 * 9] behind it
constexpr int OB0 = 32 * YS;                          // in floats
static_assert((OB0 + 760 * CO) * 4 <= LDS_BYTES, "epilogue buffers");
constexpr int NXE = 18 * 4;                           // X table entries: up to 18 rows x 2 planes x 2 parts
constexpr int NENT = NXE + WCH / 64;                  // + 27 weight pieces = 99
constexpr int EPW = (NENT + NW - 1) / NW;             // entries per wave: 13

struct Geom {
  int H, W, HW, P, S, tpi, items;                     // S = H*P padded pixels per image; tpi tiles per image
};
}  // namespace kxf

using namespace kxf;

#define KX_T(i) do { } while (0)

template <int MR, int KY>
__device__ __forceinline__ void kx_loads(f32x4 (&fa)[MRMAX], f32x4 (&fb)[NF], const unsigned (&aaddr)[MRMAX], unsigned aoff, unsigned baddr) {
#pragma unroll
  for (int g = 0; g < NF; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[g]) : "v"(baddr), "i"(KY * 2 * NCOL * 16 + g * 512) : "memory");
#pragma unroll
  for (int f = 0; f < MR; ++f) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[f]) : "v"(aaddr[f] + aoff) : "memory");
}

// One tile: s in [s0 - 4, s0 - 4 + NW*MR*32) of image b, outputs s0 .. s0 + NW*MR*32 - 9.
template <int MR>
__device__ __forceinline__ void kx_tile(const ConvArgs& a, const Geom& gm, char* smem, int b, int s0) {
  constexpr int BMQ = NW * MR * 32, OUTQ = BMQ - 8;
  f32x4* lds = reinterpret_cast<f32x4*>(smem);
  float* ldsf = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int H = gm.H, W = gm.W, HW = gm.HW, P = gm.P;
  const int Cin = a.Cin, cin8 = Cin >> 3;
  const int WIN = BMQ + 8 * P;                        // window slots: s0 - 4 - 4P ... + WIN
  const int sB = s0 - 4 - 4 * P;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  unsigned aaddr[MRMAX], baddr;
#pragma unroll
  for (int f = 0; f < MR; ++f) aaddr[f] = lds0 + (unsigned)(XB0 + h * WINMAX + (wid * MR + f) * 32 + l31) * 16u;
  baddr = lds0 + (unsigned)(WB0 + h * NCOL + l31) * 16u;

  f32x16 acc[MR][NF];
#pragma unroll
  for (int f = 0; f < MR; ++f)
#pragma unroll
    for (int g = 0; g < NF; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][g][i] = 0.f;

  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.wp)), 0, (int)((size_t)KS * Cin * NCOL * 2), 0x00020000);
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(static_cast<const __bf16*>(a.x)) + (size_t)b * HW * Cin, 0, (int)((size_t)HW * Cin * 2), 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned xvoff = a.in_planar ? lane16 : lane16 * (unsigned)cin8;            // per-lane source offset of an X row part
  const unsigned xcs = a.in_planar ? (unsigned)(2 * HW * 16) : 32u;                 // source step per 16-channel chunk
  const unsigned wcs = (unsigned)(WCH * 16);

  // ---- DMA table: entry u = wid + 8 i lives in lane i.  tb = LDS slot | lo << 14 | hi << 21 | weights << 28 | valid << 29; ta = source offset
  unsigned ta = 0, tb = 0;
  if (lane < EPW) {
    const int u = wid + NW * lane;
    if (u < NXE) {
      const int r_lo = (sB >= 0 ? sB / P : -((-sB + P - 1) / P));
      const int r = r_lo + (u >> 2), plane = (u >> 1) & 1, part = u & 1;
      const int rowbase = r * P + 4 + 64 * part - sB;                               // window slot of lane 0
      const int lo = max(0, -rowbase), hi = min(min(64, W - 64 * part), WIN - rowbase);
      if (r >= 0 && r < H && lo < hi) {
        ta = a.in_planar ? (unsigned)(((plane * H + r) * W + 64 * part) * 16) : (unsigned)((r * W + 64 * part) * Cin * 2 + plane * 16);
        tb = (unsigned)(XB0 + plane * WINMAX + rowbase) | ((unsigned)lo << 14) | ((unsigned)hi << 21) | (1u << 29);
      }
    } else if (u < NENT) {
      const int pc = u - NXE;
      ta = (unsigned)(pc * 1024);
      tb = (unsigned)(WB0 + pc * 64) | (64u << 21) | (1u << 28) | (1u << 29);
    }
  }
  auto dma = [&](int i, int chunk, int bufsel) __attribute__((always_inline)) {      // entry i of this wave for `chunk` into buffer `bufsel`
    const unsigned eb = __builtin_amdgcn_readlane(tb, i), ea = __builtin_amdgcn_readlane(ta, i);
    if (!(eb >> 29)) return;
    const bool wts = (eb >> 28) & 1u;
    const unsigned slot = (eb & 0x3fffu) + (unsigned)bufsel * (wts ? (unsigned)WCH : (unsigned)XBUF);
    auto dst = (__attribute__((address_space(3))) char*)(size_t)(lds0 + slot * 16u);
    const unsigned lo = (eb >> 14) & 0x7fu, hi = (eb >> 21) & 0x7fu;
    if ((unsigned)lane - lo < hi - lo) {
      if (wts) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, lane16 + ea + (unsigned)chunk * wcs, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)dst, 16, xvoff, ea + (unsigned)chunk * xcs, 0, 0);
    }
  };

  // ---- zero both X buffers (gaps, rows outside the image), then chunk 0 into buffer 0
  __builtin_amdgcn_s_barrier();                        // the previous item's epilogue reads are done
  for (int i = tid; i < 2 * XBUF; i += NT) lds[XB0 + i] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < EPW; ++i) dma(i, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  KX_T(0);
  const int nchunk = Cin >> 4;
  const unsigned P16 = (unsigned)P * 16u;
  f32x4 fa[2][MRMAX], fb[2][NF];
  kx_loads<MR, 0>(fa[0], fb[0], aaddr, 0u, baddr);

  // One chunk = 9 kernel rows; PAR = the fragment set that holds kernel row 0.  9 is odd, so PAR flips every chunk: the loop
  // body is a pair of chunks in straight-line code.
  //
  // Everything that is not an MFMA is issued BETWEEN the MFMAs of a kernel row, one piece per MFMA: the two waves of a SIMD
  // are served alternately, so they reach the end of a row together and whatever is issued in one block there is exposed
  // (measured: 6 ds_reads per row in a block cost 17 % of the kernel, the DMA entries 19 %).
  //
  // The chunk's one barrier sits between kernel rows 7 and 8.  In front of it a wave has completed every LDS read of this
  // chunk's buffers (row 8's fragments were requested during row 7) and its share of the next chunk's DMA has landed; behind
  // it, row 8 requests the first fragments of the next chunk from the other buffers, and the DMA of the chunk after that may
  // overwrite this chunk's buffers from the next row on: no bubble at the chunk boundary.
  auto one_chunk = [&](auto par, int chunk) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    const int bufsel = chunk & 1;
    const unsigned aoff = (unsigned)(bufsel * XBUF * 16), aoffn = (unsigned)((bufsel ^ 1) * XBUF * 16);
    const unsigned bcur = baddr + (unsigned)(bufsel * WCH * 16), bnxt = baddr + (unsigned)((bufsel ^ 1) * WCH * 16);
    const bool more = chunk + 1 < nchunk;
    auto step = [&](auto kyc) __attribute__((always_inline)) {
      constexpr int KY = decltype(kyc)::value;
      constexpr int cur = (KY + PAR) & 1, nxt = cur ^ 1;
      constexpr int KN = KY < KS - 1 ? KY + 1 : 0;                 // the kernel row whose fragments this row requests
      const unsigned ao = KY < KS - 1 ? aoff + (unsigned)KN * P16 : aoffn;
      const unsigned bo = KY < KS - 1 ? bcur : bnxt;
      // request j of the next row's fragments: A0 B0 A1 A2 B1 B2 (fragments beyond MR dropped)
      auto req = [&](int j) __attribute__((always_inline)) {
        if (KY == KS - 1 && !more) return;
        if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[nxt][0]) : "v"(aaddr[0] + ao) : "memory");
        if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nxt][0]) : "v"(bo), "i"(KN * 2 * NCOL * 16) : "memory");
        if (j == 2 && MR > 1) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[nxt][1]) : "v"(aaddr[1] + ao) : "memory");
        if (j == 3 && MR > 2) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[nxt][2]) : "v"(aaddr[2] + ao) : "memory");
        if (j == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nxt][1]) : "v"(bo), "i"(KN * 2 * NCOL * 16 + 512) : "memory");
        if (j == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nxt][2]) : "v"(bo), "i"(KN * 2 * NCOL * 16 + 1024) : "memory");
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this row's fragments (requested one row ago)
      if constexpr (KY == KS - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NM = MR * NF;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int g = m / MR, f = m % MR;
        acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[cur][g]), __builtin_bit_cast(bf16x8, fa[cur][f]), acc[f][g], 0, 0, 0);   // D^T: rows = columns of Y, columns = pixels
        __builtin_amdgcn_sched_barrier(0);
        // behind MFMA m: request m (the last MFMA takes all that are left), then the DMA entries of this row behind the last two
        if (m < NM - 1) req(m);
        else
          for (int j = NM - 1; j < 6; ++j) req(j);
        if (KY < KS - 2 && more) {                                  // rows 0..6 carry the 13 entries of the next chunk
          if (m == (NM >= 3 ? NM - 3 : 0) && 2 * KY < EPW) dma(2 * KY, chunk + 1, bufsel ^ 1);
          if (m == NM - 1 && 2 * KY + 1 < EPW) dma(2 * KY + 1, chunk + 1, bufsel ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    one_chunk(std::integral_constant<int, 0>{}, chunk);
    one_chunk(std::integral_constant<int, 1>{}, chunk + 1);
  }
  // (every wave completed its last LDS read before the last chunk's barrier: the epilogue may reuse the LDS right away)
  KX_T(1);

  // ---- epilogue: the shifted sum over kx, one 32-column fragment at a time through Ybuf[column][s] (the whole LDS
  // is free).  acc[f][g][i]: pixel (wid*MR+f)*32 + l31, column 32 g + 8 (i>>2) + 4 h + (i&3).
  // Thread t sums outputs pq = t and t + 512 for all 9 joints.
  float sum[CO][2];
#pragma unroll
  for (int k = 0; k < CO; ++k) sum[k][0] = sum[k][1] = 0.f;
  constexpr bool TWO = OUTQ > NT;                        // MR = 3: 760 outputs per joint on 512 threads
  // The thread index goes through an empty asm: every address below is then computed HERE.  (Otherwise LLVM hoists the
  // tile-invariant epilogue addresses out of the persistent item loop, spills them around the MFMA loop and reloads
  // them one by one with a full vmcnt(0) wait each: measured 23 000 cycles per tile.)
  int te = tid;
  asm volatile("" : "+v"(te));
  const int eh = (te >> 5) & 1, el = te & 31, ew = te >> 6;
  const int pq1 = te + NT;                               // (threads past the last output read along and drop their sums: no branch around a read)
  // the 9 biases: scalar loads (load and wait in one statement: the compiler may copy the result registers right away)
  typedef float f32x8s __attribute__((ext_vector_type(8)));
  f32x8s bias8;
  float bias9;
  asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(bias8), "=&s"(bias9) : "s"(a.bias) : "memory");
#pragma unroll
  for (int g = 0; g < NF; ++g) {
    if (g) __syncthreads();                             // the previous fragment's reads are done
#pragma unroll
    for (int f = 0; f < MR; ++f)
#pragma unroll
      for (int i = 0; i < 16; ++i) ldsf[(8 * (i >> 2) + 4 * eh + (i & 3)) * YS + (ew * MR + f) * 32 + el] = acc[f][g][i];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int col = 32 * g + n;
      if (col < KS * CO) {
        const int kx = col / CO, k = col % CO;
        sum[k][0] += ldsf[n * YS + te + kx];
        if constexpr (TWO) sum[k][1] += ldsf[n * YS + pq1 + kx];      // te + 512 + 8 < 1032: inside the LDS for every thread
      }
    }
  }
  KX_T(2);
  // bias, then through obuf[pq][k] (stride 9: conflict-free) to coalesced NHWC stores
#pragma unroll
  for (int k = 0; k < CO; ++k) {
    const float bi = k < 8 ? bias8[k < 8 ? k : 0] : bias9;
    if (te < OUTQ) ldsf[OB0 + te * CO + k] = sum[k][0] + bi;
    if (TWO && pq1 < OUTQ) ldsf[OB0 + pq1 * CO + k] = sum[k][1] + bi;
  }
  __syncthreads();
  float* __restrict__ out = static_cast<float*>(a.out) + (size_t)b * HW * CO;
  const int r0 = s0 / P, c0 = s0 - r0 * P;               // scalar; below, (c0 + pq) / P through a float reciprocal (exact: the operand stays below 2^11)
  const float rP = 1.0f / (float)P;
#pragma unroll 2
  for (int o = te; o < OUTQ * CO; o += NT) {
    const int pq = o / CO, k = o - pq * CO;
    const int sr = c0 + pq, dr = (int)(((float)sr + 0.5f) * rP);
    const int r = r0 + dr, cc = sr - dr * P;
    if (cc >= 4 && r < H) out[(r * W + cc - 4) * CO + k] = ldsf[OB0 + o];
  }
  KX_T(3);
}

__global__ __launch_bounds__(NT, 2) void conv_kxfold_bf16_kernel(ConvArgs a, Geom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int L = blockIdx.x; L < gm.items; L += gridDim.x) {
    const int b = L % a.B, ti = L / a.B;
    const int s0 = ti * (NW * MRMAX * 32 - 8);
    const int left = gm.S - s0;                         // outputs still to produce in this image
    if (left > NW * 2 * 32 - 8) kx_tile<3>(a, gm, smem, b, s0);
    else if (left > NW * 32 - 8) kx_tile<2>(a, gm, smem, b, s0);
    else kx_tile<1>(a, gm, smem, b, s0);
  }
}

// weights HWIO fp32 [9][9][Cin][9] -> bf16 [Cin/16][ky][unit][96 columns][8 channels], column = kx * 9 + k (columns 81..95 zero)
__global__ void pack_weights_kxfold_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Cin) {
  const size_t n = (size_t)KS * Cin * NCOL;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = i & 7;
    size_t r = i >> 3;
    const int col = r % NCOL; r /= NCOL;
    const int unit = r & 1; r >>= 1;
    const int ky = r % KS;
    const int chunk = r / KS;
    const int ci = chunk * 16 + unit * 8 + c8;
    float v = 0.f;
    if (col < KS * CO) {
      const int kx = col / CO, k = col % CO;
      v = w[(((size_t)ky * KS + kx) * Cin + ci) * CO + k];
    }
    wp[i] = static_cast<__bf16>(v);
  }
}

namespace {
bool make_geom(const ConvArgs& a, Geom& gm) {
  if (a.Cout != CO || a.relu_bn || a.Cin % 32 || a.W < 8 || a.W > 90 || a.H < 1 || a.B < 1) return false;
  const long long HW = (long long)a.H * a.W;
  if (HW * a.Cin * 2 >= (1ll << 31) || (long long)a.H * (a.W + 4) >= (1 << 24)) return false;
  gm.H = a.H; gm.W = a.W; gm.HW = (int)HW; gm.P = a.W + 4;
  gm.S = a.H * gm.P;
  const int outq = NW * MRMAX * 32 - 8;
  gm.tpi = (gm.S + outq - 1) / outq;
  if ((long long)gm.tpi * a.B >= (1ll << 30)) return false;
  gm.items = gm.tpi * a.B;
  // a window spans at most 18 image rows (the table has 18 x 4 entries)
  if ((NW * MRMAX * 32 + 8 * gm.P + gm.P - 1) / gm.P + 1 > 18) return false;
  return true;
}
}  // namespace

size_t conv_kxfold_weight_bytes(int Cin) { return (size_t)KS * Cin * NCOL * 2; }

hipError_t pack_weights_kxfold(const float* w_hwio, void* wp, int Cin, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_kxfold_kernel, dim3(512), dim3(256), 0, st, w_hwio, static_cast<__bf16*>(wp), Cin);
  return hipGetLastError();
}

bool conv_kxfold_bf16_supported(const ConvArgs& a, int ks) {
  Geom gm;
  return ks == KS && make_geom(a, gm);
}

hipError_t conv_kxfold_bf16(const ConvArgs& a, hipStream_t st) {
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
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv_kxfold_bf16_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_kxfold_bf16_kernel, dim3(blocks), dim3(NT), LDS_BYTES, st, a, gm);
  return hipGetLastError();
}

}  // namespace jcm
