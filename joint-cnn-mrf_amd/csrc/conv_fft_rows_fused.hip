// Fused hand-overs between frequency-domain layers of an fp32 handle where the graph has an op BETWEEN the two convolutions (conv_fft.hip; the plain
// hand-over conv3 -> conv4, conv5 -> conv6 is rows_inv_fwd_kernel / rows_inv_fwd_reg_kernel):
//
//   rows_inv_pool_fwd_kernel   conv2 -> max_pool 2x2/2 SAME -> conv3 (main.py:46-48,54-56,63-65): a work group owns a ROW PAIR of conv2's output --
//                              inverse row transform + bias / ReLU / BatchNorm of both rows, the 2x2 maximum, and the forward row transform of the pooled
//                              row at the NEXT layer's (half) length.  Neither conv2's output nor the pooled map reaches HBM (round 5: 0.93 GB written by
//                              the inverse row pass at 64 images only to be re-read by max_pool_kernel, which wrote the pooled map for conv3's row pass).
//   rows_inv_merge_fwd_kernel  conv4_fullres -> ((x1 + up(x2)) + up(x3)) / 3 -> conv5 (main.py:58,67,69-71): inverse row transform + epilogue of the
//                              full-resolution branch, plus the two coarse branches through the TF-1.x bilinear taps (the arithmetic and association order
//                              of upsample_merge3_kernel / rows_fwd_merge_kernel: lerp along x, then along y, correctly rounded third), then conv5's forward
//                              row transform.  x1 (11 MB per image) is never written; the coarse rows of a tile are staged in LDS, the next tile's already
//                              in flight (registers) while the current one computes.
// Both are persistent LDS kernels with register prefetch like rows_inv_fwd_kernel; T_in[b][y][kx][c] -> T_out[kx][c/16][b][y'][16].
#include "conv_fft_common.h"
#include "resize_tf1.h"

namespace jcm {
namespace cfft {

namespace {
// Z = Y_c + i Y_{c+1} with the Hermitian extension: one half-spectrum row (K float4 per thread) -> buf
template <int NX, int NTR, int K>
__device__ __forceinline__ void fill_hermitian(cf* buf, const float4 (&pre)[K], int tid) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int t = tid + i * NTR, k = t / CH, v = t % CH;
    if (t < NXH * CH) {
      float4 q = pre[i];
      const bool edge = k == 0 || k == NX / 2;
      if (edge) { q.y = 0.f; q.w = 0.f; }
      buf[k * CH + v] = cf{q.x - q.w, q.y + q.z};
      if (!edge) buf[(NX - k) * CH + v] = cf{q.x + q.w, q.z - q.y};
    }
  }
}
// The half spectrum of one (row, 64-channel block) of T'[b][y][kx][c] -> K float4 per thread, entry k = tid / 32 + i * (NTR / 32), channel pair tid % 32.
// Buffer loads: the row is the descriptor (tile-uniform: scalar registers), the thread's (k0, pair) ONE vector offset, the step between a thread's entries a
// scalar offset; entries behind the row (k >= NX / 2 + 1) come back as the zeros of the range check.
template <int NX, int NTR, int K>
__device__ __forceinline__ void fetch_row(float4 (&pre)[K], const cf* __restrict__ T, size_t row, int cblk, int C, int tid) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1, XP = NTR / CH;
  const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(T) + (row * NXH) * C + cblk * CB, 0, (NXH * C - cblk * CB) * 8, 0x00020000);
  const int vo = ((tid / CH) * (C >> 1) + (tid % CH)) * 16;
  const int step = XP * (C >> 1) * 16;
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(d, vo, i * step, 0));
    pre[i] = make_float4(q[0], q[1], q[2], q[3]);
  }
}
struct Epi { float b0, b1, s0, s1, h0, h1, norm; int relu_bn; };
__device__ __forceinline__ cf epi_act(const Epi& e, cf z) {
  float v0 = z.x * e.norm + e.b0, v1 = z.y * e.norm + e.b1;
  if (e.relu_bn) { v0 = fmaxf(v0, 0.f) * e.s0 + e.h0; v1 = fmaxf(v1, 0.f) * e.s1 + e.h1; }
  return cf{v0, v1};
}
}  // namespace

// ---- conv2 -> pool -> conv3.  H x W: conv2's map (NXI-point rows); the pooled map is Ho x Wo = ceil(H/2) x ceil(W/2), its rows NXO points.
template <int NXI, int NXO>
__global__ __launch_bounds__(rows_threads<NXI>()) __attribute__((amdgpu_waves_per_eu(4))) void rows_inv_pool_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Tn, const cf* __restrict__ twgi, const cf* __restrict__ twgo,
                                                                const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn,
                                                                int B, int H, int W, int C, int pad, float norm0, int ntiles, Fp16Scale sc) {
  constexpr int CH = CB / 2, NXH = NXI / 2 + 1, NTR = rows_threads<NXI>(), K = (NXH * CH + NTR - 1) / NTR, XP = NTR / CH;
  constexpr int KP = (NXO + XP - 1) / XP;      // pooled pixels per thread (the thread also writes the zeros behind the pooled row)
  static_assert(NXO <= NXI, "the pooled row is transformed in the buffer of the full row");
  __shared__ cf buf[NXI * CH];
  __shared__ cf twi[NXI];
  __shared__ cf two[NXO];
  __shared__ float par[3 * kParMax];
  __shared__ float red[NTR / 64];
  const float ncommon = sc.tmax ? norm0 * sc.winv[0] * (sc.common ? fp16_unscale(tmax_of(sc.tmax, 0, sc.nb, 1), sc.hf) : 1.f) : norm0;
  const int tid = threadIdx.x, ncb = C / CB, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  twiddles<NXI, NTR>(twi, twgi, tid);
  twiddles<NXO, NTR>(two, twgo, tid);
  const bool par_lds = C <= kParMax;
  if (par_lds)
    for (int i = tid; i < C; i += NTR) {
      par[i] = bias[i];
      par[kParMax + i] = relu_bn ? scale[i] : 1.f;
      par[2 * kParMax + i] = relu_bn ? shift[i] : 0.f;
    }
  __syncthreads();      // the epilogue parameters are read at the top of the first tile, before its first barrier
  // ONE prefetch set: the kernel walks half tiles (row a, row b of a pair); the row after the one being transformed is always in flight
  float4 pre[K];
  float pre_t = 0.f;
  const bool per_image = sc.tmax && !sc.common;
  auto fetch = [&](int tile, int r) __attribute__((always_inline)) {
    const int cblk = tile % ncb, byo = tile / ncb;      // byo = b * Ho + yo
    const int b = byo / Ho, yo = byo - b * Ho;
    if (per_image && r == 0) pre_t = sc.tmax[b];
    fetch_row<NXI, NTR, K>(pre, T, (size_t)(b * H + 2 * yo + r), cblk, C, tid);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile, 0);
  while (tile < ntiles) {
    const int cblk = tile % ncb, byo = tile / ncb;
    const int b = byo / Ho, yo = byo - b * Ho;
    const int v = tid % CH, c = cblk * CB + 2 * v, xq = tid / CH;
    Epi e;
    e.norm = per_image ? ncommon * fp16_unscale(pre_t, sc.hf) : ncommon;
    e.relu_bn = relu_bn;
    e.s0 = e.s1 = 1.f; e.h0 = e.h1 = 0.f;
    if (par_lds) {
      e.b0 = par[c]; e.b1 = par[c + 1]; e.s0 = par[kParMax + c]; e.s1 = par[kParMax + c + 1]; e.h0 = par[2 * kParMax + c]; e.h1 = par[2 * kParMax + c + 1];
    } else {
      e.b0 = bias[c]; e.b1 = bias[c + 1];
      if (relu_bn) { e.s0 = scale[c]; e.h0 = shift[c]; e.s1 = scale[c + 1]; e.h1 = shift[c + 1]; }
    }
    // one activated row -> its horizontal maxima, folded into o (max_pool SAME: a window hanging over the edge holds the cells inside the map only)
    cf o[KP];
    auto hmax = [&](bool first) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const int xo = xq + i * XP;
        if (first) o[i] = cf{0.f, 0.f};
        if (xo < Wo) {
          cf m = epi_act(e, buf[pos<NXI>(2 * xo + pad) * CH + v]);
          if (2 * xo + 1 < W) {
            const cf r = epi_act(e, buf[pos<NXI>(2 * xo + 1 + pad) * CH + v]);
            m = cf{fmaxf(m.x, r.x), fmaxf(m.y, r.y)};
          }
          o[i] = first ? m : cf{fmaxf(o[i].x, m.x), fmaxf(o[i].y, m.y)};
        }
      }
    };
    const bool two_rows = 2 * yo + 1 < H;      // (uniform over the work group)
    const int next = tile + gridDim.x;
    fill_hermitian<NXI, NTR, K>(buf, pre, tid);
    if (two_rows) fetch(tile, 1);
    else if (next < ntiles) fetch(next, 0);
    __syncthreads();
    fft<NXI, 1, CH, NTR>(buf, twi, tid);
    hmax(true);
    if (two_rows) {
      __syncthreads();      // every wave has taken its part of row a out of buf
      fill_hermitian<NXI, NTR, K>(buf, pre, tid);
      if (next < ntiles) fetch(next, 0);
      __syncthreads();
      fft<NXI, 1, CH, NTR>(buf, twi, tid);
      hmax(false);
    }
    __syncthreads();      // every wave has taken its part of the inverse row out of buf
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int xo = xq + i * XP;
      if (xo < NXO) buf[xo * CH + v] = o[i];      // (zero at and behind Wo: the next layer's padding)
    }
    __syncthreads();
    fft<NXO, -1, CH, NTR>(buf, two, tid);
    const float tm = rows_fwd_store<NXO, NTR>(buf, Tn, tid, cblk, b, yo, B, Ho, C);
    if (sc.tmax_next) wave_max_stash(tm, red);
    __syncthreads();      // every wave is done reading buf
    if (sc.tmax_next) stash_to_word<NTR>(red, sc.tmax_next + b);
    tile = next;
  }
}

// ---- conv4_fullres -> merge -> conv5.  x2 / x3: the coarse branches, NHWC fp32 [B][H2][W2][C] / [B][H3][W3][C].  Dynamic LDS: the four coarse rows of
// the tile's 64 channels, [x2 lo | x2 hi | x3 lo | x3 hi][column][32 channel pairs].  KC: complex numbers of them a thread prefetches (KC / 3 column slots per x2 row, KC / 6 per x3 row).
template <int NX, int KC>
__global__ __launch_bounds__(rows_threads<NX>()) __attribute__((amdgpu_waves_per_eu(3))) void rows_inv_merge_fwd_kernel(const cf* __restrict__ T, cf* __restrict__ Tn, const cf* __restrict__ twg, const float* __restrict__ bias,
                                                                const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, const cf* __restrict__ x2, int H2,
                                                                int W2, const cf* __restrict__ x3, int H3, int W3, int B, int H, int W, int C, int pad, float norm0, int ntiles,
                                                                float sy2, float sx2, float sy3, float sx3, Fp16Scale sc) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1, NTR = rows_threads<NX>(), K = (NXH * CH + NTR - 1) / NTR, XP = NTR / CH, KX = (NX + XP - 1) / XP;
  __shared__ cf buf[NX * CH];
  __shared__ cf tw[NX];
  __shared__ float par[3 * kParMax];
  __shared__ float red[NTR / 64];
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  cf* coarse = reinterpret_cast<cf*>(dyn_lds);
  const float ncommon = sc.tmax ? norm0 * sc.winv[0] * (sc.common ? fp16_unscale(tmax_of(sc.tmax, 0, sc.nb, 1), sc.hf) : 1.f) : norm0;
  const int tid = threadIdx.x, ncb = C / CB, CP = C >> 1;
  twiddles<NX, NTR>(tw, twg, tid);
  const bool par_lds = C <= kParMax;
  if (par_lds)
    for (int i = tid; i < C; i += NTR) {
      par[i] = bias[i];
      par[kParMax + i] = relu_bn ? scale[i] : 1.f;
      par[2 * kParMax + i] = relu_bn ? shift[i] : 0.f;
    }
  __syncthreads();      // the epilogue parameters are read at the top of the first tile, before its first barrier
  // coarse rows in LDS: [x2 lo | x2 hi][KS2 * XP columns][32 pairs], [x3 lo | x3 hi][KS3 * XP columns][32 pairs]; a thread owns column tid / 32 + j * XP of each
  constexpr int KS2 = KC / 3, KS3 = KC / 6, N2 = KS2 * XP * CH, N3 = KS3 * XP * CH;      // W2 <= KS2 * XP, W3 <= KS3 * XP (host)
  static_assert(2 * KS2 + 2 * KS3 == KC, "slots");
  float4 pre[K];
  cf prc[KC];
  float pre_t = 0.f;
  const bool per_image = sc.tmax && !sc.common;
  const int vo_c = ((tid / CH) * CP + (tid % CH)) * 8;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int cblk = tile % ncb, by = tile / ncb;
    const int b = by / H, y = by - b * H;
    if (per_image) pre_t = sc.tmax[b];
    fetch_row<NX, NTR, K>(pre, T, (size_t)by, cblk, C, tid);
    // the tile's four coarse rows: one descriptor per coarse map and image (scalar), the row and the column step scalar offsets.  Columns behind a row
    // read the next row (or the zeros of the range check) and are never used: the taps clamp at W - 1.
    const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);
    const auto d2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x2) + (size_t)b * H2 * W2 * CP + cblk * CH, 0, (H2 * W2 * CP - cblk * CH) * 8, 0x00020000);
    const auto d3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x3) + (size_t)b * H3 * W3 * CP + cblk * CH, 0, (H3 * W3 * CP - cblk * CH) * 8, 0x00020000);
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < KS2; ++j) {
      prc[j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d2, vo_c, ((ty2.lo * W2 + j * XP) * CP) * 8, 0));
      prc[KS2 + j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d2, vo_c, ((ty2.hi * W2 + j * XP) * CP) * 8, 0));
    }
#pragma unroll
    for (int j = 0; j < KS3; ++j) {
      prc[2 * KS2 + j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d3, vo_c, ((ty3.lo * W3 + j * XP) * CP) * 8, 0));
      prc[2 * KS2 + KS3 + j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d3, vo_c, ((ty3.hi * W3 + j * XP) * CP) * 8, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  while (tile < ntiles) {
    const int cblk = tile % ncb, by = tile / ncb;
    const int b = by / H, y = by - b * H;
    const int v = tid % CH, c = cblk * CB + 2 * v, xq = tid / CH;
    Epi e;
    e.norm = per_image ? ncommon * fp16_unscale(pre_t, sc.hf) : ncommon;
    e.relu_bn = relu_bn;
    e.s0 = e.s1 = 1.f; e.h0 = e.h1 = 0.f;
    if (par_lds) {
      e.b0 = par[c]; e.b1 = par[c + 1]; e.s0 = par[kParMax + c]; e.s1 = par[kParMax + c + 1]; e.h0 = par[2 * kParMax + c]; e.h1 = par[2 * kParMax + c + 1];
    } else {
      e.b0 = bias[c]; e.b1 = bias[c + 1];
      if (relu_bn) { e.s0 = scale[c]; e.h0 = shift[c]; e.s1 = scale[c + 1]; e.h1 = shift[c + 1]; }
    }
    fill_hermitian<NX, NTR, K>(buf, pre, tid);
#pragma unroll
    for (int j = 0; j < KS2; ++j) {      // (the coarse rows were last read before the barrier that closed the previous tile)
      coarse[tid + j * NTR] = prc[j];
      coarse[N2 + tid + j * NTR] = prc[KS2 + j];
    }
#pragma unroll
    for (int j = 0; j < KS3; ++j) {
      coarse[2 * N2 + tid + j * NTR] = prc[2 * KS2 + j];
      coarse[2 * N2 + N3 + tid + j * NTR] = prc[2 * KS2 + KS3 + j];
    }
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    __syncthreads();
    fft<NX, 1, CH, NTR>(buf, tw, tid);
    const float t2y = tf1_tap(y, H2, sy2).t, t3y = tf1_tap(y, H3, sy3).t;
    const cf* c2lo = coarse;
    const cf* c2hi = coarse + N2;
    const cf* c3lo = coarse + 2 * N2;
    const cf* c3hi = c3lo + N3;
    cf o[KX];
#pragma unroll
    for (int i = 0; i < KX; ++i) {
      const int x = xq + i * XP;
      o[i] = cf{0.f, 0.f};
      if (x < W) {
        const cf a = epi_act(e, buf[pos<NX>(x + pad) * CH + v]);
        const Tap t2 = tf1_tap(x, W2, sx2), t3 = tf1_tap(x, W3, sx3);
        const cf tl2 = c2lo[t2.lo * CH + v], tr2 = c2lo[t2.hi * CH + v], bl2 = c2hi[t2.lo * CH + v], br2 = c2hi[t2.hi * CH + v];
        const cf tl3 = c3lo[t3.lo * CH + v], tr3 = c3lo[t3.hi * CH + v], bl3 = c3hi[t3.lo * CH + v], br3 = c3hi[t3.hi * CH + v];
        const float u2x = lerp2(tl2.x, tr2.x, bl2.x, br2.x, t2.t, t2y), u2y = lerp2(tl2.y, tr2.y, bl2.y, br2.y, t2.t, t2y);
        const float u3x = lerp2(tl3.x, tr3.x, bl3.x, br3.x, t3.t, t3y), u3y = lerp2(tl3.y, tr3.y, bl3.y, br3.y, t3.t, t3y);
        o[i] = cf{div3((a.x + u2x) + u3x), div3((a.y + u2y) + u3y)};
      }
      if (i % 2 == 1) asm volatile("" ::: "memory");      // two pixels' taps at a time: the compiler otherwise hoists all 64 LDS reads to the front (290 registers)
    }
    __syncthreads();      // every wave has taken its part of the inverse row out of buf (and is done with the coarse rows)
#pragma unroll
    for (int i = 0; i < KX; ++i) {
      const int x = xq + i * XP;
      if (x < NX) buf[x * CH + v] = o[i];
    }
    __syncthreads();
    fft<NX, -1, CH, NTR>(buf, tw, tid);
    const float tm = rows_fwd_store<NX, NTR>(buf, Tn, tid, cblk, b, y, B, H, C);
    if (sc.tmax_next) wave_max_stash(tm, red);
    __syncthreads();      // every wave is done reading buf
    if (sc.tmax_next) stash_to_word<NTR>(red, sc.tmax_next + b);
    tile = next;
  }
}

bool cfft_rows_inv_pool_fwd_supported(int NXI, int NXO, int Cout) {
  return Cout % CB == 0 && ((NXI == 192 && NXO == 96) || (NXI == 96 && NXO == 50) || (NXI == 50 && NXO == 28));
}
constexpr int kMergeKC = 12;
bool cfft_rows_inv_merge_fwd_supported(int NX, const ConvArgs& a, const FftMerge& m) {
  if (a.Cout % CB || NX != 96 || m.H2 < 1 || m.W2 < 1 || m.H3 < 1 || m.W3 < 1 || (m.H2 == a.H && m.W2 == a.W) || (m.H3 == a.H && m.W3 == a.W)) return false;
  constexpr int XP = rows_threads<96>() / (CB / 2);      // columns a slot of the kernel's coarse-row prefetch covers
  return m.W2 <= (kMergeKC / 3) * XP && m.W3 <= (kMergeKC / 6) * XP && (size_t)m.H2 * m.W2 * a.Cout * 4 < (size_t)1 << 31;
}
// true: launched.  a: the producing layer (conv2) on its H x W map; NXO: row length of the pooled map's transform; two: twiddles of that length
bool cfft_rows_inv_pool_fwd(int NXI, int NXO, const ConvArgs& a, const cf* T, cf* Tn, const cf* twi, const cf* two, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  if (!cfft_rows_inv_pool_fwd_supported(NXI, NXO, a.Cout)) return false;
  const int Ho = (a.H + 1) / 2;
  const int ntiles = a.B * Ho * (a.Cout / CB);
#define RPF_LAUNCH(NI, NO)                                                                                                                                    \
  do {                                                                                                                                                        \
    const dim3 grid(persistent_grid(reinterpret_cast<const void*>(rows_inv_pool_fwd_kernel<NI, NO>), ntiles, rows_threads<NI>()));                             \
    hipLaunchKernelGGL((rows_inv_pool_fwd_kernel<NI, NO>), grid, dim3(rows_threads<NI>()), 0, st, T, Tn, twi, two, a.bias, a.scale, a.shift, a.relu_bn, a.B, \
                       a.H, a.W, a.Cout, pad, norm, ntiles, sc);                                                                                              \
    return true;                                                                                                                                              \
  } while (0)
  // the model's three branches (120 x 180 -> 60 x 90, 60 x 90 -> 30 x 45, 30 x 45 -> 15 x 23) and the debug-size / small-image maps of the tests
  if (NXI == 192 && NXO == 96) RPF_LAUNCH(192, 96);
  if (NXI == 96 && NXO == 50) RPF_LAUNCH(96, 50);
  if (NXI == 50 && NXO == 28) RPF_LAUNCH(50, 28);
#undef RPF_LAUNCH
  return false;
}

// true: launched.  a: the producing layer (conv4_fullres); m: the coarse branches (fp32 NHWC)
bool cfft_rows_inv_merge_fwd(int NX, const ConvArgs& a, const FftMerge& m, const cf* T, cf* Tn, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  if (!cfft_rows_inv_merge_fwd_supported(NX, a, m)) return false;
  constexpr int KC = kMergeKC;
  const int dyn = kMergeKC * rows_threads<96>() * (int)sizeof(cf);
  const int ntiles = a.B * a.H * (a.Cout / CB);
  const void* k = reinterpret_cast<const void*>(rows_inv_merge_fwd_kernel<96, KC>);
  static LdsAttr attr;      // static + dynamic LDS pass 64 KB on the model's geometry
  if (attr.ensure(k, dyn) != hipSuccess) return false;
  const dim3 grid(persistent_grid(k, ntiles, rows_threads<96>(), dyn));
  hipLaunchKernelGGL((rows_inv_merge_fwd_kernel<96, KC>), grid, dim3(rows_threads<96>()), dyn, st, T, Tn, tw, a.bias, a.scale, a.shift, a.relu_bn, static_cast<const cf*>(m.x2), m.H2, m.W2,
                     static_cast<const cf*>(m.x3), m.H3, m.W3, a.B, a.H, a.W, a.Cout, pad, norm, ntiles, (float)m.H2 / (float)a.H, (float)m.W2 / (float)a.W, (float)m.H3 / (float)a.H,
                     (float)m.W3 / (float)a.W, sc);
  return true;
}

}  // namespace cfft
}  // namespace jcm
