// Row passes of the frequency-domain convolution with the whole transform in the registers of one thread (fft_reg.h).
//
// The LDS row kernels (conv_fft_rows_fwd.hip, conv_fft_rows_inv.hip) give a work group one (image, row, 64 channels) tile: load -> barrier ->
// radix stage -> barrier -> radix stage -> barrier -> store, one butterfly per thread and stage.  Measured (round 4, SQ_INSTS_* / SQ_WAVE_CYCLES,
// DESIGN.md 4.1f): 2-3 work groups per CU (80-110 VGPRs at 6 waves per group), 2.2 resident waves per SIMD, the butterflies a fifth of the
// executed instructions -- the kernels are bound by latency behind barriers, not by HBM (2.2-3.7 TB/s).  Here a THREAD owns one (image, row,
// channel pair): it loads its half spectrum (one 8- or 16-byte load per kx; consecutive lanes are consecutive channel pairs, so every load or
// store instruction of a wave is one contiguous run), transforms it in registers with compile-time indices and literal twiddles, and stores its
// row.  No LDS, no barrier, no index arithmetic; all loads of a thread are in flight before the first butterfly.
#include "conv_fft_common.h"
#include "fft_reg.h"
#include "resize_tf1.h"

namespace jcm {
namespace cfft {
using namespace fftr;

// ---- rows, inverse + epilogue (the contract of rows_inv_kernel, conv_fft_rows_inv.hip): T'[b][y][kx][c] -> out, LAYOUT 0 = fp32 NHWC, 1 = bf16 NHWC,
// 2 = bf16 planar [B][C/8][H*W][8].  Planar: a 16-byte unit is 8 channels = the words of four threads, so the row goes through a per-wave LDS stage
// ([m][parity][32 channel pairs] words -- exactly the order in which the units are then read back, 16 bytes per lane, no barrier: one wave writes
// and reads its own stage) and leaves as 128-byte runs of 8 consecutive pixels per channel plane.
// TWO threads per (image, row, channel pair) -- a 96-point transform does not fit one thread's registers next to its loads (256 VGPRs: 178 spilled).
// Thread h of the two takes the outputs of parity h (decimation in frequency):
//   X[2 m + h] = sum_{n < M} u_h[n] w_M^(n m),   u_h[n] = (Z[n] + (-1)^h Z[n + M]) w_NX^(n h),   M = NX / 2
// i.e. one radix-2 stage whose twiddle is selected per lane, then an M-point transform in registers.  Z comes from the half spectrum of the
// channel pair (Z = Y_c + i Y_{c+1}, Hermitian extension): Z[n] and Z[n + M] = Z[NX - (M - n)] need the loaded entries n and M - n, so the
// entries are consumed in pairs (n, M - n) and both threads load all of them (the same addresses in adjacent lanes: one request).
// ALL: every output goes to `store` (the bf16 layouts stage the whole row in LDS and test the columns when they copy it out: no range test per output here)
template <int NX, int K1, bool ALL = false, class St>
__device__ __forceinline__ void inv_rows_out2(const cf (&u)[NX / 2], int h, int W, int pad, St&& store) {
  constexpr int M = NX / 2, R1 = RPlan<M>::R1, R2 = RPlan<M>::R2;
  cf o[R2];
  step2_row<M, 1, K1>(u, o);
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) {
    const int xo = 2 * (K1 + R1 * k2) + h - pad;      // output column of X[2 m + h], m = K1 + R1 k2
    if (ALL || (xo >= 0 && xo < W)) store(K1 + R1 * k2, xo, o[k2]);
  }
  __builtin_amdgcn_sched_barrier(0);      // one row of step 2 and its stores at a time (the scheduler otherwise interleaves all R1 rows and spills)
  if constexpr (K1 + 1 < R1) inv_rows_out2<NX, K1 + 1, ALL>(u, h, W, pad, store);
}
// u[N] (and u[M - N]) of thread h from the half-spectrum entries N and M - N:  q = (Ya.re, Ya.im, Yb.re, Yb.im)
template <int NX, int N>
__device__ __forceinline__ void inv_rows_in2(cf (&u)[NX / 2], const float4& q1, const float4& q2, float sg, bool odd) {
  constexpr int M = NX / 2;
  if constexpr (N == 0) {
    // Z[0] = (Ya[0].re, Yb[0].re), Z[M] = (Ya[M].re, Yb[M].re)   (DC and Nyquist are real); twiddle 1
    u[0] = cf{fmaf(sg, q2.x, q1.x), fmaf(sg, q2.z, q1.z)};
  } else {
    // entry N: Z[N] = (a.x - a.w, a.y + a.z);  entry M - N gives Z[NX - (M - N)] = Z[N + M] = (b.x + b.w, b.z - b.y)
    const cf zn = cf{q1.x - q1.w, q1.y + q1.z}, znm = cf{q2.x + q2.w, q2.z - q2.y};
    cf v = cf{fmaf(sg, znm.x, zn.x), fmaf(sg, znm.y, zn.y)};
    const float wr = odd ? Tw<N, NX>::re : 1.f, wi = odd ? Tw<N, NX>::im : 0.f;
    u[N] = cf{fmaf(-v.y, wi, v.x * wr), fmaf(v.x, wi, v.y * wr)};
    if constexpr (2 * N != M) {
      // entry M - N: Z[M - N] = (b.x - b.w, b.y + b.z);  entry N gives Z[NX - N] = Z[(M - N) + M] = (a.x + a.w, a.z - a.y)
      const cf zm = cf{q2.x - q2.w, q2.y + q2.z}, zmm = cf{q1.x + q1.w, q1.z - q1.y};
      v = cf{fmaf(sg, zmm.x, zm.x), fmaf(sg, zmm.y, zm.y)};
      const float wr2 = odd ? Tw<M - N, NX>::re : 1.f, wi2 = odd ? Tw<M - N, NX>::im : 0.f;
      u[M - N] = cf{fmaf(-v.y, wi2, v.x * wr2), fmaf(v.x, wi2, v.y * wr2)};
    }
  }
}
template <int NX, bool T16, int N, class Ld>
__device__ __forceinline__ void inv_rows_load2(cf (&u)[NX / 2], float sg, bool odd, Ld&& load) {
  constexpr int M = NX / 2;
  // the loads go out in batches of LDB pairs (a compiler fence between the batches): hoisting all NX/2+1 of them in front of the arithmetic
  // costs more registers than the thread has (measured: 390-640 bytes of scratch per lane)
  constexpr int LDB = 6;
  if constexpr (N % LDB == 0 && N > 0) __builtin_amdgcn_sched_barrier(0);
  const float4 q1 = load(N), q2 = load(M - N);
  inv_rows_in2<NX, N>(u, q1, q2, sg, odd);
  if constexpr (2 * (N + 1) <= M) inv_rows_load2<NX, T16, N + 1>(u, sg, odd, load);
}

template <int NX, int LAYOUT, bool T16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NX >= 64 ? 2 : 4, NX >= 64 ? 2 : 8))) void rows_inv_reg_kernel(const void* __restrict__ T, void* __restrict__ out, const float* __restrict__ bias,
                                                                                              const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn,
                                                                                              int nrows, int H, int W, int C, int Cout, int pad, float norm0, Fp16Scale sc,
                                                                                              int smH, int smW, int sTY, int sTX) {
  constexpr int NXH = NX / 2 + 1, M = NX / 2;
  __shared__ __attribute__((aligned(16))) unsigned stage[LAYOUT != 0 ? 4 * M * 64 : 4];      // bf16 outputs: one row (M x 2 pixels x 32 pairs) per wave
  const int CP = C >> 1;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int h = (int)(g & 1);
  const int p = (int)((g >> 1) % CP);
  // the row is the same for the 64 lanes of a wave (32 channel pairs, C % 64 == 0): scalar registers, every address a scalar base + lane offset
  const size_t by = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)((g >> 1) / CP));
  if (by >= (size_t)nrows) return;
  const int b = (int)(by / H), c = 2 * p;
  const bool odd = h != 0;
  const float sg = odd ? -1.f : 1.f;
  cf u[M];
  // Buffer loads: the row's T' is one descriptor, the entry k a SCALAR offset, the lane's channel pair the only vector offset -- no per-lane 64-bit
  // address arithmetic (a tenth of this kernel's vector instructions when the compiler forms global addresses; the kernel is bound by their issue slots)
  if constexpr (T16) {
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(static_cast<const unsigned*>(T)) + by * NXH * C, 0, NXH * C * 4, 0x00020000);
    const int vo = p * 8, ko = CP * 8;
    // the scale words of this wave's tile: one 64-channel block of one image (a wave is 32 channel pairs of one row, C % 64 == 0), contiguous over
    // kx and the same for all lanes: scalar loads
    const int nblk = C / sc.t16_cb;
    const float* ssrc = sc.t16_inv + ((size_t)__builtin_amdgcn_readfirstlane(b) * nblk + __builtin_amdgcn_readfirstlane(c / sc.t16_cb)) * NXH;
    // every load goes out first; the conversions of the first entries then overlap the latency of the later loads
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 raw[NXH];
#pragma unroll
    for (int n = 0; 2 * n <= M; ++n) {      // in the order of use: the entries are consumed in pairs (n, M - n)
      raw[n] = __builtin_amdgcn_raw_buffer_load_b64(d, vo, n * ko, 0);
      if (2 * n != M) raw[M - n] = __builtin_amdgcn_raw_buffer_load_b64(d, vo, (M - n) * ko, 0);
    }
    auto load = [&](int k) __attribute__((always_inline)) {
      const float s = ssrc[k];
      const cf ya = unpack_h2_mix_s(raw[k][0], s), yb = unpack_h2_mix_s(raw[k][1], s);
      return make_float4(ya.x, ya.y, yb.x, yb.y);
    };
    inv_rows_load2<NX, T16, 0>(u, sg, odd, load);
  } else {
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(static_cast<const float*>(T)) + by * NXH * C * 2, 0, NXH * C * 8, 0x00020000);
    const int vo = p * 16, ko = CP * 16;
    auto load = [&](int k) __attribute__((always_inline)) {      // (Ya.re, Ya.im, Yb.re, Yb.im)
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(d, vo, k * ko, 0));
      return make_float4(q[0], q[1], q[2], q[3]);
    };
    inv_rows_load2<NX, T16, 0>(u, sg, odd, load);      // (in batches of LDB pairs: 49 x 16 bytes in flight at once do not fit the registers)
  }
  float norm = norm0;
  if (sc.tmax) {
    float tm;
    if (sc.common) {      // one scale for the tensor (a handle with training state): the largest of its nb words, taken by the wave, not by every thread
      tm = 0.f;
      for (int i = (int)(threadIdx.x & 63); i < sc.nb; i += 64) tm = fmaxf(tm, sc.tmax[i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o));
    } else {
      tm = sc.tmax[b];
    }
    norm = norm0 * sc.winv[0] * fp16_unscale(tm, sc.hf);      // powers of two: exact
  }
  const bool two = c + 1 < Cout;
  float b0v = 0.f, b1v = 0.f, s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
  if (c < Cout) {
    b0v = bias[c];
    if (two) b1v = bias[c + 1];
    if (relu_bn) { s0 = scale[c]; h0 = shift[c]; if (two) { s1 = scale[c + 1]; h1 = shift[c + 1]; } }
  }
  step1<M, 1>(u);
  const int lane = threadIdx.x & 63;
  // scatter of overlap-save windows (sTX > 0; scalars: the row is the wave's): row `by` = valid row yv of window bw = (image, ty, tx)
  int sc_row = -1, sc_x0 = 0;
  if (LAYOUT == 0 && sTX > 0) {
    const int bw = (int)(by / H), yv = (int)(by % H);
    const int tx = bw % sTX, ty = (bw / sTX) % sTY, bi = bw / (sTX * sTY), ym = ty * H + yv;
    sc_row = ym < smH ? bi * smH + ym : -1;
    sc_x0 = tx * W;
  }
  unsigned* wst = stage + (LAYOUT != 0 ? (threadIdx.x >> 6) * (M * 64) : 0);
  auto store = [&](int m, int xo, cf z) __attribute__((always_inline)) {
    float v0 = fmaf(z.x, norm, b0v), v1 = fmaf(z.y, norm, b1v);      // (single roundings: this kernel is instruction-bound)
    if (relu_bn) { v0 = fmaf(fmaxf(v0, 0.f), s0, h0); v1 = fmaf(fmaxf(v1, 0.f), s1, h1); }
    if constexpr (LAYOUT == 0) {
      if (sTX > 0) {      // overlap-save windows: the valid region goes straight to its place in the map (the rows / columns of the last windows that hang over the map are dropped)
        const int xm = sc_x0 + xo;
        if (sc_row >= 0 && xm < smW) st_stream(reinterpret_cast<cf*>(static_cast<float*>(out) + ((size_t)sc_row * smW + xm) * Cout + c), cf{v0, v1});
      } else {
        st_stream(reinterpret_cast<cf*>(static_cast<float*>(out) + (by * W + xo) * Cout + c), cf{v0, v1});
      }
    } else {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      wst[(m * 2 + h) * 32 + (lane >> 1)] = __builtin_bit_cast(unsigned, bf16x2{static_cast<__bf16>(v0), static_cast<__bf16>(v1)});
    }
  };
  if (two || LAYOUT != 0) inv_rows_out2<NX, 0, LAYOUT != 0>(u, h, W, pad, store);      // (the launcher takes even channel counts only; bf16: whole 8-channel units)
  if constexpr (LAYOUT != 0) {
    // a wave = 32 channel pairs = eight 8-channel units of ONE row (C % 64 == 0); unit q = (m, parity, unit) lies at word 4 q of the stage
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int y = (int)(by % H);
    const int c0 = c - (lane >> 1) * 2;      // first channel of this wave
    const uint4* rd = reinterpret_cast<const uint4*>(wst);
#pragma unroll 4
    for (int q = lane; q < M * 16; q += 64) {
      const int m = q >> 4, hh = (q >> 3) & 1, j = q & 7;
      const int xo = 2 * m + hh - pad, cu = c0 + 8 * j;
      if (xo >= 0 && xo < W && cu < Cout) {
        __bf16* o = LAYOUT == 1 ? static_cast<__bf16*>(out) + (by * W + xo) * Cout + cu      // NHWC: the wave's 8 units of a pixel are one 128-byte run
                                : static_cast<__bf16*>(out) + (((size_t)b * (Cout >> 3) + (cu >> 3)) * H * W + (size_t)y * W + xo) * 8;      // planar: 8 consecutive pixels of a plane
        *reinterpret_cast<uint4*>(o) = rd[q];
      }
    }
  }
}

// ---- columns, inverse (the contract of cols_inv_kernel, conv_fft_cols.hip): Yf[ky][kx][b][ldy] -> T'[b][y][kx][c], y < H = row y + pad of the circular
// convolution.  ONE thread per (image, kx, channel): a 64-point transform is 128 registers.  Lanes are consecutive channels: every load is a 512-byte
// run of Yf, every store a 256- / 512-byte run of T'.  T16: the block-floating-point scale of the (image, kx, 64 channels) tile is the WAVE's maximum.
template <int NY, int K1>
__device__ __forceinline__ void inv_cols_rows(cf (&x)[NY]) {      // step 2 in place: x[R2 K1 + k2] <- X[K1 + R1 k2]
  constexpr int R1 = RPlan<NY>::R1, R2 = RPlan<NY>::R2;
  cf o[R2];
  step2_row<NY, 1, K1>(x, o);
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) x[R2 * K1 + k2] = o[k2];
  if constexpr (K1 + 1 < R1) inv_cols_rows<NY, K1 + 1>(x);
}
// T16 (bf16 handles, 16-bit intermediates): Yf holds complex FP16 = product * 2^-k (cgemm_split.hip, Y16) and T' is written as complex fp16 in block floating
// point; yinv = 2^k rides in the tile's scale word, so nothing is multiplied here.  A complex number is 4 bytes then: the two lanes of adjacent channels
// (c, c + 1) share their accesses -- the even lane fetches (c, c + 1) of the even ky and stores both channels of the even output rows, the odd lane the
// odd ones, 8 bytes per lane, and they swap the halves they fetched for each other (DPP) -- half the memory instructions of one 4-byte access per lane.
__device__ __forceinline__ unsigned lane_pair_swap(unsigned w) { return (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true); }      // quad_perm [1,0,3,2]
template <int NY, bool T16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NY >= 64 ? 2 : 3, NY >= 64 ? 2 : 8))) void cols_inv_reg_kernel(const cf* __restrict__ Yf, void* __restrict__ T, int B, int H, int NXH, int C, int ldy,
                                                                                              int pad, float* __restrict__ t16, float yinv) {
  static_assert(NY % 2 == 0, "ky pairs");
  constexpr int R1 = RPlan<NY>::R1, R2 = RPlan<NY>::R2;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(g % C);
  // (image, kx) are the same for the 64 lanes of a wave (C % 64 == 0): scalar registers, so that every address below is a scalar base + lane offset
  const unsigned bk = (unsigned)__builtin_amdgcn_readfirstlane((int)(g / C));
  const int kx = (int)(bk % (unsigned)NXH), b = (int)(bk / (unsigned)NXH);
  if (b >= B) return;
  const bool odd = (c & 1) != 0;      // = the lane's parity (C is even)
  cf x[NY];
  if constexpr (T16) {
    const uint2* src = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned*>(Yf) + ((size_t)kx * NY * B + b) * ldy + (c & ~1));      // ldy is even
    const size_t kstep = (size_t)B * ldy / 2;
    uint2 raw[NY / 2];      // every load goes out before the first conversion (left to itself the compiler waits for each load in turn)
#pragma unroll
    for (int i = 0; i < NY / 2; ++i) raw[i] = src[(size_t)(2 * i + (odd ? 1 : 0)) * kstep];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NY / 2; ++i) {
      const unsigned keep = odd ? raw[i].y : raw[i].x, recv = lane_pair_swap(odd ? raw[i].x : raw[i].y);
      x[2 * i] = unpack_h2(odd ? recv : keep, 1.f);
      x[2 * i + 1] = unpack_h2(odd ? keep : recv, 1.f);
    }
  } else {
    const cf* src = Yf + ((size_t)kx * NY * B + b) * ldy + c;
#pragma unroll
    for (int ky = 0; ky < NY; ++ky) x[ky] = src[(size_t)ky * B * ldy];
  }
  step1<NY, 1>(x);
  inv_cols_rows<NY, 0>(x);
  if constexpr (T16) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int y = (i / R2) + R1 * (i % R2) - pad;      // x[i] = X[i / R2 + R1 (i % R2)]
      if (y >= 0 && y < H) m = fmaxf(m, fmaxf(fabsf(x[i].x), fabsf(x[i].y)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float s = bfp_scale(m);
    if ((threadIdx.x & 63) == 0) t16[((size_t)b * (C >> 6) + (c >> 6)) * NXH + kx] = (1.0f / s) * yinv;      // powers of two: exact
    uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(T) + ((size_t)b * H * NXH + kx) * C + (c & ~1));
    const size_t ystep = (size_t)NXH * C / 2;
#pragma unroll
    for (int k = 0; k < NY; k += 2) {      // output samples X[k], X[k + 1] = rows k - pad, k + 1 - pad: the even lane stores the first, the odd lane the second
      const int ia = R2 * (k % R1) + k / R1, ib = R2 * ((k + 1) % R1) + (k + 1) / R1;
      const unsigned va = pack_h2(x[ia].x * s, x[ia].y * s), vb = pack_h2(x[ib].x * s, x[ib].y * s);
      const unsigned recv = lane_pair_swap(odd ? va : vb);
      const int y = k - pad + (odd ? 1 : 0);
      if (y >= 0 && y < H) st_stream(dst + (size_t)y * ystep, make_uint2(odd ? recv : va, odd ? vb : recv));
    }
  } else {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int y = (i / R2) + R1 * (i % R2) - pad;
      if (y >= 0 && y < H) st_stream(reinterpret_cast<cf*>(T) + ((size_t)(b * H + y) * NXH + kx) * C + c, x[i]);
    }
  }
}
// true: launched (64-point columns, 64-channel tiles)
bool cfft_cols_inv_reg(int NY, const ConvArgs& a, const cf* Yf, cf* T, int NXH, int ldy, int pad, hipStream_t st, float* t16, float y16_inv) {
  if (a.CoutP % 64 || ((y16_inv != 0.f) != (t16 != nullptr)) || (t16 && ldy % 2)) return false;      // 16-bit T' comes with fp16 product spectra
  const size_t threads = (size_t)a.B * NXH * a.CoutP;
  const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
#define CI_LAUNCH(N)                                                                                                                                             \
  do {                                                                                                                                                           \
    if (t16) hipLaunchKernelGGL((cols_inv_reg_kernel<N, true>), grid, blk, 0, st, Yf, static_cast<void*>(T), a.B, a.H, NXH, a.CoutP, ldy, pad, t16, y16_inv);            \
    else hipLaunchKernelGGL((cols_inv_reg_kernel<N, false>), grid, blk, 0, st, Yf, static_cast<void*>(T), a.B, a.H, NXH, a.CoutP, ldy, pad, nullptr, 0.f);          \
  } while (0)
  switch (NY) {      // 64: the 60 x 90 maps; 36 / 20: the half- and quarter-resolution branches; 32: the training step's overlap-save windows
    case 64: CI_LAUNCH(64); break;
    case 36: CI_LAUNCH(36); break;
    case 32: CI_LAUNCH(32); break;
    case 20: CI_LAUNCH(20); break;
    default: return false;
  }
#undef CI_LAUNCH
  return true;
}

// ---- rows, forward (the contract of rows_fwd_kernel<NX, 1, T16 = true>, conv_fft_rows_fwd.hip): bf16 NHWC -> T16[kx][c/16][b][y][16], complex fp16 in block floating
// point, one scale per (image, row, 64 channels) tile.  Two threads per channel pair again; the forward transform is decimation in frequency as well, so thread h
// produces the outputs of parity h:  X[2 m + h] = sum_{j < M} u_h[j] w_M^(j m),  u_h[j] = (z[j] + (-1)^h z[j + M]) w_NX^(j h),  z = x_c + i x_{c+1}.
// Thread h LOADS pixels [h M, h M + M) only and swaps words with the other thread of the pair: each pixel is fetched once.  The Hermitian split into the two channels'
// spectra pairs X[k] with X[NX - k], which has the parity of k: both live in the same thread (at a lane-selected register).  A wave is the 32 channel pairs =
// 64 channels of one row = one block-floating-point tile: its scale is the wave's maximum (six shuffles, no barrier).
// The forward kernels put thread 0 of a pair in lanes 0..31 and thread 1 in lanes 32..63 of the wave (the inverse kernels above: adjacent lanes): each half
// wave then reads one contiguous 128-byte run per pixel instead of two runs interleaved lane by lane -- nothing for the plain row pass, 1.51 -> 1.15 ms for
// the merge below with its 126 loads per thread.  The word of the other thread of the pair comes by v_permlane32_swap (lanes i and i + 32).
__device__ __forceinline__ unsigned pair_word(unsigned w, bool odd) {
  const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return odd ? r[0] : r[1];
}
// (thread h of the pair, channel pair p, row by) of a thread
__device__ __forceinline__ void pair_coords(int CP, int& h, int& p, size_t& by) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const size_t q = (g >> 6) * 32;      // first pair of the wave
  h = lane >> 5;
  p = (int)((q + (lane & 31)) % CP);
  by = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(q / CP));      // the row: the same for the 64 lanes (C % 64 == 0)
}
template <int NX, int J>
__device__ __forceinline__ void fwd_rows_in2(cf (&u)[NX / 2], const unsigned (&raw)[NX / 2], float sg, bool odd) {
  constexpr int M = NX / 2;
  const unsigned oth = pair_word(raw[J], odd);
  const cf a = bf16pair(raw[J]), o = bf16pair(oth);
  cf v = cf{fmaf(sg, a.x, o.x), fmaf(sg, a.y, o.y)};      // h = 0: z[J] = a, z[J + M] = o -> a + o;  h = 1: z[J] = o, z[J + M] = a -> o - a
  if constexpr (J > 0) {
    const float wr = odd ? Tw<-J, NX>::re : 1.f, wi = odd ? Tw<-J, NX>::im : 0.f;      // e^{-2 pi i J / NX} for the odd outputs
    v = cf{fmaf(-v.y, wi, v.x * wr), fmaf(v.x, wi, v.y * wr)};
  }
  u[J] = v;
  if constexpr (J + 1 < M) fwd_rows_in2<NX, J + 1>(u, raw, sg, odd);
}
template <int N, int S, int K1>
__device__ __forceinline__ void step2_inplace(cf (&x)[N]) {      // x[R2 K1 + k2] <- X[K1 + R1 k2]
  constexpr int R1 = RPlan<N>::R1, R2 = RPlan<N>::R2;
  cf o[R2];
  step2_row<N, S, K1>(x, o);
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) x[R2 * K1 + k2] = o[k2];
  if constexpr (K1 + 1 < R1) step2_inplace<N, S, K1 + 1>(x);
}
// visit the outputs k = 2 m + h <= NX / 2 of this thread: f(m, X_c[k], X_{c+1}[k]) as (re, im, re, im)
template <int NX, int MI, class F>
__device__ __forceinline__ void fwd_rows_visit(const cf (&u)[NX / 2], bool odd, F&& f) {
  constexpr int M = NX / 2, R1 = RPlan<M>::R1, R2 = RPlan<M>::R2;
  constexpr int m0 = (M - MI) % M, m1 = M - 1 - MI;      // X[NX - k] = X_h[m0] (h = 0) or X_h[m1] (h = 1)
  const cf zk = u[R2 * (MI % R1) + MI / R1], za = u[R2 * (m0 % R1) + m0 / R1], zb = u[R2 * (m1 % R1) + m1 / R1];
  const cf zn = cf{odd ? zb.x : za.x, odd ? zb.y : za.y};
  f(MI, make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)));
  if constexpr (2 * (MI + 1) <= M) fwd_rows_visit<NX, MI + 1>(u, odd, f);
}
// the transform, the block-floating-point scale and the stores of one row whose pixels sit in raw[] as bf16 pairs (thread h: pixels [h M, h M + M))
template <int NX>
__device__ __forceinline__ void fwd_rows_finish(const unsigned (&raw)[NX / 2], uint2* __restrict__ T, int h, int p, int b, int y, int B, int H, int C, float* __restrict__ tmax,
                                                float* __restrict__ t16) {
  constexpr int M = NX / 2;
  const bool odd = h != 0;
  const int lane = threadIdx.x & 63;
  cf u[M];
  fwd_rows_in2<NX, 0>(u, raw, odd ? -1.f : 1.f, odd);
  step1<M, -1>(u);
  step2_inplace<M, -1, 0>(u);
  // the tile's largest |component| (k = 2 m + h <= NX / 2 only: the thread of odd parity has one output less)
  float m = 0.f;
  fwd_rows_visit<NX, 0>(u, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
    if (2 * mi + h <= M) m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  });
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
  const float sc = bfp_scale(m);
  const int cblk = p >> 5, v = p & 31;
  if (lane == 0) {
    t16[((size_t)b * (C >> 6) + cblk) * H + y] = 1.0f / sc;
    if (tmax && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(tmax + b), __float_as_uint(m));      // the image's word of the spectra's scale (values >= 0 order like unsigned)
  }
  uint2* dst = T + ((((size_t)cblk * 4 + (v >> 3)) * B + b) * H + y) * 8 + (v & 7);      // t_fwd_index(k = 0); per kx: + (C / 16) B H 8
  const size_t kstride = (size_t)(C >> 4) * B * H * 8;
  fwd_rows_visit<NX, 0>(u, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
    const int k = 2 * mi + h;
    if (k <= M) dst[(size_t)k * kstride] = make_uint2(pack_h2(o.x * sc, o.y * sc), pack_h2(o.z * sc, o.w * sc));
  });
}
template <int NX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NX >= 96 ? 3 : 4, NX >= 96 ? 3 : 8))) void rows_fwd_reg_kernel(const unsigned* __restrict__ in, uint2* __restrict__ T, int nrows, int B, int H, int W, int C,
                                                                                              float* __restrict__ tmax, float* __restrict__ t16) {
  constexpr int M = NX / 2;
  const int CP = C >> 1;
  int h, p;
  size_t by;
  pair_coords(CP, h, p, by);
  if (by >= (size_t)nrows) return;
  const int b = (int)(by / H), y = (int)(by % H);
  const unsigned* src = in + (by * W) * CP + p;      // a word = channels (2 p, 2 p + 1) of a pixel
  unsigned raw[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int n = h * M + j;
    raw[j] = n < W ? src[(size_t)n * CP] : 0u;
  }
  fwd_rows_finish<NX>(raw, T, h, p, b, y, B, H, C, tmax, t16);
}
// true: launched (96- / 50- / 28-point rows of a bf16 NHWC tensor -- the 60 x 90, 30 x 45 and 15 x 23 maps --, 16-bit T; W <= NX)
bool cfft_rows_fwd_reg(int NX, const ConvArgs& a, int layout, cf* T, float* tmax, hipStream_t st, float* t16) {
  if ((NX != 96 && NX != 50 && NX != 28) || layout != 1 || !t16 || a.Cin % 64 || a.W > NX) return false;
  const int nrows = a.B * a.H;
  const size_t threads = (size_t)nrows * a.Cin;      // two threads per channel pair
  const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
#define RFR_LAUNCH(N) hipLaunchKernelGGL(rows_fwd_reg_kernel<N>, grid, blk, 0, st, static_cast<const unsigned*>(a.x), reinterpret_cast<uint2*>(T), nrows, a.B, a.H, a.W, a.Cin, tmax, t16)
  if (NX == 96) RFR_LAUNCH(96);
  else if (NX == 50) RFR_LAUNCH(50);      // (round 6: the half- and quarter-resolution branches left the LDS kernel too)
  else RFR_LAUNCH(28);
#undef RFR_LAUNCH
  return true;
}

// ---- rows, forward, of OVERLAP-SAVE WINDOWS read straight from the map they are cut from (fp32 handles, the training step's 32 x 32 windows; the contract of
// window_gather_kernel + rows_fwd_kernel<32, 0>): ONE thread per (window row, channel pair) -- a 32-point transform is 64 registers.  A wave is 64 channel
// pairs of one window row (Cin % 128 == 0), so the window, its row and the validity of each of its 32 pixels are scalars: a pixel inside the map is one buffer
// load (the map row is the descriptor, the column a scalar offset, the lane's channel pair the vector offset), a pixel outside it -- or in the 4-pixel halo
// when the caller wants the valid region only (the weight gradient's dZ) -- a literal zero.  The gathered window tensor (0.4 GB per conv5 pass at 16 images)
// is never written or read.
template <int NX>
__global__ __launch_bounds__(256) void rows_fwd_win_reg_kernel(const float* __restrict__ map, float4* __restrict__ T, int nrows, int BW, int H, int W, int C, int TY, int TX,
                                                               int valid_only, float* __restrict__ tmax) {
  constexpr int NXH = NX / 2 + 1, R1 = RPlan<NX>::R1, R2 = RPlan<NX>::R2, V = NX - 8;
  const int CP = C >> 1;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int p = (int)(g % CP);
  const int row = __builtin_amdgcn_readfirstlane((int)(g / CP));      // (window, wy): the same for the 64 lanes (C % 128 == 0)
  if (row >= nrows) return;
  const int wy = row % NX, bw = row / NX;
  const int tx = bw % TX, ty = (bw / TX) % TY, b = bw / (TX * TY);
  const int y = ty * V - 4 + wy, x0 = tx * V - 4;
  const bool row_in = y >= 0 && y < H && !(valid_only && (wy < 4 || wy >= NX - 4));
  cf z[NX];
  {
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(map) + ((size_t)b * H + (row_in ? y : 0)) * W * C, 0, W * C * 4, 0x00020000);
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int wx = 0; wx < NX; ++wx) {
      const int x = x0 + wx;
      const bool in = row_in && x >= 0 && x < W && !(valid_only && (wx < 4 || wx >= NX - 4));      // wave-uniform
      z[wx] = cf{0.f, 0.f};
      if (in) z[wx] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d, p * 8, x * C * 4, 0));
    }
  }
  step1<NX, -1>(z);
  step2_inplace<NX, -1, 0>(z);      // z[R2 (k % R1) + k / R1] = Z[k]
  // Z = FFT(x_c + i x_{c+1}):  X_c[k] = (Z[k] + conj Z[-k]) / 2,  X_{c+1}[k] = (Z[k] - conj Z[-k]) / (2i)  ->  T[kx][c/16][window][wy][16]
  const int cblk = p >> 5, v = p & 31;
  float4* dst = T + ((((size_t)cblk * 4 + (v >> 3)) * BW + bw) * NX + wy) * 8 + (v & 7);      // t_fwd_index(k = 0); per kx: + (C / 16) BW NX 8
  const size_t kstride = (size_t)(C >> 4) * BW * NX * 8;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < NXH; ++k) {
    const int kn = k == 0 ? 0 : NX - k;
    const cf zk = z[R2 * (k % R1) + k / R1], zn = z[R2 * (kn % R1) + kn / R1];
    const float4 o = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    dst[(size_t)k * kstride] = o;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (tmax) {      // the window's word of the spectra's scale (values >= 0 order like unsigned)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(tmax + bw), __float_as_uint(m));
  }
}
bool cfft_rows_fwd_win_reg_supported(int NX, int Cin) { return NX == 32 && Cin % 128 == 0; }
bool cfft_rows_fwd_win_reg(int NX, const ConvArgs& a, cf* T, float* tmax, hipStream_t st) {
  if (!a.win_map || !cfft_rows_fwd_win_reg_supported(NX, a.Cin) || a.H != NX || a.W != NX || a.B != a.win_B * a.win_TY * a.win_TX) return false;
  if ((size_t)a.win_W * a.Cin * 4 >= (size_t)1 << 31) return false;
  const int nrows = a.B * NX;
  const size_t threads = (size_t)nrows * (a.Cin / 2);
  hipLaunchKernelGGL(rows_fwd_win_reg_kernel<32>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(a.win_map), reinterpret_cast<float4*>(T), nrows, a.B,
                     a.win_H, a.win_W, a.Cin, a.win_TY, a.win_TX, a.win_valid_only, tmax);
  return true;
}

// ---- rows, forward, of the MERGED map x = ((x1 + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70) of a bf16 handle, for the model's own geometry
// (W x 2 W2 maps: 90 / 45 / 23 columns): the contract of rows_fwd_merge_kernel<NX, true, true> (conv_fft_rows_fwd.hip), whose generic taps -- eight
// gathers, two tap computations and twelve lerps per element -- make it the one transform pass bound by vector-ALU issue slots (1.5 ms per 256 images at
// 2.2 TB/s).  Here the row lives in the registers of its two threads, so the TF-1.x taps along x are COMPILE-TIME constants (UpTaps: the same float32
// products tf1_tap() forms) and a thread fetches each coarse pixel it needs once: 48 + 2 x 25 + 2 x 14 words instead of 9 per element.  The lerp along y
// is taken first, on the few coarse pixels, then the lerp along x per fine pixel (the other order in the generic kernel and in TF: the two differ in the
// last fp32 bit, four orders of magnitude below the bf16 rounding that follows).  The half row of thread 1 sees other x3 taps than thread 0's: both
// candidates are compile-time registers, one v_cndmask picks.
template <int W, int WC> struct UpTaps {      // tf1_tap(x, WC, (float)WC / (float)W) at compile time
  static constexpr float scale = (float)WC / (float)W;
  static constexpr int lo(int x) { return (int)((float)x * scale); }
  static constexpr int hi(int x) { return lo(x) + 1 < WC ? lo(x) + 1 : WC - 1; }
  static constexpr float t(int x) { return (float)x * scale - (float)lo(x); }
};
template <int NX, int W, int W3> struct MergeGeom {
  static constexpr int M = NX / 2;
  using T3 = UpTaps<W, W3>;
  static constexpr int base3(int h) { return T3::lo(h * M); }
  static constexpr int span3(int h) { return T3::hi((h + 1) * M - 1 < W ? (h + 1) * M - 1 : W - 1) - base3(h) + 1; }
  static constexpr int N3 = span3(0) > span3(1) ? span3(0) : span3(1);      // x3 pixels a thread fetches per source row
  static constexpr int N2 = M / 2 + 1;                                        // x2 pixels (W = 2 W2: pixel j of either thread lerps locals j / 2 and j / 2 + 1)
};
__device__ __forceinline__ cf lerp_cf(cf a, cf b, float t) { return cf{fmaf(b.x - a.x, t, a.x), fmaf(b.y - a.y, t, a.y)}; }
template <int NX, int W, int W3, int J>
__device__ __forceinline__ void merge_px(unsigned (&raw)[NX / 2], const unsigned (&r1)[NX / 2], const cf (&v2)[MergeGeom<NX, W, W3>::N2], const cf (&v3)[MergeGeom<NX, W, W3>::N3], bool odd) {
  using G = MergeGeom<NX, W, W3>;
  using T3 = typename G::T3;
  constexpr int M = NX / 2;
  constexpr int n1 = M + J < W ? M + J : W - 1;      // thread 1's pixel (clamped behind the map: that word becomes 0 below)
  // 2 W2 = W: source position n / 2 for both threads (M is even), i.e. locals J / 2 and J / 2 + 1 with weight 0 or 1/2
  cf u2 = v2[J >> 1];
  if constexpr (J & 1) {
    constexpr int hi2_1 = M / 2 + (J >> 1) + 1 < W / 2 ? (J >> 1) + 1 : W / 2 - 1 - M / 2;      // thread 1: hi = min(lo + 1, W2 - 1), as a local index
    const cf h2 = hi2_1 == (J >> 1) + 1 ? v2[(J >> 1) + 1] : cf{odd ? v2[hi2_1].x : v2[(J >> 1) + 1].x, odd ? v2[hi2_1].y : v2[(J >> 1) + 1].y};
    u2 = lerp_cf(u2, h2, 0.5f);
  }
  constexpr int lo0 = T3::lo(J) - G::base3(0), hi0 = T3::hi(J) - G::base3(0), lo1 = T3::lo(n1) - G::base3(1), hi1 = T3::hi(n1) - G::base3(1);
  static_assert(lo0 >= 0 && hi0 < G::N3 && lo1 >= 0 && hi1 < G::N3, "x3 taps inside the fetched span");
  const cf a3 = cf{odd ? v3[lo1].x : v3[lo0].x, odd ? v3[lo1].y : v3[lo0].y}, b3 = cf{odd ? v3[hi1].x : v3[hi0].x, odd ? v3[hi1].y : v3[hi0].y};
  const cf u3 = lerp_cf(a3, b3, odd ? T3::t(n1) : T3::t(J));
  const cf a = bf16pair(r1[J]);
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  // (the third as one multiplication by RN(1/3): against the exact quotient it moves the bf16 rounding of one value in 2^15 -- the class of the lerp order)
  constexpr float k3 = 0.333333343267440796f;
  const unsigned w = __builtin_bit_cast(unsigned, bf16x2{static_cast<__bf16>(((a.x + u2.x) + u3.x) * k3), static_cast<__bf16>(((a.y + u2.y) + u3.y) * k3)});
  raw[J] = (M + J < W || !odd) ? w : 0u;
  if constexpr (J % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // a few pixels at a time: interleaving all of them costs more registers than the thread has
  if constexpr (J + 1 < M) merge_px<NX, W, W3, J + 1>(raw, r1, v2, v3, odd);
}
template <int NX, int W, int W3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rows_fwd_merge_reg_kernel(const unsigned* __restrict__ x1, const unsigned* __restrict__ x2, int H2,
                                                                                                    const unsigned* __restrict__ x3, int H3, uint2* __restrict__ T, int nrows, int B, int H,
                                                                                                    int C, float sy2, float sy3, float* __restrict__ tmax, float* __restrict__ t16) {
  using G = MergeGeom<NX, W, W3>;
  constexpr int M = NX / 2, W2 = W / 2, N2 = G::N2, N3 = G::N3;
  static_assert(W == 2 * W2 && M % 2 == 0 && M <= W && W <= NX, "the x2 taps above");
  const int CP = C >> 1;
  int h, p;
  size_t by;
  if (C == 512) {
    // Eight 64-channel blocks per row: work group i runs on XCD i % 8, so let it be block i % 8 of FOUR rows (one per wave) instead of four blocks of one
    // row -- an XCD then sees one channel block of every row, and the coarse rows, which 2-8 fine rows share, stay in its L2 (counters: 3.27 -> 1.9 GB fetched)
    const int lane = threadIdx.x & 63;
    h = lane >> 5;
    p = (int)(blockIdx.x & 7) * 32 + (lane & 31);
    by = (size_t)(blockIdx.x >> 3) * 4 + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  } else {
    pair_coords(CP, h, p, by);
  }
  if (by >= (size_t)nrows) return;
  const int b = (int)(by / H), y = (int)(by % H);
  const bool odd = h != 0;
  const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);      // the source rows: the same for the whole wave
  unsigned r1[M], r2a[N2], r2b[N2], r3a[N3], r3b[N3];
  {
    // Buffer loads: ONE lane offset per source (channel pair + the half row's first column) and a SCALAR offset per pixel -- no vector address arithmetic
    // and no address registers.  A descriptor covers one image; what a half row reads behind its row (thread 1, j >= W - M; the spare coarse columns) is
    // the next row or, behind the image, the zeros of the range check, and is never used: the taps below are clamped at compile time.
    const int bs = __builtin_amdgcn_readfirstlane(b), ys = __builtin_amdgcn_readfirstlane(y);
    const auto d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(x1) + (size_t)bs * H * W * CP, 0, H * W * CP * 4, 0x00020000);
    const auto d2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(x2) + (size_t)bs * H2 * W2 * CP, 0, H2 * W2 * CP * 4, 0x00020000);
    const auto d3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(x3) + (size_t)bs * H3 * W3 * CP, 0, H3 * W3 * CP * 4, 0x00020000);
    const int cp4 = CP * 4;
    const int o2 = (p + h * (M / 2) * CP) * 4, o3 = (p + (odd ? G::base3(1) : G::base3(0)) * CP) * 4, o1 = (p + h * M * CP) * 4;
    const int s2a = ty2.lo * W2 * cp4, s2b = ty2.hi * W2 * cp4, s3a = ty3.lo * W3 * cp4, s3b = ty3.hi * W3 * cp4, s1 = ys * W * cp4;
    // the coarse rows first: their conversions and the lerps along y run while the row of x1 is still on its way
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      r2a[i] = __builtin_amdgcn_raw_buffer_load_b32(d2, o2, s2a + i * cp4, 0);
      r2b[i] = __builtin_amdgcn_raw_buffer_load_b32(d2, o2, s2b + i * cp4, 0);
    }
#pragma unroll
    for (int i = 0; i < N3; ++i) {
      r3a[i] = __builtin_amdgcn_raw_buffer_load_b32(d3, o3, s3a + i * cp4, 0);
      r3b[i] = __builtin_amdgcn_raw_buffer_load_b32(d3, o3, s3b + i * cp4, 0);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) r1[j] = __builtin_amdgcn_raw_buffer_load_b32(d1, o1, s1 + j * cp4, 0);
  }
  __builtin_amdgcn_sched_barrier(0);      // every load is out before the first conversion
  cf v2[N2], v3[N3];
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    v2[i] = lerp_cf(bf16pair(r2a[i]), bf16pair(r2b[i]), ty2.t);
    if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < N3; ++i) {
    v3[i] = lerp_cf(bf16pair(r3a[i]), bf16pair(r3b[i]), ty3.t);
    if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned raw[M];
  merge_px<NX, W, W3, 0>(raw, r1, v2, v3, odd);
  fwd_rows_finish<NX>(raw, T, h, p, b, y, B, H, C, tmax, t16);
}
// true: launched (the model's merge on a bf16 handle with 16-bit T: 90-column maps, x2 at half and x3 at a quarter of the width)
bool cfft_rows_fwd_merge_reg(int NX, const ConvArgs& a, const FftMerge& m, int in_layout, cf* T, float* tmax, hipStream_t st, float* t16) {
  if (NX != 96 || in_layout != 1 || !t16 || a.Cin % 64 || a.W != 90 || m.W2 != 45 || m.W3 != 23 || m.H2 < 1 || m.H3 < 1) return false;
  const int nrows = a.B * a.H;
  const size_t threads = a.Cin == 512 ? (size_t)((nrows + 3) / 4) * 8 * 256 : (size_t)nrows * a.Cin;      // (512 channels: eight work groups per four rows)
  hipLaunchKernelGGL((rows_fwd_merge_reg_kernel<96, 90, 23>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned*>(a.x), static_cast<const unsigned*>(m.x2), m.H2,
                     static_cast<const unsigned*>(m.x3), m.H3, reinterpret_cast<uint2*>(T), nrows, a.B, a.H, a.Cin, (float)m.H2 / (float)a.H, (float)m.H3 / (float)a.H, tmax, t16);
  return true;
}

// ---- rows, inverse of layer L + epilogue + rows, forward of layer L+1 (the contract of rows_inv_fwd_kernel, conv_fft_rows_inv.hip; fp32 handles): the activation
// between two frequency-domain layers never goes to HBM.  Two threads per channel pair.  The inverse (decimation in frequency) leaves thread h with the pixels of
// parity h (the kernel's pad is even).  The forward transform wants u_h[j] = (z[j] + (-1)^h z[j + M]) w^(j h), j < M = NX / 2: pixels j and j + M have the same
// parity, so the thread of parity j % 2 forms s = z[j] + z[j + M] and d = z[j] - z[j + M] for ITS 24 values of j, keeps the one its own transform needs
// (thread 0: s, thread 1: d) and swaps the other with its neighbour (DPP) -- 24 complex numbers cross lanes, nothing goes through LDS.
template <int NX, int PAD, int I, class Act>
__device__ __forceinline__ void fused_rows_mid(const cf (&x)[NX / 2], cf (&uu)[NX / 2], bool odd, int h, int W, Act&& act) {
  constexpr int M = NX / 2, R1 = RPlan<M>::R1, R2 = RPlan<M>::R2, Q = M / 2;
  // this thread's pixels n = 2 i + h, i < M:  the inverse's output X[2 (i + PAD / 2) + h], activated; zero behind the map (the next layer's padding)
  constexpr int ma = I + PAD / 2, mb = I + Q + PAD / 2;      // sample indices of pixels 2 I + h and 2 (I + Q) + h
  cf a = cf{0.f, 0.f}, bq = cf{0.f, 0.f};
  if constexpr (ma < M) { if (2 * I + h < W) a = act(x[R2 * (ma % R1) + ma / R1]); }
  if constexpr (mb < M) { if (2 * (I + Q) + h < W) bq = act(x[R2 * (mb % R1) + mb / R1]); }
  const cf sm = a + bq, df = a - bq;                         // j = 2 I + h:  z[j] + z[j + M],  z[j] - z[j + M]
  const cf keep = cf{odd ? df.x : sm.x, odd ? df.y : sm.y}, send = cf{odd ? sm.x : df.x, odd ? sm.y : df.y};
  const cf recv = cf{__uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send.x), 0xB1, 0xF, 0xF, true)),
                     __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send.y), 0xB1, 0xF, 0xF, true))};
  // thread 0: u[2 I] = its s, u[2 I + 1] = the neighbour's s;  thread 1: u[2 I] = the neighbour's d, u[2 I + 1] = its d -- times w^(-j) for the odd outputs
  cf e = cf{odd ? recv.x : keep.x, odd ? recv.y : keep.y}, o = cf{odd ? keep.x : recv.x, odd ? keep.y : recv.y};
  if constexpr (I > 0) {
    const float wr = odd ? Tw<-2 * I, NX>::re : 1.f, wi = odd ? Tw<-2 * I, NX>::im : 0.f;
    e = cf{fmaf(-e.y, wi, e.x * wr), fmaf(e.x, wi, e.y * wr)};
  }
  {
    const float wr = odd ? Tw<-(2 * I + 1), NX>::re : 1.f, wi = odd ? Tw<-(2 * I + 1), NX>::im : 0.f;
    o = cf{fmaf(-o.y, wi, o.x * wr), fmaf(o.x, wi, o.y * wr)};
  }
  uu[2 * I] = e;
  uu[2 * I + 1] = o;
  if constexpr (I + 1 < Q) fused_rows_mid<NX, PAD, I + 1>(x, uu, odd, h, W, act);
}
template <int NX, int PAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rows_inv_fwd_reg_kernel(const float4* __restrict__ T, float4* __restrict__ Tn, const float* __restrict__ bias,
                                                                                                  const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn, int nrows,
                                                                                                  int B, int H, int W, int C, float norm0, Fp16Scale sc) {
  constexpr int NXH = NX / 2 + 1, M = NX / 2;
  const int CP = C >> 1;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int h = (int)(g & 1), lane = threadIdx.x & 63;
  const int p = (int)((g >> 1) % CP);
  const size_t by = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)((g >> 1) / CP));      // the row: the same for the 64 lanes (C % 64 == 0)
  if (by >= (size_t)nrows) return;
  const int b = (int)(by / H), y = (int)(by % H), c = 2 * p;
  const bool odd = h != 0;
  cf u[M];
  {
    const float4* src = T + (by * NXH * C) / 2 + p;
    auto load = [&](int k) __attribute__((always_inline)) { return src[(size_t)k * CP]; };      // (Ya.re, Ya.im, Yb.re, Yb.im)
    inv_rows_load2<NX, false, 0>(u, odd ? -1.f : 1.f, odd, load);
  }
  float norm = norm0;
  if (sc.tmax) {
    float tm;
    if (sc.common) {
      tm = 0.f;
      for (int i = lane; i < sc.nb; i += 64) tm = fmaxf(tm, sc.tmax[i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o));
    } else {
      tm = sc.tmax[b];
    }
    norm = norm0 * sc.winv[0] * fp16_unscale(tm, sc.hf);
  }
  const float b0v = bias[c], b1v = bias[c + 1];
  float s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
  if (relu_bn) { s0 = scale[c]; h0 = shift[c]; s1 = scale[c + 1]; h1 = shift[c + 1]; }
  step1<M, 1>(u);
  step2_inplace<M, 1, 0>(u);      // u[R2 (m % R1) + m / R1] = X[2 m + h]: the pixel 2 m + h - PAD of this thread's parity
  cf uu[M];
  auto act = [&](cf z) __attribute__((always_inline)) {
    float v0 = fmaf(z.x, norm, b0v), v1 = fmaf(z.y, norm, b1v);
    if (relu_bn) { v0 = fmaf(fmaxf(v0, 0.f), s0, h0); v1 = fmaf(fmaxf(v1, 0.f), s1, h1); }
    return cf{v0, v1};
  };
  fused_rows_mid<NX, PAD, 0>(u, uu, odd, h, W, act);
  step1<M, -1>(uu);
  step2_inplace<M, -1, 0>(uu);
  const int cblk = p >> 5, v = p & 31;
  float4* dst = Tn + ((((size_t)cblk * 4 + (v >> 3)) * B + b) * H + y) * 8 + (v & 7);      // t_fwd_index(k = 0); per kx: + (C / 16) B H 8
  const size_t kstride = (size_t)(C >> 4) * B * H * 8;
  float m = 0.f;
  fwd_rows_visit<NX, 0>(uu, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
    const int k = 2 * mi + h;
    if (k <= M) {
      dst[(size_t)k * kstride] = o;
      m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  });
  if (sc.tmax_next) {      // the next layer's per-image word (max |T|) of its spectra's scale
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(sc.tmax_next + b), __float_as_uint(m));
  }
}
// true: launched (96-point rows, pad 2 or 4, whole 64-channel blocks)
bool cfft_rows_inv_fwd_reg(int NX, const ConvArgs& a, const cf* T, cf* Tn, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  if (NX != 96 || (pad != 2 && pad != 4) || a.Cout % 64 || a.W > NX) return false;
  const int nrows = a.B * a.H;
  const size_t threads = (size_t)nrows * a.Cout;
  const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
  if (pad == 4) hipLaunchKernelGGL((rows_inv_fwd_reg_kernel<96, 4>), grid, blk, 0, st, reinterpret_cast<const float4*>(T), reinterpret_cast<float4*>(Tn), a.bias, a.scale, a.shift, a.relu_bn, nrows, a.B, a.H, a.W, a.Cout, norm, sc);
  else hipLaunchKernelGGL((rows_inv_fwd_reg_kernel<96, 2>), grid, blk, 0, st, reinterpret_cast<const float4*>(T), reinterpret_cast<float4*>(Tn), a.bias, a.scale, a.shift, a.relu_bn, nrows, a.B, a.H, a.W, a.Cout, norm, sc);
  return true;
}

// ---- rows, inverse of the full-resolution branch's conv4 + epilogue + BRANCH MERGE + rows, forward of conv5 (the contract of rows_inv_merge_fwd_kernel,
// conv_fft_rows_fused.hip, for the model's geometry: 90-column maps, x2 at half and x3 at a quarter of the width): x1 never reaches HBM.
// rows_inv_fwd_reg_kernel with one more step between the transforms: merged = ((act(x1) + up(x2)) + up(x3)) / 3 (main.py:58,67,69-70).  The coarse rows a
// fine row needs -- two source rows of x2 and of x3 -- are lerped ALONG Y FIRST by the wave that owns the row (the same for its 32 channel pairs: the row
// taps are scalars) and staged in the wave's own LDS slice as [x2: 45 columns + a copy of the last | x3: 23 columns][32 pairs]; the lerp along x then
// reads two staged neighbours per map at compile-time offsets (x2: columns i, i + 1 with weight 0 or 1/2 by the thread's parity; x3: the TF-1.x taps of
// UpTaps<90, 23>, the thread's parity picks between two literals).  TF lerps along x first: the two orders differ in the last fp32 bit of the coarse terms
// (as in rows_fwd_merge_reg_kernel), eight orders of magnitude below the 1e-4 the heat maps are held to.  No work-group barrier: a wave reads what it wrote.
__device__ __forceinline__ float bf16_rn(float v) { return static_cast<float>(static_cast<__bf16>(v)); }
// BF (bf16 handles): the three branches are bf16 tensors and so is the merged map -- act() rounds x1 to bf16, the lerps are the FMA forms of
// rows_fwd_merge_reg_kernel, the merged value is rounded to bf16 before it enters conv5's transform.
template <int NX, int PAD, int W, int W3, int IPX, bool BF, class Act>
__device__ __forceinline__ cf merged_px(cf z, bool odd, const cf* c2, const cf* c3, float t2h, Act&& act) {      // pixel n = 2 IPX + h of this thread
  using T3 = UpTaps<W, W3>;
  const cf a = act(z);
  const cf l2 = c2[IPX * 32], r2 = c2[(IPX + 1) * 32];
  constexpr int n0 = 2 * IPX, n1 = 2 * IPX + 1;
  const int lo = odd ? T3::lo(n1) * 32 : T3::lo(n0) * 32, hi = odd ? T3::hi(n1) * 32 : T3::hi(n0) * 32;
  const float t3 = odd ? T3::t(n1) : T3::t(n0);
  const cf l3 = c3[lo], r3 = c3[hi];
  if constexpr (BF) {
    const cf u2 = lerp_cf(l2, r2, t2h), u3 = lerp_cf(l3, r3, t3);
    // the third as ONE multiplication by RN(1/3), as rows_fwd_merge_reg_kernel forms it: the two kernels are bit-identical arms of a bf16 handle.  (Against the
    // correctly rounded quotient the product differs in the last fp32 bit for a third of the values, which moves the bf16 rounding of two values in a million.)
    constexpr float k3 = 0.333333343267440796f;
    return cf{bf16_rn(((a.x + u2.x) + u3.x) * k3), bf16_rn(((a.y + u2.y) + u3.y) * k3)};
  } else {
    const cf u2 = cf{l2.x + (r2.x - l2.x) * t2h, l2.y + (r2.y - l2.y) * t2h};
    const cf u3 = cf{l3.x + (r3.x - l3.x) * t3, l3.y + (r3.y - l3.y) * t3};
    return cf{div3((a.x + u2.x) + u3.x), div3((a.y + u2.y) + u3.y)};
  }
}
template <int NX, int PAD, int W, int W3, int I, bool BF, class Act>
__device__ __forceinline__ void fused_rows_mid_merge(const cf (&x)[NX / 2], cf (&uu)[NX / 2], bool odd, const cf* c2, const cf* c3, float t2h, Act&& act) {
  constexpr int M = NX / 2, R1 = RPlan<M>::R1, R2 = RPlan<M>::R2, Q = M / 2;
  static_assert(W % 2 == 0 && PAD % 2 == 0, "a thread's pixels 2 i + h are inside the map for both parities or for neither");
  constexpr int ma = I + PAD / 2, mb = I + Q + PAD / 2;      // sample indices of pixels 2 I + h and 2 (I + Q) + h
  cf a = cf{0.f, 0.f}, bq = cf{0.f, 0.f};
  if constexpr (ma < M && 2 * I < W) a = merged_px<NX, PAD, W, W3, I, BF>(x[R2 * (ma % R1) + ma / R1], odd, c2, c3, t2h, act);
  if constexpr (mb < M && 2 * (I + Q) < W) bq = merged_px<NX, PAD, W, W3, I + Q, BF>(x[R2 * (mb % R1) + mb / R1], odd, c2, c3, t2h, act);
  const cf sm = a + bq, df = a - bq;                         // j = 2 I + h:  z[j] + z[j + M],  z[j] - z[j + M]
  const cf keep = cf{odd ? df.x : sm.x, odd ? df.y : sm.y}, send = cf{odd ? sm.x : df.x, odd ? sm.y : df.y};
  const cf recv = cf{__uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send.x), 0xB1, 0xF, 0xF, true)),
                     __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send.y), 0xB1, 0xF, 0xF, true))};
  cf e = cf{odd ? recv.x : keep.x, odd ? recv.y : keep.y}, o = cf{odd ? keep.x : recv.x, odd ? keep.y : recv.y};
  if constexpr (I > 0) {
    const float wr = odd ? Tw<-2 * I, NX>::re : 1.f, wi = odd ? Tw<-2 * I, NX>::im : 0.f;
    e = cf{fmaf(-e.y, wi, e.x * wr), fmaf(e.x, wi, e.y * wr)};
  }
  {
    const float wr = odd ? Tw<-(2 * I + 1), NX>::re : 1.f, wi = odd ? Tw<-(2 * I + 1), NX>::im : 0.f;
    o = cf{fmaf(-o.y, wi, o.x * wr), fmaf(o.x, wi, o.y * wr)};
  }
  uu[2 * I] = e;
  uu[2 * I + 1] = o;
  if constexpr (I % 2 == 1) asm volatile("" ::: "memory");      // two steps' staged reads at a time (the compiler otherwise hoists all of them to the front)
  if constexpr (I + 1 < Q) fused_rows_mid_merge<NX, PAD, W, W3, I + 1, BF>(x, uu, odd, c2, c3, t2h, act);
}
// BF = false: fp32 handles (T' and T complex fp32, x2 / x3 fp32 NHWC).  BF = true: bf16 handles on the one-part route -- T' and T complex fp16 in block floating
// point (sc.t16_inv: the scale words of T'; t16n: those of the T written here, one per (image, row, 64 channels) = per wave), x2 / x3 bf16 NHWC.
template <int NX, int PAD, int W, int W2, int W3, bool BF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rows_inv_merge_fwd_reg_kernel(const void* __restrict__ T, void* __restrict__ Tn, const float* __restrict__ bias,
                                                                                                        const float* __restrict__ scale, const float* __restrict__ shift, int relu_bn,
                                                                                                        int nrows, int B, int H, int C, float norm0, Fp16Scale sc,
                                                                                                        const void* __restrict__ x2, int H2, const void* __restrict__ x3, int H3, float sy2,
                                                                                                        float sy3, float* __restrict__ t16n) {
  constexpr int NXH = NX / 2 + 1, M = NX / 2;
  constexpr int NC2 = W2 + 1, NCS = NC2 + W3 + 1;      // staged columns per wave: x2 (+ a copy of its last column), x3 (+ one the loop below writes and nobody reads)
  static_assert(W == 2 * W2 && W2 % 2 == 1 && W3 % 2 == 1, "the staging loops below walk the coarse columns in pairs");
  __shared__ cf stage[4][NCS * 32];
  const int CP = C >> 1;
  const int lane = threadIdx.x & 63, h = lane & 1;
  int p;
  size_t by;
  if (C == 512) {
    // Eight 64-channel blocks per row: work group i runs on XCD i % 8, so let it be block i % 8 of FOUR rows (one per wave) instead of four blocks of one
    // row -- an XCD then sees one channel block of every row, and the coarse rows, which 2-8 fine rows share, stay in its L2 (the mapping of
    // rows_fwd_merge_reg_kernel; counters, round 6: 3.30 GB fetched per 256 bf16 images with the linear mapping for 1.54 GB of T' and 0.44 GB of coarse maps)
    p = (int)(blockIdx.x & 7) * 32 + (lane >> 1);
    by = (size_t)(blockIdx.x >> 3) * 4 + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  } else {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    p = (int)((g >> 1) % CP);
    by = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)((g >> 1) / CP));      // the row: the same for the 64 lanes (C % 64 == 0)
  }
  if (by >= (size_t)nrows) return;
  const int b = (int)(by / H), y = (int)(by % H), c = 2 * p;
  const bool odd = h != 0;
  cf* cs = stage[threadIdx.x >> 6];
  // bf16 handles: T' (complex fp16) goes out FIRST -- 49 eight-byte loads per thread that stay in flight under the coarse rows' loads, lerps and LDS stores
  // (the kernel is bound by the latency of its dependent memory round trips at two waves per SIMD: one round trip less).  fp32 handles load T' in batches
  // behind the staging (a thread cannot hold 49 x 16 bytes next to the coarse rows).
  typedef unsigned u2t __attribute__((ext_vector_type(2)));
  u2t raw[BF ? NXH : 1];
  if constexpr (BF) {
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(static_cast<const unsigned*>(T)) + by * NXH * C, 0, NXH * C * 4, 0x00020000);
    const int vo = p * 8, ko = CP * 8;
#pragma unroll
    for (int n = 0; 2 * n <= M; ++n) {      // in the order of use: the entries are consumed in pairs (n, M - n)
      raw[n] = __builtin_amdgcn_raw_buffer_load_b64(d, vo, n * ko, 0);
      if (2 * n != M) raw[M - n] = __builtin_amdgcn_raw_buffer_load_b64(d, vo, (M - n) * ko, 0);
    }
  }
  {
    // the wave's coarse rows: lanes 0..31 take an even column of the wave's 32 channel pairs, lanes 32..63 the odd one next to it; the image is the
    // descriptor, the source row and the column pair scalar offsets
    constexpr int EB = BF ? 4 : 8;      // bytes of a channel pair
    const int bs = __builtin_amdgcn_readfirstlane(b), c0 = __builtin_amdgcn_readfirstlane(p - (lane >> 1));      // first pair of the wave
    const Tap ty2 = tf1_tap(y, H2, sy2), ty3 = tf1_tap(y, H3, sy3);
    const auto d2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(x2)) + (size_t)bs * H2 * W2 * CP * EB, 0, H2 * W2 * CP * EB, 0x00020000);
    const auto d3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(x3)) + (size_t)bs * H3 * W3 * CP * EB, 0, H3 * W3 * CP * EB, 0x00020000);
    const int vo = ((lane >> 5) * CP + c0 + (lane & 31)) * EB, vo_last = (c0 + (lane & 31)) * EB;      // (the last column pair: both halves read the last column)
    const int r2a = ty2.lo * W2 * CP * EB, r2b = ty2.hi * W2 * CP * EB, r3a = ty3.lo * W3 * CP * EB, r3b = ty3.hi * W3 * CP * EB;
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int J2 = (W2 + 1) / 2, J3 = (W3 + 1) / 2;
    auto ld = [&](const auto& d, int v, int so) __attribute__((always_inline)) {
      if constexpr (BF) return bf16pair(__builtin_amdgcn_raw_buffer_load_b32(d, v, so, 0));
      else return cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d, v, so, 0)));
    };
    auto ylerp = [&](cf a, cf bb, float t) __attribute__((always_inline)) {
      if constexpr (BF) return lerp_cf(a, bb, t);
      else return cf{a.x + (bb.x - a.x) * t, a.y + (bb.y - a.y) * t};
    };
    cf a2[J2], b2[J2], a3[J3], b3[J3];
#pragma unroll
    for (int j = 0; j < J2; ++j) {
      const int v = j == J2 - 1 ? vo_last : vo;
      a2[j] = ld(d2, v, r2a + 2 * j * CP * EB);
      b2[j] = ld(d2, v, r2b + 2 * j * CP * EB);
    }
#pragma unroll
    for (int j = 0; j < J3; ++j) {
      const int v = j == J3 - 1 ? vo_last : vo;
      a3[j] = ld(d3, v, r3a + 2 * j * CP * EB);
      b3[j] = ld(d3, v, r3b + 2 * j * CP * EB);
    }
    __builtin_amdgcn_sched_barrier(0);      // every load is out before the first lerp
#pragma unroll
    for (int j = 0; j < J2; ++j) cs[lane + 64 * j] = ylerp(a2[j], b2[j], ty2.t);
#pragma unroll
    for (int j = 0; j < J3; ++j) cs[NC2 * 32 + lane + 64 * j] = ylerp(a3[j], b3[j], ty3.t);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  cf u[M];
  if constexpr (BF) {
    // T' as complex fp16 in block floating point (requested above): the conversions of rows_inv_reg_kernel<NX, *, true>
    const int nblk = C / sc.t16_cb;
    const float* ssrc = sc.t16_inv + ((size_t)__builtin_amdgcn_readfirstlane(b) * nblk + __builtin_amdgcn_readfirstlane(c / sc.t16_cb)) * NXH;
    auto load = [&](int k) __attribute__((always_inline)) {
      const float sk = ssrc[k];
      const cf ya = unpack_h2_mix_s(raw[k][0], sk), yb = unpack_h2_mix_s(raw[k][1], sk);
      return make_float4(ya.x, ya.y, yb.x, yb.y);
    };
    inv_rows_load2<NX, true, 0>(u, odd ? -1.f : 1.f, odd, load);
  } else {
    const float4* src = static_cast<const float4*>(T) + (by * NXH * C) / 2 + p;
    auto load = [&](int k) __attribute__((always_inline)) { return src[(size_t)k * CP]; };      // (Ya.re, Ya.im, Yb.re, Yb.im)
    inv_rows_load2<NX, false, 0>(u, odd ? -1.f : 1.f, odd, load);
  }
  float norm = norm0;
  if (sc.tmax) {
    float tm;
    if (sc.common) {
      tm = 0.f;
      for (int i = lane; i < sc.nb; i += 64) tm = fmaxf(tm, sc.tmax[i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o));
    } else {
      tm = sc.tmax[b];
    }
    norm = norm0 * sc.winv[0] * fp16_unscale(tm, sc.hf);
  }
  const float b0v = bias[c], b1v = bias[c + 1];
  float s0 = 1.f, s1 = 1.f, h0 = 0.f, h1 = 0.f;
  if (relu_bn) { s0 = scale[c]; h0 = shift[c]; s1 = scale[c + 1]; h1 = shift[c + 1]; }
  step1<M, 1>(u);
  step2_inplace<M, 1, 0>(u);      // u[R2 (m % R1) + m / R1] = X[2 m + h]: the pixel 2 m + h - PAD of this thread's parity
  cf uu[M];
  auto act = [&](cf z) __attribute__((always_inline)) {
    float v0 = fmaf(z.x, norm, b0v), v1 = fmaf(z.y, norm, b1v);
    if (relu_bn) { v0 = fmaf(fmaxf(v0, 0.f), s0, h0); v1 = fmaf(fmaxf(v1, 0.f), s1, h1); }
    if constexpr (BF) { v0 = bf16_rn(v0); v1 = bf16_rn(v1); }      // x1 is a bf16 tensor
    return cf{v0, v1};
  };
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const cf* c2 = cs + (lane >> 1);
  fused_rows_mid_merge<NX, PAD, W, W3, 0, BF>(u, uu, odd, c2, c2 + NC2 * 32, odd ? 0.5f : 0.f, act);
  step1<M, -1>(uu);
  step2_inplace<M, -1, 0>(uu);
  const int cblk = p >> 5, v = p & 31;
  const size_t kstride = (size_t)(C >> 4) * B * H * 8;
  const size_t d0 = ((((size_t)cblk * 4 + (v >> 3)) * B + b) * H + y) * 8 + (v & 7);      // t_fwd_index(k = 0); per kx: + (C / 16) B H 8
  float m = 0.f;
  if constexpr (BF) {
    // the tile's largest |component| (k = 2 m + h <= NX / 2) -> its block-floating-point scale: a wave is one (image, row, 64 channels) tile
    fwd_rows_visit<NX, 0>(uu, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
      if (2 * mi + h <= M) m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    });
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
    const float sb = bfp_scale(m);
    if (lane == 0) t16n[((size_t)b * (C >> 6) + cblk) * H + y] = 1.0f / sb;
    uint2* dst = static_cast<uint2*>(Tn) + d0;
    fwd_rows_visit<NX, 0>(uu, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
      const int k = 2 * mi + h;
      if (k <= M) dst[(size_t)k * kstride] = make_uint2(pack_h2(o.x * sb, o.y * sb), pack_h2(o.z * sb, o.w * sb));
    });
  } else {
    float4* dst = static_cast<float4*>(Tn) + d0;
    fwd_rows_visit<NX, 0>(uu, odd, [&](int mi, const float4& o) __attribute__((always_inline)) {
      const int k = 2 * mi + h;
      if (k <= M) {
        dst[(size_t)k * kstride] = o;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  }
  // the next layer's per-image word (max |T|) of its spectra's scale
  if (sc.tmax_next && lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(sc.tmax_next + b), __float_as_uint(m));
}
// true: launched (the model's geometry: 96-point rows, pad 4, 90 / 45 / 23 columns, whole 64-channel blocks)
bool cfft_rows_inv_merge_fwd_reg_supported(int NX, const ConvArgs& a, const FftMerge& m, int pad) {
  if (NX != 96 || pad != 4 || a.Cout % 64 || a.W != 90 || m.W2 != 45 || m.W3 != 23 || m.H2 < 1 || m.H3 < 1) return false;
  return (size_t)m.H2 * m.W2 * a.Cout * 4 < (size_t)1 << 31;      // one buffer descriptor per coarse image
}
// t16n: null = fp32 handles (T', T complex fp32; x2, x3 fp32 NHWC); else the scale words of the 16-bit T written here (bf16 handles: T' 16-bit with sc.t16_inv, x2 / x3 bf16 NHWC)
bool cfft_rows_inv_merge_fwd_reg(int NX, const ConvArgs& a, const FftMerge& m, const cf* T, cf* Tn, int pad, float norm, const Fp16Scale& sc, hipStream_t st, float* t16n) {
  if (!cfft_rows_inv_merge_fwd_reg_supported(NX, a, m, pad) || ((t16n != nullptr) != (sc.t16_inv != nullptr))) return false;
  const int nrows = a.B * a.H;
  const size_t threads = a.Cout == 512 ? (size_t)((nrows + 3) / 4) * 8 * 256 : (size_t)nrows * a.Cout;      // (512 channels: eight work groups per four rows)
  const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
  const float sy2 = (float)m.H2 / (float)a.H, sy3 = (float)m.H3 / (float)a.H;
  if (t16n)
    hipLaunchKernelGGL((rows_inv_merge_fwd_reg_kernel<96, 4, 90, 45, 23, true>), grid, blk, 0, st, static_cast<const void*>(T), static_cast<void*>(Tn), a.bias, a.scale, a.shift, a.relu_bn, nrows, a.B,
                       a.H, a.Cout, norm, sc, m.x2, m.H2, m.x3, m.H3, sy2, sy3, t16n);
  else
    hipLaunchKernelGGL((rows_inv_merge_fwd_reg_kernel<96, 4, 90, 45, 23, false>), grid, blk, 0, st, static_cast<const void*>(T), static_cast<void*>(Tn), a.bias, a.scale, a.shift, a.relu_bn, nrows, a.B,
                       a.H, a.Cout, norm, sc, m.x2, m.H2, m.x3, m.H3, sy2, sy3, nullptr);
  return true;
}

template <int NX> static bool launch_rows_inv_reg(const ConvArgs& a, int layout, const cf* T, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  const int nrows = a.B * a.H;
  if (a.wout_TX > 0 && (layout != 0 || sc.t16_inv)) return false;      // the scatter of overlap-save windows exists for fp32 outputs
  if ((a.Cout & 1) || a.CoutP % 64) return false;      // channel pairs are stored as one word; a wave = 32 pairs of ONE row (the kernel keeps the row in scalar registers)
  const size_t threads = (size_t)nrows * a.CoutP;      // two threads per channel pair
  const dim3 grid((unsigned)((threads + 255) / 256)), blk(256);
  const bool h16 = sc.t16_inv != nullptr;
#define RR_LAUNCH(L, H16) hipLaunchKernelGGL((rows_inv_reg_kernel<NX, L, H16>), grid, blk, 0, st, T, a.out, a.bias, a.scale, a.shift, a.relu_bn, nrows, a.H, a.W, a.CoutP, a.Cout, pad, norm, sc, a.wout_H, a.wout_W, a.wout_TY, a.wout_TX)
  if (layout == 0 && !h16) RR_LAUNCH(0, false);
  else if (layout == 1 && h16 && a.Cout % 8 == 0) RR_LAUNCH(1, true);
  else if (layout == 1 && a.Cout % 8 == 0) RR_LAUNCH(1, false);
  else if (layout == 2 && h16 && a.Cout % 8 == 0) RR_LAUNCH(2, true);
  else if (layout == 2 && a.Cout % 8 == 0) RR_LAUNCH(2, false);
  else return false;
#undef RR_LAUNCH
  return true;
}
// true: launched.  false: no register kernel for this (length, layout) -- the caller takes the LDS kernel.
bool cfft_rows_inv_reg(int NX, const ConvArgs& a, int layout, const cf* T, int pad, float norm, const Fp16Scale& sc, hipStream_t st) {
  if (NX == 96) return launch_rows_inv_reg<96>(a, layout, T, pad, norm, sc, st);
  if (NX == 32 && layout == 0) return launch_rows_inv_reg<32>(a, layout, T, pad, norm, sc, st);      // the training step's overlap-save windows (fp32)
  if (NX == 50) return launch_rows_inv_reg<50>(a, layout, T, pad, norm, sc, st);      // the half- and quarter-resolution branches (36 x 50, 20 x 28 transforms)
  if (NX == 28) return launch_rows_inv_reg<28>(a, layout, T, pad, norm, sc, st);
  return false;
}

}  // namespace cfft
}  // namespace jcm
