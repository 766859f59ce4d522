// Weight gradients of the SAME convolutions (training step).
//
//   dW[ky][kx][ci][co] = sum_{b,y,x} X[b, y+ky-PAD, x+kx-PAD, ci] * dZ[b, y, x, co]
//
// GEMM view: M = Cin, N = Cout, K = B*H*W pixels, one GEMM per tap.  A workgroup owns one tap
// ROW (ky fixed, all KS values of kx) of a 64(ci) x 64(co) tile and walks K in strips of one
// image row x 30 pixels: the strip's dZ [30 px][64 co] and the matching input-row segment
// [30+KS-1 px][64 ci] are staged in LDS once, each wave lifts its 32 channels of both into
// registers (KS+43 values), and the KS taps then run 15 v_mfma_f32_32x32x2_f32 each straight
// from registers -- the kx shift is a register index.  K is split over `splits` workgroups that
// write partial tiles; wgrad_reduce sums them in a fixed order (deterministic) and adds the
// weight-decay term lmbd*W (main.py:540).
#include "kernels.h"

namespace jcm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_PW = 30;        // pixels per strip: 15 MFMA k-steps; divides the 90- and 180-wide maps of the model exactly
constexpr int WG_T = 64;         // channel tile (both ci and co)

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
                                                        int B, int H, int W, int Cin, int Cout, int ldz, int n_ci, int n_co, int splits) {
  constexpr int PAD = (KS - 1) / 2;
  constexpr int XW = WG_PW + KS - 1;           // input pixels per strip
  __shared__ float Xs[2][XW][WG_T];
  __shared__ float Zs[2][WG_PW][WG_T];

  int bid = blockIdx.x;
  const int ky = bid % KS; bid /= KS;
  const int cit = bid % n_ci; bid /= n_ci;
  const int cot = bid % n_co; bid /= n_co;
  const int split = bid;
  const int ci0 = cit * WG_T, co0 = cot * WG_T;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wo = wid & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const int nseg = (W + WG_PW - 1) / WG_PW;
  // output rows whose input row y + ky - PAD lies inside the image (the others contribute zeros: skipped)
  const int ylo = ky < PAD ? PAD - ky : 0;
  const int nvalid = H - (ky < PAD ? PAD - ky : ky - PAD);
  const long rows = nvalid > 0 ? (long)B * nvalid : 0;
  const long r0 = rows * split / splits, r1 = rows * (split + 1) / splits;
  const long nstrip = (r1 - r0) * nseg;

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // global -> register staging of one strip (zero outside the image / beyond the channel counts)
  constexpr int XF4 = XW * WG_T / 4, ZF4 = WG_PW * WG_T / 4;
  constexpr int XR = (XF4 + 255) / 256, ZR = (ZF4 + 255) / 256;
  f32x4 xr[XR], zr[ZR];
  auto gload = [&](long s) {
    const long row = r0 + s / nseg;
    const int seg = (int)(s % nseg);
    const int b = (int)(row / nvalid), y = ylo + (int)(row % nvalid);
    const int yi = y + ky - PAD;
    const int px0 = seg * WG_PW;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int idx = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < XF4) {
        const int c4 = idx % (WG_T / 4), p = idx / (WG_T / 4);
        const int xi = px0 + p - PAD, ci = ci0 + c4 * 4;
        if ((unsigned)xi < (unsigned)W && ci < Cin)
          v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + yi) * W + xi) * Cin + ci);
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const int idx = tid + i * 256;
      const int c4 = idx % (WG_T / 4), p = idx / (WG_T / 4);
      const int xo = px0 + p, co = co0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < ZF4 && xo < W && co < ldz) v = *reinterpret_cast<const f32x4*>(dz + (((size_t)b * H + y) * W + xo) * ldz + co);
      zr[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int idx = tid + i * 256;
      if (idx < XF4) *reinterpret_cast<f32x4*>(&Xs[buf][idx / (WG_T / 4)][(idx % (WG_T / 4)) * 4]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const int idx = tid + i * 256;
      if (idx < ZF4) *reinterpret_cast<f32x4*>(&Zs[buf][idx / (WG_T / 4)][(idx % (WG_T / 4)) * 4]) = zr[i];
    }
  };

  if (nstrip > 0) gload(0);
  int buf = 0;
  for (long s = 0; s < nstrip; ++s) {
    lstore(buf);
    __syncthreads();                       // strip s visible.  One barrier per strip is enough with two buffers: a wave reaches
                                           // this barrier only after its reads of strip s-1, and buffer `buf` is rewritten at s+2
    if (s + 1 < nstrip) gload(s + 1);      // in flight behind this strip's MFMAs
    // lift this wave's operands into registers: lane (l31, h) needs input pixels h + m, m = 0 .. 30+KS-1 (even steps use m = 2kk+kx)
    constexpr int NA = WG_PW - 2 + KS;     // avv[2kk + kx] = input pixel 2kk + kx + h of the strip, 2kk + kx = 0 .. 30+KS-1
    float avv[NA];
    float bv[WG_PW / 2];
#pragma unroll
    for (int m = 0; m < NA; ++m) avv[m] = Xs[buf][m + h][wi * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < WG_PW / 2; ++kk) bv[kk] = Zs[buf][2 * kk + h][wo * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < WG_PW / 2; ++kk)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[2 * kk + kx], bv[kk], acc[kx], 0, 0, 0);
    buf ^= 1;
  }

  // ---- partial tile store: D col = lane&31 -> co, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> ci
  float* out = partial + (size_t)split * KS * KS * Cin * Cout;
  const int co = co0 + wo * 32 + l31;
  if (co < Cout) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = ci0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (ci < Cin) out[(((size_t)ky * KS + kx) * Cin + ci) * Cout + co] = acc[kx][i];
      }
    }
  }
}

int wgrad_splits(int ks, int Cin, int Cout, int B, int H) {
  const int n_ci = (Cin + WG_T - 1) / WG_T, n_co = (Cout + WG_T - 1) / WG_T;
  const int base = ks * n_ci * n_co;
  int s = (2048 + base - 1) / base;            // ~8 workgroups per CU
  const long rows = (long)B * H;
  if (s > rows) s = (int)rows;
  if (s > 256) s = 256;
  return s < 1 ? 1 : s;
}

// x [B,H,W,Cin] (Cin % 4 == 0), dz [B,H,W,ldz] (ldz % 4 == 0, first Cout channels used), partial [splits][KS][KS][Cin][Cout]
hipError_t wgrad_f32(const float* x, const float* dz, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                     hipStream_t st) {
  if (Cin % 4 || ldz % 4) return hipErrorInvalidValue;
  const int n_ci = (Cin + WG_T - 1) / WG_T, n_co = (Cout + WG_T - 1) / WG_T;
  const int blocks = ks * n_ci * n_co * splits;
  if (ks == 9)
    hipLaunchKernelGGL(wgrad_kernel<9>, dim3(blocks), dim3(256), 0, st, x, dz, partial, B, H, W, Cin, Cout, ldz, n_ci, n_co, splits);
  else if (ks == 5)
    hipLaunchKernelGGL(wgrad_kernel<5>, dim3(blocks), dim3(256), 0, st, x, dz, partial, B, H, W, Cin, Cout, ldz, n_ci, n_co, splits);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// dw[i] = sum_s partial[s][i] + lmbd * w[i]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, size_t n, const float* __restrict__ w, float lmbd,
                                    float* __restrict__ dw, const float* __restrict__ out_scale) {
  const float os = out_scale ? *out_scale : 1.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0;                           // partial tiles cancel heavily (BN backward makes sum(dz) ~ 0)
    for (int k = 0; k < splits; ++k) s += (double)partial[(size_t)k * n + i];
    dw[i] = (float)s * os + lmbd * w[i];
  }
}
hipError_t wgrad_reduce(const float* partial, int splits, size_t n, const float* w, float lmbd, float* dw, hipStream_t st, const float* out_scale) {
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, partial, splits, n, w, lmbd, dw, out_scale);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv1: 5x5 stride-2 SAME (pad 1 before, 2 after), Cin = 3, on the sub-sampled image x[:, ::sub, ::sub].
// dW[(tap, ci)][co] is a [75 x Cout <= 64] matrix, K = B*Ho*Wo output pixels.  Round 5: on v_mfma_f32_32x32x2_f32 (the exact fp32 chain; rounds 1-4
// ran it on the vector ALU with one LDS read per multiply-add: 1.1 ms per step + 0.76 ms for the sum of its 1024 partial tiles).  A work group
// walks strips of 64 output pixels of one output row: the 5 input rows the strip touches ([5][131][3]) and dz [64][64] go to LDS; wave w takes
// the strip's pixels 16 w .. 16 w + 15 as 8 k-steps of the 96 x 64 tile (3 x 2 fragments: A lane = (row = (tap, ci), k = pixel), gathered from
// the patch by a per-lane offset; B lane = (k = pixel, column = co)).  At the end the four waves' tiles are summed through LDS in wave order
// and the work group writes ONE partial tile; wgrad_reduce_wide adds the work groups' tiles in a fixed order (double), + lmbd * w.
// ------------------------------------------------------------------------------------------------
constexpr int W1_PX = 64;          // output pixels per LDS strip
constexpr int W1_XW = 2 * W1_PX + 3;
__device__ __forceinline__ float4 w1_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 w1_ld4(const __bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <class TZ>
__global__ __launch_bounds__(256) void wgrad_conv1_kernel(const float* __restrict__ x, const TZ* __restrict__ dz,
                                                          float* __restrict__ partial, int B, int H0, int W0, int sub, int Ho, int Wo,
                                                          int Cout, int nblk) {
  constexpr int XN = 5 * W1_XW * 3, ZN = W1_PX * 64, RN = 96 * 64;
  constexpr int XNP = (XN + 3) / 4 * 4;      // dz behind the patch, 16-byte aligned
  __shared__ __attribute__((aligned(16))) float buf[(XNP + ZN > RN ? XNP + ZN : RN)];
  float* Xs = buf;             // [5][131][3]
  float* Zs = buf + XNP;       // [64][64]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this lane's three A rows: (tap, ci) index rix = 32 mb + l31 -> offset of Xs[ky][kx][ch] (pixel 0), or invalid (rows 75..95 multiply zeros)
  int aoff[3];
  bool aval[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) {
    const int rix = mb * 32 + l31;
    aval[mb] = rix < 75;
    const int tap = aval[mb] ? rix / 3 : 0, ch = aval[mb] ? rix - tap * 3 : 0;
    const int ky = tap / 5, kx = tap - ky * 5;
    aoff[mb] = (ky * W1_XW + kx) * 3 + ch;
  }
  f32x16 acc[3][2];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  const int Hs = H0 / sub, Ws = W0 / sub;           // sub-sampled image size
  const int nseg = (Wo + W1_PX - 1) / W1_PX;
  const long nstrip = (long)B * Ho * nseg;
  for (long s = blockIdx.x; s < nstrip; s += nblk) {
    const int seg = (int)(s % nseg);
    const long row = s / nseg;
    const int b = (int)(row / Ho), oy = (int)(row % Ho);
    const int ox0 = seg * W1_PX;
    // every load of the strip goes out before the first LDS store (a load -> store loop would wait for each load in turn: that, not the MFMAs,
    // was 90 % of this kernel's time in its first version)
    constexpr int KX = (XN + 255) / 256, KZ = ZN / 4 / 256;
    float xv[KX];
    float4 zv[KZ];
#pragma unroll
    for (int i = 0; i < KX; ++i) {
      const int idx = tid + i * 256;
      const int ch = idx % 3, r = idx / 3;
      const int px = r % W1_XW, ry = r / W1_XW;
      const int iy = 2 * oy + ry - 1, ix = 2 * ox0 + px - 1;
      xv[i] = 0.f;
      if (idx < XN && (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws) xv[i] = x[(((size_t)b * H0 + (size_t)iy * sub) * W0 + (size_t)ix * sub) * 3 + ch];
    }
#pragma unroll
    for (int i = 0; i < KZ; ++i) {
      const int idx = tid + i * 256, c4 = (idx & 15) * 4, p = idx >> 4;
      const int ox = ox0 + p;
      zv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ox < Wo && c4 < Cout) zv[i] = w1_ld4(dz + (((size_t)b * Ho + oy) * Wo + ox) * Cout + c4);      // Cout % 4 == 0 (launcher)
    }
    __syncthreads();      // every wave is done with the previous strip's LDS image
#pragma unroll
    for (int i = 0; i < KX; ++i)
      if (tid + i * 256 < XN) Xs[tid + i * 256] = xv[i];
#pragma unroll
    for (int i = 0; i < KZ; ++i) *reinterpret_cast<float4*>(Zs + 4 * (tid + i * 256)) = zv[i];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int p = wid * 16 + ks * 2 + h;           // this lane's k = one output pixel of the strip
      float a[3], bz[2];
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) { const float v = Xs[aoff[mb] + 6 * p]; a[mb] = aval[mb] ? v : 0.f; }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) bz[nb] = Zs[p * 64 + nb * 32 + l31];
#pragma unroll
      for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], bz[nb], acc[mb][nb], 0, 0, 0);
    }
  }
  // the four waves' tiles, summed in wave order (deterministic); accumulator i of a fragment is row (i & 3) + 8 (i >> 2) + 4 h, column l31
  float* Red = buf;            // [96][64]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wid == w) {
#pragma unroll
      for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, cidx = nb * 32 + l31;
            Red[r * 64 + cidx] = (w == 0 ? 0.f : Red[r * 64 + cidx]) + acc[mb][nb][i];
          }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 75 * 64; idx += 256) {
    const int co = idx & 63, rix = idx >> 6;
    if (co < Cout) partial[((size_t)blockIdx.x * 75 + rix) * Cout + co] = Red[idx];
  }
}
int wgrad_conv1_blocks(void) { return 1024; }      // four work groups per CU: the strips' loads of one overlap the MFMAs of the others
// partial: [wgrad_conv1_blocks()][5][5][3][Cout]
hipError_t wgrad_conv1(const float* x, const void* dz, bool dz_bf16, float* partial, int B, int H0, int W0, int sub, int Cout, hipStream_t st) {
  if (Cout > 64 || Cout % 4) return hipErrorInvalidValue;
  const int Hs = H0 / sub, Ws = W0 / sub;
  const int Ho = (Hs + 1) / 2, Wo = (Ws + 1) / 2;
  const int nblk = wgrad_conv1_blocks();
  if (dz_bf16)
    hipLaunchKernelGGL(wgrad_conv1_kernel<__bf16>, dim3(nblk), dim3(256), 0, st, x, static_cast<const __bf16*>(dz), partial, B, H0, W0, sub, Ho, Wo, Cout, nblk);
  else
    hipLaunchKernelGGL(wgrad_conv1_kernel<float>, dim3(nblk), dim3(256), 0, st, x, static_cast<const float*>(dz), partial, B, H0, W0, sub, Ho, Wo, Cout, nblk);
  return hipGetLastError();
}

// dw[i] = sum_s partial[s][i] + lmbd * w[i] for MANY partial tiles of a SMALL tensor (conv1: 512 x 4800): a work group owns 64 outputs, its four
// waves take the tiles s = q, q + 4, ... (double sums), and the four sums are added in wave order -- a fixed association, hence deterministic
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ partial, int splits, int n, const float* __restrict__ w, float lmbd,
                                                                float* __restrict__ dw) {
  __shared__ double red[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double s = 0.0;
  if (i < n) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;      // four loads in flight (a fixed association: deterministic)
    int k = q;
    for (; k + 12 < splits; k += 16) {
      s += (double)partial[(size_t)k * n + i];
      s1 += (double)partial[(size_t)(k + 4) * n + i];
      s2 += (double)partial[(size_t)(k + 8) * n + i];
      s3 += (double)partial[(size_t)(k + 12) * n + i];
    }
    for (; k < splits; k += 4) s += (double)partial[(size_t)k * n + i];
    s = (s + s1) + (s2 + s3);
  }
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && i < n) dw[i] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]) + lmbd * w[i];
}
hipError_t wgrad_reduce_wide(const float* partial, int splits, size_t n, const float* w, float lmbd, float* dw, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, partial, splits, (int)n, w, lmbd, dw);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Data-gradient weights: dX = conv_SAME(dZ, Wd) with Wd[ky][kx][co][ci] = W[KS-1-ky][KS-1-kx][ci][co]
// (stride 1, odd KS).  `CoP` pads the (new) input-channel axis with zeros (dZ channel stride).
// ------------------------------------------------------------------------------------------------
// One work group per (tap, 32 x 32 block of (ci, co)): read coalesced over co, transposed through LDS, written coalesced over ci.  (Rounds 3-5 read w with a
// stride of Cout floats per lane: 121 us for conv5's 21 M weights, 0.7 TB/s; the step runs this for eleven layers after every update.)
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wd, int ks, int Cin, int Cout, int CoP) {
  __shared__ float tile[32][33];
  const int nci = (Cin + 31) / 32;
  const int tci = (int)(blockIdx.x % (unsigned)nci), tco = (int)(blockIdx.x / (unsigned)nci), tap = (int)blockIdx.y;
  const int ky = tap / ks, kx = tap - ky * ks;
  const size_t src = ((size_t)(ks - 1 - ky) * ks + (ks - 1 - kx)) * Cin, dst = (size_t)tap * CoP;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ci = tci * 32 + ty + 8 * k, co = tco * 32 + tx;
    tile[ty + 8 * k][tx] = (ci < Cin && co < Cout) ? w[(src + ci) * Cout + co] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = tco * 32 + ty + 8 * k, ci = tci * 32 + tx;
    if (co < CoP && ci < Cin) wd[(dst + co) * Cin + ci] = tile[tx][ty + 8 * k];
  }
}
hipError_t flip_transpose_weights(const float* w_hwio, float* wd, int ks, int Cin, int Cout, int CoP, hipStream_t st) {
  const dim3 grid((unsigned)(((Cin + 31) / 32) * ((CoP + 31) / 32)), (unsigned)(ks * ks));
  hipLaunchKernelGGL(flip_transpose_kernel, grid, dim3(256), 0, st, w_hwio, wd, ks, Cin, Cout, CoP);
  return hipGetLastError();
}

}  // namespace jcm
