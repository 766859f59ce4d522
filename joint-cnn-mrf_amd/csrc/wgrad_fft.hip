// Weight gradient of a stride-1 SAME convolution in the FREQUENCY domain (fp32 handles, training step; reference: the gradient of
// conv2d in main.py:133-135 that tf.gradients builds at main.py:557-560).
//
// With X = transform of the layer input and dZ = transform of the gradient of the layer output (both zero-padded to the circular size
// NY x NX of conv_fft.hip, both at the origin), the cross-correlation theorem gives, for every lag (ly, lx),
//     C[ly][lx] = sum_{b,y,x} dz[b][y][x] x[b][y - ly][x - lx] = 1/(NY NX) sum_f P[f] e^{+2 pi i (ky ly / NY + kx lx / NX)},
//     P[f][ci][co] = sum_b conj(X[f][b][ci]) dZ[f][b][co],
// alias-free for |l| <= pad because NY >= H + pad (the same argument as the forward pass), and dw[j][i] = C[pad - j][pad - i].
//
//   wgrad_spec_kernel : P for every frequency = one complex [Cin x B] x [B x Cout] product -- K is the BATCH (16 images per GPU:
//                       one k16 step), so the kernel is bound by writing P (8 bytes per (f, ci, co)).  Operands are the split spectra the
//                       forward transforms already produced for the channel GEMM (cgemm_split.hip: units of 8 CHANNELS of one image);
//                       the MFMA wants 8 IMAGES of one channel per lane, which is what gfx950's transposing LDS read delivers
//                       (ds_read_b64_tr_b16, as in wgrad_split.hip): the units are staged as they lie, [image][channel], and read transposed.
//                       Three bf16 parts per operand, six products: fp32-class, like the forward pass.
//   wgrad_taps_*      : the k x k taps of the inverse transform, separable: k column sums over ky for every kx (streams P once), then k x k
//                       row sums over kx with the Hermitian weights of the half spectrum (double), + lmbd * w.
#include "conv_fft_common.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace wf {
constexpr int TM = 128, TN = 128, NT = 256;
constexpr int PITCH = 2 * TM + 16;                 // bytes per image row of an LDS plane: 128 channels + one 16-byte pad (rows 4 apart hit different banks)
constexpr int PLANE = 16 * PITCH;                  // one (re|im, part) plane: 16 images
constexpr int lds_bytes(int np) { return 2 * 2 * np * PLANE; }      // two operands of (re|im) x np planes

struct Args {
  const uint4* xs;      // [f][mtile][Cin/16][re|im][part][k-half][MT][8]
  const uint4* zs;      // [f][mtile][Cout/16][...]
  float2* P;            // [f][Cin][Cout]
  int F, B, MTx, MTz, Cin, Cout;      // MTx / MTz: rows per M tile of the two spectra (cgemm_split_mtile of the GEMM each was laid out for)
};

__device__ __forceinline__ u32x2 tr_read(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// the 8 images (k) x 32 channels fragment of plane `pl` whose first channel is ch0: two transposing reads (images 0..3, 4..7 of the lane's
// k half); the caller waits once for all the fragments of a product
struct Frag { u32x2 lo, hi; };
__device__ __forceinline__ Frag frag_read(unsigned base, int pl, int ch0, unsigned lane_off) {
  const unsigned a = base + (unsigned)(pl * PLANE + ch0 * 2) + lane_off;
  Frag f;
  f.lo = tr_read(a);
  f.hi = tr_read(a + 4 * PITCH);
  return f;
}
template <bool HALF>
__device__ __forceinline__ f32x16 mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 frag_pack(const Frag& f) {
  const u32x4 r = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
  return __builtin_bit_cast(bf16x8, r);
}

// NP = 2, HALF = true: two FP16 parts of the scaled spectra (conv_fft's np = 4), three products; P comes out scaled by both operands' powers of two, which the taps kernel undoes.
template <int NP, bool HALF>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_spec_kernel(Args a) {
  constexpr int OPER = 2 * NP * PLANE, SEGS = 8 * 2 * NP * 2;      // one operand in LDS; (chunk, re|im, part, k-half) segments of 16 images x 16 bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (a.Cout + TN - 1) / TN, tiles_m = (a.Cin + TM - 1) / TM;
  const int tile = blockIdx.x % (tiles_m * tiles_n), f = blockIdx.x / (tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int KCi = a.Cin / 16, KCo = a.Cout / 16;
  const int rows_valid = (a.B + 3) / 4 * 4;      // the forward column pass writes whole groups of 4 or 8 images (zeros behind the last one): rows it surely wrote
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // transposing read: in each 16-lane group g, lane t supplies the 8-byte piece (image t/4, channels 4(t%4)..+3) of a 4 x 16 block and
  // receives channel t of it; group g -> channel block g&1, k half g>>1 (the MFMA operand lane l holds row/column l&31, k half l>>5)
  const int g = lane >> 4, t = lane & 15;
  const unsigned lane_off = (unsigned)((8 * (g >> 1) + (t >> 2)) * PITCH + (16 * (g & 1) + 4 * (t & 3)) * 2);

  f32x16 pr[2][2], pi[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { pr[i][j][e] = 0.f; pi[i][j][e] = 0.f; }

  // One k16 step = 16 images of both operands.  The operands of step r0 + 16 are requested (into registers) before the products of step r0 run: with
  // overlap-save windows (jcm_train.hip) K is 12 steps, not 1, and a synchronous stage per step left the kernel waiting on the L2 (1.7 ms for conv5).
  constexpr int NLD = 2 * SEGS * 16 / NT;      // 16-byte loads per thread and step
  static_assert(2 * SEGS * 16 % NT == 0, "whole loads per thread");
  uint4 pre[NLD];
  auto fetch = [&](int r0) __attribute__((always_inline)) {
    int tt = tid;      // opaque per call: the segment decode below is recomputed (a few integer instructions) instead of being hoisted out of the k loop and spilled
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tt + q * NT;
      const int op = idx / (SEGS * 16), r = idx - op * SEGS * 16;
      const int seg = r >> 4, j = r & 15;
      const int kg = seg & 1, cp = (seg >> 1) % (2 * NP), kcl = seg / (4 * NP);      // cp = re|im * NP + part
      const int KC = op ? KCo : KCi, kc = (op ? tn : tm) * 8 + kcl;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (kc < KC && r0 + j < rows_valid) {
        const uint4* src = op ? a.zs : a.xs;
        const int MT = op ? a.MTz : a.MTx, mtiles = (a.B + MT - 1) / MT, mt = r0 / MT, rl = r0 - mt * MT;
        v = src[((((size_t)f * mtiles + mt) * KC + kc) * (4 * NP) + cp * 2 + kg) * MT + rl + j];
      }
      pre[q] = v;
    }
  };
  fetch(0);
  for (int r0 = 0; r0 < rows_valid; r0 += 16) {      // (16 divides both tile heights: a step never straddles two M tiles)
    __syncthreads();      // the previous step's fragments have been read
    // stage both operands: (8 chunks x re|im x part x k-half) segments of 16 images x 16 bytes, as they lie in HBM
    int ts = tid;
    asm volatile("" : "+v"(ts));
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = ts + q * NT;
      const int op = idx / (SEGS * 16), r = idx - op * SEGS * 16;
      const int seg = r >> 4, j = r & 15;
      const int kg = seg & 1, cp = (seg >> 1) % (2 * NP), kcl = seg / (4 * NP);
      *reinterpret_cast<uint4*>(smem + op * OPER + (cp * 16 + j) * PITCH + (kcl * 16 + kg * 8) * 2) = pre[q];
    }
    if (r0 + 16 < rows_valid) fetch(r0 + 16);
    __syncthreads();
    // products (x part, z part) with px + pz <= NP - 1, small terms first
    constexpr int NPROD = NP == 3 ? 6 : 3;
#pragma unroll
    for (int s = 0; s < NPROD; ++s) {
      const int px = NP == 3 ? (s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 2 : s == 3 ? 1 : 0) : (s == 0 ? 0 : s == 1 ? 1 : 0);
      const int pz = NP == 3 ? (s == 0 ? 2 : s == 1 ? 1 : s == 2 ? 0 : s == 3 ? 0 : s == 4 ? 1 : 0) : (s == 0 ? 1 : 0);
      Frag fxr[2], fxi[2], fzr[2], fzi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fxr[i] = frag_read(lds0, 0 * NP + px, (wm * 2 + i) * 32, lane_off);
        fxi[i] = frag_read(lds0, 1 * NP + px, (wm * 2 + i) * 32, lane_off);
        fzr[i] = frag_read(lds0 + OPER, 0 * NP + pz, (wn * 2 + i) * 32, lane_off);
        fzi[i] = frag_read(lds0 + OPER, 1 * NP + pz, (wn * 2 + i) * 32, lane_off);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 xr[2], xi[2], nxi[2], zr[2], zi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xr[i] = frag_pack(fxr[i]); xi[i] = frag_pack(fxi[i]); zr[i] = frag_pack(fzr[i]); zi[i] = frag_pack(fzi[i]);
        const u32x4 n = __builtin_bit_cast(u32x4, xi[i]) ^ u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        nxi[i] = __builtin_bit_cast(bf16x8, n);
      }
      // P = conj(X)^T dZ:  Pr += Xr Zr + Xi Zi,  Pi += Xr Zi - Xi Zr
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          pr[i][j] = mma<HALF>(xr[i], zr[j], pr[i][j]);
          pi[i][j] = mma<HALF>(xr[i], zi[j], pi[i][j]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          pr[i][j] = mma<HALF>(xi[i], zi[j], pr[i][j]);
          pi[i][j] = mma<HALF>(nxi[i], zr[j], pi[i][j]);
        }
    }
  }
  // ---- P[f][ci][co]: accumulator register e is row (ci) (e&3) + 8 (e>>2) + 4 (lane>>5), column (co) lane&31
  const int h = lane >> 5, l31 = lane & 31;
  float2* pf = a.P + (size_t)f * a.Cin * a.Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = tm * TM + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (ci < a.Cin) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int co = tn * TN + (wn * 2 + j) * 32 + l31;
          if (co < a.Cout) {      // streaming store: P (F Cin Cout complex numbers, 6.6 GB for conv5) is read once, by the taps kernel, gigabytes later
            typedef float f2n __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store(f2n{pr[i][j][e], pi[i][j][e]}, reinterpret_cast<f2n*>(pf + (size_t)ci * a.Cout + co));
          }
        }
      }
    }
}

// ---- taps: dw[j][i][ci][co] = 1/(NY NX) sum_{ky, kx <= NX/2} wgt(kx) Re(P[ky][kx] e^{+2 pi i (ky (pad - j) / NY + kx (pad - i) / NX)}) + lmbd w
// Separable, in two launches that both fill the chip (one thread per (ci, co) alone would be Cin Cout / 64 waves):
//   cols : thread (pair, kx): R[kx][j][pair] = sum_ky P[kx][ky][pair] e^{+2 pi i ky (pad - j) / NY}     -- streams P once, coalesced over pairs
//   rows : thread (pair, j):  dw[j][i][pair] = sum_kx Re(R[kx][j][pair] wgt(kx) e^{+2 pi i kx (pad - i) / NX}) / (NY NX) + lmbd w (double sums)
template <int KS>
__global__ __launch_bounds__(256) void wgrad_taps_cols_kernel(const float2* __restrict__ P, float2* __restrict__ R, size_t n, int NY) {
  constexpr int PAD = (KS - 1) / 2;
  __shared__ float2 twy[192 * KS];       // e^{+2 pi i ky l / NY}, l = PAD - j
  for (int i = threadIdx.x; i < NY * KS; i += 256) {
    const int ky = i / KS, l = PAD - i % KS;
    double sn, cs;
    sincospi(2.0 * (double)(((ky * l) % NY + NY) % NY) / (double)NY, &sn, &cs);
    twy[i] = float2{(float)cs, (float)sn};
  }
  __syncthreads();
  // a thread owns TWO consecutive (ci, co) pairs: 16-byte loads (n = Cin * ldz is even), a wave instruction reads 1 KB contiguous
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= n) return;
  const int kx = blockIdx.y;
  float2 acc[KS][2];
#pragma unroll
  for (int j = 0; j < KS; ++j) acc[j][0] = acc[j][1] = float2{0.f, 0.f};
  const float2* col = P + ((size_t)kx * NY) * n + e;
  constexpr int U = 8;                   // loads in flight per thread (the tail is guarded)
  for (int ky0 = 0; ky0 < NY; ky0 += U) {
    f32x4 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ky0 + u < NY) p[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(col + (size_t)(ky0 + u) * n));
      else p[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ky = ky0 + u < NY ? ky0 + u : 0;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const float2 t = twy[ky * KS + j];
        acc[j][0].x += p[u].x * t.x - p[u].y * t.y;
        acc[j][0].y += p[u].x * t.y + p[u].y * t.x;
        acc[j][1].x += p[u].z * t.x - p[u].w * t.y;
        acc[j][1].y += p[u].z * t.y + p[u].w * t.x;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KS; ++j)
    *reinterpret_cast<f32x4*>(R + ((size_t)kx * KS + j) * n + e) = f32x4{acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y};
}

template <int KS>
__global__ __launch_bounds__(256) void wgrad_taps_rows_kernel(const float2* __restrict__ R, const float* __restrict__ w, float* __restrict__ dw, size_t n, int NY, int NX,
                                                              float lmbd, int ldp, int Cout, const float* __restrict__ tmax_x, const float* __restrict__ tmax_z, float hf, int nb) {
  constexpr int PAD = (KS - 1) / 2;
  __shared__ double2 twx[97 * KS];        // wgt(kx) e^{+2 pi i kx l / NX} / (NY NX), l = PAD - i
  const int NXH = NX / 2 + 1;
  for (int i = threadIdx.x; i < NXH * KS; i += 256) {
    const int kx = i / KS, l = PAD - i % KS;
    double sn, cs;
    sincospi(2.0 * (double)(((kx * l) % NX + NX) % NX) / (double)NX, &sn, &cs);
    double wgt = ((kx == 0 || 2 * kx == NX) ? 1.0 : 2.0) / ((double)NY * (double)NX);
    if (tmax_x) wgt *= (double)cfft::fp16_unscale(cfft::tmax_of(tmax_x, 0, nb, 1), hf) * (double)cfft::fp16_unscale(cfft::tmax_of(tmax_z, 0, nb, 1), hf);      // the scaled fp16 operands (np = 4): powers of two
    twx[i] = double2{cs * wgt, sn * wgt};
  }
  __syncthreads();
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const size_t ci = e / (size_t)ldp;
  const int co = (int)(e - ci * ldp);      // P has ldp >= Cout columns (dZ's channel stride); the filter has Cout
  if (co >= Cout) return;
  const size_t nw = n / (size_t)ldp * Cout, ew = ci * Cout + co;
  const int j = blockIdx.y;
  double acc[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) acc[i] = 0.0;
#pragma unroll 4
  for (int kx = 0; kx < NXH; ++kx) {
    const float2 r = R[((size_t)kx * KS + j) * n + e];
    const double rr = r.x, ri = r.y;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const double2 t = twx[kx * KS + i];
      acc[i] += rr * t.x - ri * t.y;
    }
  }
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const size_t o = ((size_t)(j * KS + i)) * nw + ew;
    dw[o] = (float)acc[i] + lmbd * w[o];
  }
}

}  // namespace wf

static size_t p_bytes(int NY, int NX, int Cin, int Cout) { return ((size_t)NY * (NX / 2 + 1) * Cin * Cout * sizeof(float2) + 255) & ~size_t(255); }
// P[f][ci][co] + R[kx][j][ci][co] (j < 9); Cout = dZ's channel stride
size_t wgrad_fft_scratch_bytes(int NY, int NX, int Cin, int Cout) { return p_bytes(NY, NX, Cin, Cout) + (size_t)(NX / 2 + 1) * 9 * Cin * Cout * sizeof(float2); }

hipError_t wgrad_fft(const void* xs, const void* zs, void* scratch, const float* w, float lmbd, float* dw, int ks, int NY, int NX, int B, int MTx, int MTz, int Cin,
                     int ldz, int Cout, hipStream_t st, int np, const float* tmax_x, const float* tmax_z, int H) {
  if ((ks != 9 && ks != 5) || Cin % 16 || ldz % 16 || Cout > ldz || Cout < 1 || NY > 192 || NX > 192 || B < 1 || np != 4) return hipErrorInvalidValue;
  if ((!tmax_x || !tmax_z || H < 1)) return hipErrorInvalidValue;
  const int NXH = NX / 2 + 1, F = NY * NXH;
  if (MTx % 16 || MTz % 16) return hipErrorInvalidValue;
  wf::Args a{static_cast<const uint4*>(xs), static_cast<const uint4*>(zs), static_cast<float2*>(scratch), F, B, MTx, MTz, Cin, ldz};
  float2* R = reinterpret_cast<float2*>(static_cast<char*>(scratch) + p_bytes(NY, NX, Cin, ldz));
  static LdsAttr attr4;
  const int tiles = ((Cin + wf::TM - 1) / wf::TM) * ((ldz + wf::TN - 1) / wf::TN);
  if (hipError_t e = attr4.ensure(reinterpret_cast<const void*>(wf::wgrad_spec_kernel<2, true>), wf::lds_bytes(2)); e != hipSuccess) return e;
  hipLaunchKernelGGL((wf::wgrad_spec_kernel<2, true>), dim3((unsigned)(F * tiles)), dim3(wf::NT), wf::lds_bytes(2), st, a);
  const size_t n = (size_t)Cin * ldz;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (ks == 9) {
    hipLaunchKernelGGL(wf::wgrad_taps_cols_kernel<9>, dim3((blocks + 1) / 2, NXH), dim3(256), 0, st, a.P, R, n, NY);
    hipLaunchKernelGGL(wf::wgrad_taps_rows_kernel<9>, dim3(blocks, 9), dim3(256), 0, st, R, w, dw, n, NY, NX, lmbd, ldz, Cout, tmax_x, tmax_z, (float)H, B);
  } else {
    hipLaunchKernelGGL(wf::wgrad_taps_cols_kernel<5>, dim3((blocks + 1) / 2, NXH), dim3(256), 0, st, a.P, R, n, NY);
    hipLaunchKernelGGL(wf::wgrad_taps_rows_kernel<5>, dim3(blocks, 5), dim3(256), 0, st, R, w, dw, n, NY, NX, lmbd, ldz, Cout, tmax_x, tmax_z, (float)H, B);
  }
  return hipGetLastError();
}

}  // namespace jcm
