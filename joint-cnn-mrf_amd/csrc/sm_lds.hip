// Whole-frame 120 x 180 transforms of the spatial model in LDS (round 5: they replace the rocFFT / hipFFT calls this library used to make for
// everything but the fused forward of sm_fused.hip).  Same building blocks as that file (sm_lds_fft.h): a work group owns one frame, its
// 120 x 91 half spectrum sits in LDS as the column buffer [91][121], two real rows travel as one complex 180-point transform through the row
// buffer [31][181], and spectra live in HBM TRANSPOSED, [frame][91][120] -- the layout of sm_fused.hip's prior and likelihood spectra.
//
//   sm_lds_fwd_frames   real [n][120][180]                      -> spectra_t [n][91][120]      (prior spectra at jcm_finalize, conv_mrf's A)
//   sm_lds_fwd_dframes  D[b][p] = R^T (G_j / T_p) on the window -> spectra_t [nb * P][91][120] (training step, main.py:94-125 backward: the
//                       adjoint of the 61x91 -> 60x90 resize is evaluated while the rows are loaded; the zero frame around the window
//                       [59..119] x [89..179] is never materialised)
//   sm_lds_inv_frames   spectra_t [n][91][120] -> rows [r0, r0 + nrows) of real frames [n][120][180], unnormalised like a C2R transform
//                       (conv_mrf: rows 59..119; the training step's dL frames: rows 0..59; its 81 dA frames: all 120 rows in two sweeps)
//
// Unnormalised transforms with the sign conventions of sm_fused.hip (forward e^{-i...}, inverse e^{+i...}); the imaginary parts of the DC /
// Nyquist columns are dropped on the way back as a C2R transform drops them.
#include "sm_lds_fft.h"

namespace jcm {

using namespace smf;

namespace {

constexpr float kSy = 61.0f / 60.0f, kSx = 91.0f / 90.0f;

// source taps of output index o of the 61 -> 60 (or 91 -> 90) TF-1.x resize (the forward's fp32 arithmetic)
__device__ __forceinline__ void tap61(int o, float s, int n_in, int* lo, int* hi, float* t) {
  const float f = __fmul_rn((float)o, s);
  const int l = (int)floorf(f);
  *lo = l;
  *hi = min(l + 1, n_in - 1);
  *t = f - (float)l;
}

struct FullSrc {      // a real frame as it lies in memory
  const float* p;
  static constexpr int R0 = 0, NR = FH, C0 = 0, NC = FW;
  __device__ __forceinline__ float at(int n, int y, int x) const { return p[((size_t)n * FH + y) * FW + x]; }
};

struct DSrc {         // (R^T q_p)[y][x], q = G_j / T_p, as frame entry (59 + y, 89 + x); n = b * P + p
  const float* G;     // [nb][5400][K]
  const float* T;     // [nb][P][5400]
  int K, P;
  static constexpr int R0 = 59, NR = 61, C0 = 89, NC = 91;
  __device__ __forceinline__ float at(int n, int y, int x) const {
    const int p = n % P, b = n / P, j = p / (P / K);
    const float* Gb = G + (size_t)b * MHW * K + j;
    const float* Tb = T + (size_t)n * MHW;
    float v = 0.f;
    for (int oy = max(y - 1, 0); oy <= min(y, MH - 1); ++oy) {
      int ylo, yhi; float ty;
      tap61(oy, kSy, 61, &ylo, &yhi, &ty);
      float wy = 0.f;
      if (ylo == y) wy += 1.f - ty;
      if (yhi == y) wy += ty;
      if (wy == 0.f) continue;
      for (int ox = max(x - 1, 0); ox <= min(x, MW - 1); ++ox) {
        int xlo, xhi; float tx;
        tap61(ox, kSx, 91, &xlo, &xhi, &tx);
        float wx = 0.f;
        if (xlo == x) wx += 1.f - tx;
        if (xhi == x) wx += tx;
        if (wx == 0.f) continue;
        const int pix = oy * MW + ox;
        v += wy * wx * (Gb[(size_t)pix * K] / Tb[pix]);
      }
    }
    return v;
  }
};

// out_t[n][v][u] = sum_{y, x} f_n[y][x] e^{-2 pi i (u y / 120 + v x / 180)},  v < 91, f nonzero on rows [R0, R0 + NR) x columns [C0, C0 + NC)
template <class Src>
__global__ __launch_bounds__(NT) void sm_lds_fwd_kernel(Src src, float2* __restrict__ out_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* cb = lds + CB;
  cf* rb = lds + RB;
  float* rbf = reinterpret_cast<float*>(rb);
  const int tid = threadIdx.x, n = blockIdx.x;
  make_twiddles(lds, tid);
  for (int i = tid; i < WC * PU; i += NT) cb[i] = cf{0.f, 0.f};
  constexpr int NPAIR = (Src::NR + 1) / 2;                               // row pairs (R0 + 2 i, R0 + 2 i + 1)
  constexpr int SWEEP = NPAIR <= NROWP ? NPAIR : (NPAIR + 1) / 2;        // pairs per sweep of the 31-row buffer
  static_assert(SWEEP <= NROWP && 2 * SWEEP >= NPAIR, "at most two sweeps");
  for (int i0 = 0; i0 < NPAIR; i0 += SWEEP) {
    const int np = min(SWEEP, NPAIR - i0);
    __syncthreads();      // (the previous sweep's unpacking has read the row buffer; the twiddles and the zeroed column buffer are visible)
    for (int i = tid; i < SWEEP * PX; i += NT) rb[i] = cf{0.f, 0.f};
    __syncthreads();
    for (int t = tid; t < 2 * np * Src::NC; t += NT) {
      const int x = t % Src::NC, r = t / Src::NC;                        // r = local row of the sweep
      const int y = 2 * i0 + r;
      if (y < Src::NR) rbf[((r >> 1) * PX + Src::C0 + x) * 2 + (r & 1)] = src.at(n, y, x);
    }
    __syncthreads();
    fft180<PX, -1, SWEEP>(rb, lds + TW180, tid);
    // Z_i = FFT(row_a + i row_b):  X_a[k] = (Z[k] + conj Z[-k]) / 2,  X_b[k] = (Z[k] - conj Z[-k]) / (2i)
    for (int t = tid; t < np * WC; t += NT) {
      const int i = t / WC, k = t - i * WC;
      const cf zk = rb[i * PX + pos180(k)], zn = rb[i * PX + pos180(k == 0 ? 0 : FW - k)];
      const int ya = Src::R0 + 2 * (i0 + i);
      cb[k * PU + ya] = cf{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
      if (ya + 1 < Src::R0 + Src::NR) cb[k * PU + ya + 1] = cf{0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
    }
  }
  __syncthreads();
  fft120<PU, -1, WC>(cb, lds + TW120, tid);
  cf* __restrict__ out = reinterpret_cast<cf*>(out_t) + (size_t)n * (WC * FH);
  for (int e = tid; e < WC * FH; e += NT) {
    const int v = e / FH, u = e - v * FH;
    out[e] = cb[v * PU + pos120(u)];
  }
}

// frames[n][y][x] = scale * sum_{u, v} S_n[v][u] e^{+2 pi i (u y / 120 + v x / 180)} (Hermitian extension in v) for y in [r0, r0 + nrows)
__global__ __launch_bounds__(NT) void sm_lds_inv_kernel(const float2* __restrict__ spec_t, float* __restrict__ frames, int r0, int nrows, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* cb = lds + CB;
  cf* rb = lds + RB;
  const float* rbf = reinterpret_cast<const float*>(rb);
  const int tid = threadIdx.x, n = blockIdx.x;
  make_twiddles(lds, tid);
  const cf* __restrict__ in = reinterpret_cast<const cf*>(spec_t) + (size_t)n * (WC * FH);
  for (int e = tid; e < WC * FH; e += NT) cb[e + e / FH] = scale * in[e];      // [v][u] with pitch 121
  __syncthreads();
  fft120<PU, 1, WC>(cb, lds + TW120, tid);
  constexpr int NG = NT / WC;
  const int pk = tid % WC, pg = tid / WC;
  const bool edge = pk == 0 || pk == FW / 2;      // DC / Nyquist columns: real by symmetry; a C2R transform ignores their imaginary parts
  float* __restrict__ dst = frames + (size_t)n * (FH * FW);
  const int npair = (nrows + 1) / 2;
  for (int i0 = 0; i0 < npair; i0 += NROWP) {
    const int np = min(NROWP, npair - i0);
    if (pg < NG) {
      for (int i = pg; i < np; i += NG) {
        const int ya = r0 + 2 * (i0 + i);
        cf xa = cb[pk * PU + pos120(ya)];
        cf xb = (ya + 1 < r0 + nrows) ? cb[pk * PU + pos120(ya + 1)] : cf{0.f, 0.f};
        if (edge) { xa.y = 0.f; xb.y = 0.f; }
        rb[i * PX + pk] = cf{xa.x - xb.y, xa.y + xb.x};
        if (!edge) rb[i * PX + FW - pk] = cf{xa.x + xb.y, xb.x - xa.y};
      }
    }
    __syncthreads();
    fft180<PX, 1, NROWP>(rb, lds + TW180, tid);
    for (int t = tid; t < 2 * np * FW; t += NT) {
      const int x = t % FW, r = t / FW;
      const int y = r0 + 2 * i0 + r;
      if (y < r0 + nrows) dst[(size_t)y * FW + x] = rbf[((r >> 1) * PX + pos180(x)) * 2 + (r & 1)];
    }
    __syncthreads();      // the row buffer is rewritten by the next sweep
  }
}

template <class Src>
hipError_t launch_fwd(const Src& src, float2* out_t, int n, hipStream_t st) {
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(sm_lds_fwd_kernel<Src>), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(sm_lds_fwd_kernel<Src>, dim3(n), dim3(NT), LDS_BYTES, st, src, out_t);
  return hipGetLastError();
}

}  // namespace

hipError_t sm_lds_fwd_frames(const float* frames, float2* spec_t, int n, hipStream_t st) {
  if (n < 1) return hipErrorInvalidValue;
  return launch_fwd(FullSrc{frames}, spec_t, n, st);
}

hipError_t sm_lds_fwd_dframes(const float* G, const float* T, float2* dhat_t, int nb, int K, int P, hipStream_t st) {
  if (nb < 1 || K < 1 || P % K) return hipErrorInvalidValue;
  return launch_fwd(DSrc{G, T, K, P}, dhat_t, nb * P, st);
}

hipError_t sm_lds_inv_frames(const float2* spec_t, float* frames, int n, int r0, int nrows, float scale, hipStream_t st) {
  if (n < 1 || r0 < 0 || nrows < 1 || r0 + nrows > FH) return hipErrorInvalidValue;
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(sm_lds_inv_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(sm_lds_inv_kernel, dim3(n), dim3(NT), LDS_BYTES, st, spec_t, frames, r0, nrows, scale);
  return hipGetLastError();
}

}  // namespace jcm
