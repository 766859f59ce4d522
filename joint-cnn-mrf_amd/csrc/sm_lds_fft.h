// The in-LDS 120 x 180 transforms of the spatial model (sm_fused.hip: the forward; sm_lds.hip: whole-frame forward / inverse transforms for the
// prior spectra, conv_mrf and the training step's backward pass).  A 120 x 91 half spectrum is 87 KB and lives in the LDS of one CU.
#pragma once
#include "fft_lds.h"
#include "kernels.h"

namespace jcm {
namespace smf {
using namespace fftl;      // cf (complex as a 2-vector), Dft<R, S>
constexpr int FH = 120, FW = 180, WC = 91;        // frame, half-spectrum columns
constexpr int MH = 60, MW = 90, MHW = MH * MW;    // heat map
constexpr int PU = 121, PX = 181;                 // LDS pitches (complex elements) of the column buffer [91][PU] and the row buffer [31][PX]
constexpr int NROWP = 31;                         // row pairs of the inverse (61 rows); the forward has 30
constexpr int NT = 768;                           // 12 waves: 91 x 8 radix-15 butterflies in one sweep; measured 1.21 ms per 256 images against 1.24 ms with 8 waves
constexpr int CB = 0, RB = WC * PU, TW120 = RB + NROWP * PX, TW180 = TW120 + FH, TY = TW180 + FW, TX = TY + MH, LDS_C = TX + MW;   // offsets in complex elements
constexpr int LDS_BYTES = LDS_C * 8;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NE = (WC * FH + NT - 1) / NT;       // spectrum elements per thread: 15
constexpr int NPIX = (MHW + NT - 1) / NT;         // output pixels per thread: 8

__device__ __forceinline__ int pos120(int y) { return (y & 7) * 15 + (y >> 3); }       // X[m + 8 k] sits at 15 m + k
__device__ __forceinline__ int pos180(int x) { return (x % 12) * 15 + x / 12; }          // X[m + 12 k] sits at 15 m + k

// One decimation-in-frequency stage of NB transforms of length N (row pitch PITCH), radix R on blocks of length L.
// tw[k] = e^{+2 pi i k / N}.  Consecutive threads take the same butterfly of consecutive transforms: the LDS stride is the
// (odd) pitch, which is bank-conflict free, and a wave reads its twiddles as broadcasts.
template <int N, int R, int L, int PITCH, int S, int NB>
__device__ __forceinline__ void fft_stage(cf* buf, const cf* tw, int tid) {
  constexpr int M = L / R, BF = N / R;
  for (int t = tid; t < BF * NB; t += NT) {
    const int bf = t / NB, v = t - bf * NB;
    const int blk = bf / M, k = bf - blk * M;
    cf* p = buf + v * PITCH + blk * L + k;
    cf x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = p[m * M];
    Dft<R, S>::run(x);
    if (M > 1) {
#pragma unroll
      for (int m = 1; m < R; ++m) {
        cf w = tw[(N / L) * k * m];
        if (S < 0) w.y = -w.y;
        x[m] = cmul(x[m], w);
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) p[m * M] = x[m];
  }
}
template <int PITCH, int S, int NB>
__device__ __forceinline__ void fft120(cf* buf, const cf* tw, int tid) {
  fft_stage<120, 8, 120, PITCH, S, NB>(buf, tw, tid); __syncthreads();
  fft_stage<120, 15, 15, PITCH, S, NB>(buf, tw, tid); __syncthreads();
}
template <int PITCH, int S, int NB>
__device__ __forceinline__ void fft180(cf* buf, const cf* tw, int tid) {
  fft_stage<180, 12, 180, PITCH, S, NB>(buf, tw, tid); __syncthreads();
  fft_stage<180, 15, 15, PITCH, S, NB>(buf, tw, tid); __syncthreads();
}

__device__ __forceinline__ void make_twiddles(cf* lds, int tid) {
  for (int k = tid; k < FH + FW; k += NT) {
    const bool a = k < FH;
    const int kk = a ? k : k - FH;
    double sn, cs;
    sincospi(2.0 * (double)kk / (double)(a ? FH : FW), &sn, &cs);
    lds[(a ? TW120 : TW180) + kk] = cf{(float)cs, (float)sn};
  }
}

__device__ __forceinline__ float softplus5(float x) {      // the spatial model's SoftPlus (beta 5, main.py:128-131), as sm_fft.hip
  const float z = 5.0f * x;
  const float thr = 13.942385f;
  float sp;
  if (z > thr) sp = z;
  else if (z < -thr) sp = expf(z);
  else sp = log1pf(expf(z));
  return 0.2f * sp;
}
__device__ __forceinline__ float lik_of(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld, const float* __restrict__ sc,
                                        const float* __restrict__ sh, int64_t pixg, int c) {
  const float hv = c < Ca ? hm[pixg * Ca + c] : extra[pixg * extra_ld + (c - Ca)];
  return sc ? softplus5(hv * sc[c] + sh[c]) : hv;
}
}  // namespace smf

}  // namespace jcm
