// FFTs held entirely in the registers of ONE thread (conv_fft_rows_reg.hip): every index is a compile-time constant, so there is no LDS
// round trip, no barrier, no index arithmetic and no twiddle load -- the instruction stream of a transform is its butterflies.  (The LDS kernels
// of conv_fft_common.h spend 70-80 % of their instructions around the butterflies and run at 2-3 work groups per CU)
//   N = R1 * R2, n = R2 n1 + n2, k = k1 + R1 k2:   X[k] = sum_n2 w_N^(S n2 k1) [sum_n1 x[R2 n1 + n2] w_R1^(S n1 k1)] w_R2^(S n2 k2)
// step 1 (in place, per n2): R1-point DFT over n1, times the twiddle;  step 2 (per k1): R2-point DFT over n2, handed to a visitor as it is
// produced (the caller stores or keeps it).  Radix butterflies: fft_lds.h.
#pragma once
#include "fft_lds.h"

namespace jcm {
namespace fftr {
using namespace fftl;

// sin / cos of 2 pi num / den as compile-time constants (Taylor series on the argument reduced to [-pi/4, pi/4] by octant symmetry; double
// precision, error < 1e-15, rounded once to float)
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double sin_small(double x) {      // |x| <= pi / 4
  double term = x, sum = x;
  for (int i = 1; i < 12; ++i) { term *= -x * x / ((2 * i) * (2 * i + 1)); sum += term; }
  return sum;
}
constexpr double cos_small(double x) {
  double term = 1.0, sum = 1.0;
  for (int i = 1; i < 12; ++i) { term *= -x * x / ((2 * i - 1) * (2 * i)); sum += term; }
  return sum;
}
struct SinCos { double s, c; };
constexpr SinCos sincos_frac(long num, long den) {      // angle = 2 pi num / den
  num %= den;
  if (num < 0) num += den;
  // octant: angle = o * pi/4 + r, r in [0, pi/4)
  const long n8 = num * 8;
  const long o = n8 / den;
  const double r = 2.0 * kPi * (double)(n8 - o * den) / (8.0 * (double)den);
  const double sr = sin_small(r), cr = cos_small(r), h = 0.70710678118654752440084436210485;
  switch (o) {
    case 0: return {sr, cr};
    case 1: return {h * (cr + sr), h * (cr - sr)};
    case 2: return {cr, -sr};
    case 3: return {h * (cr - sr), -h * (cr + sr)};
    case 4: return {-sr, -cr};
    case 5: return {-h * (cr + sr), -h * (cr - sr)};
    case 6: return {-cr, sr};
    default: return {-h * (cr - sr), h * (cr + sr)};
  }
}
template <int NUM, int DEN> struct Tw {      // e^{2 pi i NUM / DEN}
  static constexpr float re = (float)sincos_frac(NUM, DEN).c;
  static constexpr float im = (float)sincos_frac(NUM, DEN).s;
};
// a * e^{2 pi i NUM / DEN}; multiples of a quarter turn cost nothing
template <int NUM, int DEN> __device__ __forceinline__ cf twmul(cf a) {
  constexpr int n = ((NUM % DEN) + DEN) % DEN;
  if constexpr (n == 0) return a;
  else if constexpr (4 * n == DEN) return cf{-a.y, a.x};
  else if constexpr (2 * n == DEN) return cf{-a.x, -a.y};
  else if constexpr (4 * n == 3 * DEN) return cf{a.y, -a.x};
  else {
    constexpr float wr = Tw<n, DEN>::re, wi = Tw<n, DEN>::im;
    return cf{fmaf(-a.y, wi, a.x * wr), fmaf(a.x, wi, a.y * wr)};
  }
}

template <int N> struct RPlan;
template <> struct RPlan<96> { static constexpr int R1 = 8, R2 = 12; };
template <> struct RPlan<64> { static constexpr int R1 = 8, R2 = 8; };
template <> struct RPlan<48> { static constexpr int R1 = 4, R2 = 12; };
template <> struct RPlan<32> { static constexpr int R1 = 4, R2 = 8; };
template <> struct RPlan<16> { static constexpr int R1 = 4, R2 = 4; };
template <> struct RPlan<25> { static constexpr int R1 = 5, R2 = 5; };
template <> struct RPlan<14> { static constexpr int R1 = 2, R2 = 7; };
template <> struct RPlan<50> { static constexpr int R1 = 5, R2 = 10; };
template <> struct RPlan<36> { static constexpr int R1 = 3, R2 = 12; };
template <> struct RPlan<28> { static constexpr int R1 = 4, R2 = 7; };
template <> struct RPlan<20> { static constexpr int R1 = 4, R2 = 5; };

// x[R2 k1 + N2] = a[k1] * w_N^(S N2 k1), k1 = K1 .. R1-1 (compile-time recursion: every twiddle is a literal)
template <int N, int S, int N2, int K1 = 0>
__device__ __forceinline__ void step1_twiddle(cf (&x)[N], const cf (&a)[RPlan<N>::R1]) {
  constexpr int R1 = RPlan<N>::R1, R2 = RPlan<N>::R2;
  x[R2 * K1 + N2] = twmul<S * N2 * K1, N>(a[K1]);
  if constexpr (K1 + 1 < R1) step1_twiddle<N, S, N2, K1 + 1>(x, a);
}
// step 1 for n2 = N2 .. R2-1
template <int N, int S, int N2 = 0>
__device__ __forceinline__ void step1(cf (&x)[N]) {
  constexpr int R1 = RPlan<N>::R1, R2 = RPlan<N>::R2;
  cf a[R1];
#pragma unroll
  for (int n1 = 0; n1 < R1; ++n1) a[n1] = x[R2 * n1 + N2];
  Dft<R1, S>::run(a);
  step1_twiddle<N, S, N2>(x, a);
  if constexpr (N2 + 1 < R2) step1<N, S, N2 + 1>(x);
}
// step 2 for one k1: the R2 outputs X[k1 + R1 k2], k2 = 0 .. R2-1, in out[k2]
template <int N, int S, int K1>
__device__ __forceinline__ void step2_row(const cf (&x)[N], cf (&out)[RPlan<N>::R2]) {
  constexpr int R2 = RPlan<N>::R2;
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) out[n2] = x[R2 * K1 + n2];
  Dft<R2, S>::run(out);
}

}  // namespace fftr
}  // namespace jcm
