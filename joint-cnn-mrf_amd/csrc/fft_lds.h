// Complex butterflies shared by the in-LDS FFT kernels (sm_fused.hip, conv_fft.hip).  A complex number is a 2-vector (the compiler
// keeps it in a register pair; the arithmetic is scalar fp32 instructions, see cmac below).  Dft<R, S>::run(x): y_k = sum_m x_m e^{S 2 pi i m k / R},
// in place (S = +1 inverse, -1 forward).
#pragma once
#include <hip/hip_runtime.h>

namespace jcm {
namespace fftl {
typedef float cf __attribute__((ext_vector_type(2)));

// Streaming store (nontemporal hint): for tensors that are written once and read by the NEXT kernel after gigabytes of other traffic -- the lines need not
// stay in the write-back L2, where they would push out the stream the kernel is reading.  (Round 4: the filter-spectra packer's 6.6 GB of writes went
// from 3.1 to 4.5 TB/s with it; used where an A/B of the whole step showed a gain: both column passes, the fp32 inverse row pass, the fp32 handles' GEMM --
// not the forward row passes, the staged bf16 rows, or the one-part GEMM, which measured 3-12 % slower with it.)
__device__ __forceinline__ void st_stream(cf* p, cf v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(unsigned* p, unsigned v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) { typedef unsigned u2 __attribute__((ext_vector_type(2))); __builtin_nontemporal_store(u2{v.x, v.y}, reinterpret_cast<u2*>(p)); }
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) { typedef unsigned u4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(p)); }
__device__ __forceinline__ void st_stream(float4* p, float4 v) { typedef float f4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p)); }

__device__ __forceinline__ cf cfma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cf sfma(float a, cf b, cf c) { return __builtin_elementwise_fma(cf{a, a}, b, c); }
template <int S> __device__ __forceinline__ cf muli(cf a) { return S > 0 ? cf{-a.y, a.x} : cf{a.y, -a.x}; }   // a * (S i)
__device__ __forceinline__ cf cmul(cf a, cf b) { return cfma(a.yy, cf{-b.y, b.x}, a.xx * b); }
// acc + a * b (complex), and a + (S i) b.  Written per component: this library is compiled WITHOUT packed-fp32 instructions (Makefile:
// -target-feature -packed-fp32-ops).  Rounds 1-3 had these two as hand-written v_pk_fma_f32 / v_pk_add_f32 with operand selects; round 4 found
// that the RESULT OF A PACKED-FP32 INSTRUCTION (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, compiler-generated or hand-written) can come out
// wrong in lanes 48-63 of a wave while a wave of one of this library's MFMA kernels (LDS fragment reads feeding v_mfma) is resident on the
// same CU -- which only happens when kernels of two streams / processes share the GPU.  tools/coresidency_probe.hip reproduces it against a
// plain kernel (packed arithmetic: 210-283 of 300 runs differ beside the channel GEMM; the same arithmetic on scalar instructions: 0 of
// 300); with the whole library on scalar fp32 instructions the two-engine soak without the call chain is clean (0 of 400 forwards, fp32 and
// bf16; before: 250-505 of 600 and 14 of 600) at unchanged speed (the transform kernels are bound by HBM and LDS, not by VALU issue).
__device__ __forceinline__ cf cmac(cf acc, cf a, cf b) { return cfma(a.yy, cf{-b.y, b.x}, cfma(a.xx, b, acc)); }
template <int S> __device__ __forceinline__ cf add_i(cf a, cf b) { return S > 0 ? cf{a.x - b.y, a.y + b.x} : cf{a.x + b.y, a.y - b.x}; }

// r-point DFTs, y_k = sum_m x_m e^{S 2 pi i m k / r} (S = +1 inverse, -1 forward), in place
template <int R, int S> struct Dft;
template <int S> struct Dft<2, S> {
  static __device__ __forceinline__ void run(cf (&x)[2]) {
    const cf a = x[0] + x[1], b = x[0] - x[1];
    x[0] = a; x[1] = b;
  }
};
template <int S> struct Dft<3, S> {
  static __device__ __forceinline__ void run(cf (&x)[3]) {
    const cf t1 = x[1] + x[2];
    const cf t2 = sfma(-0.5f, t1, x[0]);
    const cf t3 = 0.86602540378443865f * (x[1] - x[2]);
    x[0] = x[0] + t1;
    x[1] = add_i<S>(t2, t3);
    x[2] = add_i<-S>(t2, t3);
  }
};
template <int S> struct Dft<4, S> {
  static __device__ __forceinline__ void run(cf (&x)[4]) {
    const cf a = x[0] + x[2], b = x[0] - x[2], c = x[1] + x[3], d = x[1] - x[3];
    x[0] = a + c; x[2] = a - c; x[1] = add_i<S>(b, d); x[3] = add_i<-S>(b, d);
  }
};
template <int S> struct Dft<5, S> {
  static __device__ __forceinline__ void run(cf (&x)[5]) {
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const cf t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
    const cf a1 = sfma(c2, t2, sfma(c1, t1, x[0]));
    const cf a2 = sfma(c1, t2, sfma(c2, t1, x[0]));
    const cf b1 = sfma(s2, t4, s1 * t3);
    const cf b2 = sfma(-s1, t4, s2 * t3);
    x[0] = x[0] + (t1 + t2);
    x[1] = add_i<S>(a1, b1); x[4] = add_i<-S>(a1, b1);
    x[2] = add_i<S>(a2, b2); x[3] = add_i<-S>(a2, b2);
  }
};
template <int S> struct Dft<8, S> {
  static __device__ __forceinline__ void run(cf (&x)[8]) {
    cf e[4] = {x[0], x[2], x[4], x[6]}, o[4] = {x[1], x[3], x[5], x[7]};
    Dft<4, S>::run(e);
    Dft<4, S>::run(o);
    constexpr float h = 0.70710678118654752f;
    o[1] = h * add_i<S>(o[1], o[1]);            // * (1 + S i) / sqrt 2
    o[3] = h * add_i<S>(-o[3], o[3]);           // * (-1 + S i) / sqrt 2
    x[0] = e[0] + o[0]; x[4] = e[0] - o[0];
    x[1] = e[1] + o[1]; x[5] = e[1] - o[1];
    x[2] = add_i<S>(e[2], o[2]); x[6] = add_i<-S>(e[2], o[2]);      // o[2] * (S i)
    x[3] = e[3] + o[3]; x[7] = e[3] - o[3];
  }
};
// Prime-factor (Good-Thomas) compositions: n = (R2 n1 + R1 n2) mod R1 R2 in, k = (A k1 + B k2) mod R1 R2 out with k = k1 (mod R1),
// k = k2 (mod R2); the exponent n k then splits into n1 k1 / R1 + n2 k2 / R2 exactly: no twiddles.
template <int R1, int R2, int A, int B, int S>
__device__ __forceinline__ void dft_pfa(cf (&x)[R1 * R2]) {
  constexpr int R = R1 * R2;
  cf b[R1][R2];
#pragma unroll
  for (int n1 = 0; n1 < R1; ++n1) {
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) b[n1][n2] = x[(R2 * n1 + R1 * n2) % R];
    Dft<R2, S>::run(b[n1]);
  }
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) {
    cf c[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) c[n1] = b[n1][k2];
    Dft<R1, S>::run(c);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) x[(A * k1 + B * k2) % R] = c[k1];
  }
}
// 7 points: the three conjugate pairs (m, 7 - m) as sums and differences, 18 real-by-complex multiply-adds each way
template <int S> struct Dft<7, S> {
  static __device__ __forceinline__ void run(cf (&x)[7]) {
    constexpr float c1 = 0.62348980185873353f, c2 = -0.22252093395631440f, c3 = -0.90096886790241913f;
    constexpr float s1 = 0.78183148246802981f, s2 = 0.97492791218182361f, s3 = 0.43388373911755812f;
    const cf p1 = x[1] + x[6], p2 = x[2] + x[5], p3 = x[3] + x[4];
    const cf q1 = x[1] - x[6], q2 = x[2] - x[5], q3 = x[3] - x[4];
    // y_k = x0 + sum_m cos(2 pi m k / 7) p_m  +-  S i sum_m sin(2 pi m k / 7) q_m
    const cf a1 = sfma(c3, p3, sfma(c2, p2, sfma(c1, p1, x[0])));
    const cf a2 = sfma(c1, p3, sfma(c3, p2, sfma(c2, p1, x[0])));
    const cf a3 = sfma(c2, p3, sfma(c1, p2, sfma(c3, p1, x[0])));
    const cf b1 = sfma(s3, q3, sfma(s2, q2, s1 * q1));
    const cf b2 = sfma(-s1, q3, sfma(-s3, q2, s2 * q1));     // sin(2 pi 2 m / 7), m = 1, 2, 3: s2, -s3, -s1
    const cf b3 = sfma(s2, q3, sfma(-s1, q2, s3 * q1));      // sin(2 pi 3 m / 7): s3, -s1, s2
    x[0] = x[0] + (p1 + (p2 + p3));
    x[1] = add_i<S>(a1, b1); x[6] = add_i<-S>(a1, b1);
    x[2] = add_i<S>(a2, b2); x[5] = add_i<-S>(a2, b2);
    x[3] = add_i<S>(a3, b3); x[4] = add_i<-S>(a3, b3);
  }
};
template <int S> struct Dft<15, S> { static __device__ __forceinline__ void run(cf (&x)[15]) { dft_pfa<3, 5, 10, 6, S>(x); } };
template <int S> struct Dft<12, S> { static __device__ __forceinline__ void run(cf (&x)[12]) { dft_pfa<3, 4, 4, 9, S>(x); } };
template <int S> struct Dft<10, S> { static __device__ __forceinline__ void run(cf (&x)[10]) { dft_pfa<2, 5, 5, 6, S>(x); } };    // k = 5 k1 + 6 k2
template <int S> struct Dft<14, S> { static __device__ __forceinline__ void run(cf (&x)[14]) { dft_pfa<2, 7, 7, 8, S>(x); } };    // k = 7 k1 + 8 k2

}  // namespace fftl
}  // namespace jcm
