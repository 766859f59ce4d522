// Shared by the translation units of the frequency-domain convolution (conv_fft.hip: host side + filter spectra; conv_fft_rows_fwd.hip,
// conv_fft_rows_inv.hip, conv_fft_cols.hip: the transform kernels, one file per pass so that they compile in parallel): radix plans,
// the channel-vectorised in-LDS FFT stages, the bf16 operand split, and the size-dispatching launchers each file exports.
#pragma once
#include "fft_lds.h"
#include "kernels.h"

namespace jcm {
namespace cfft {
using namespace fftl;
constexpr int CB = 64, NT = 256;
constexpr int kParMax = 512;      // channels whose epilogue parameters the inverse row kernels keep in LDS

// radix chains (decimation in frequency, in place): the output X[n] sits at pos(n)
template <int N> struct Plan;
template <> struct Plan<192> { static constexpr int R1 = 8, R2 = 8, R3 = 3; };
template <> struct Plan<128> { static constexpr int R1 = 8, R2 = 4, R3 = 4; };
template <> struct Plan<100> { static constexpr int R1 = 4, R2 = 5, R3 = 5; };
template <> struct Plan<96> { static constexpr int R1 = 8, R2 = 12, R3 = 1; };
template <> struct Plan<72> { static constexpr int R1 = 8, R2 = 3, R3 = 3; };
template <> struct Plan<64> { static constexpr int R1 = 8, R2 = 8, R3 = 1; };
template <> struct Plan<60> { static constexpr int R1 = 4, R2 = 15, R3 = 1; };
template <> struct Plan<50> { static constexpr int R1 = 5, R2 = 10, R3 = 1; };
template <> struct Plan<40> { static constexpr int R1 = 8, R2 = 5, R3 = 1; };
template <> struct Plan<36> { static constexpr int R1 = 4, R2 = 3, R3 = 3; };
template <> struct Plan<32> { static constexpr int R1 = 8, R2 = 4, R3 = 1; };
template <> struct Plan<28> { static constexpr int R1 = 4, R2 = 7, R3 = 1; };
template <> struct Plan<24> { static constexpr int R1 = 8, R2 = 3, R3 = 1; };
template <> struct Plan<20> { static constexpr int R1 = 4, R2 = 5, R3 = 1; };
template <int N> __device__ __forceinline__ int pos(int n) {
  using P = Plan<N>;
  if constexpr (P::R3 == 1) return (n % P::R1) * (N / P::R1) + n / P::R1;
  else return (n % P::R1) * (N / P::R1) + ((n / P::R1) % P::R2) * (N / (P::R1 * P::R2)) + n / (P::R1 * P::R2);
}

// one stage over CH channel lanes: buf[position][channel], tw[k] = e^{+2 pi i k / N}
template <int N, int R, int L, int S, int CH, int NTH = NT>
__device__ __forceinline__ void stage(cf* buf, const cf* tw, int tid) {
  constexpr int M = L / R, BF = N / R;
  for (int t = tid; t < BF * CH; t += NTH) {
    const int bf = t / CH, v = t % CH;
    const int blk = bf / M, k = bf - blk * M;
    cf* p = buf + (blk * L + k) * CH + v;
    cf x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = p[m * M * CH];
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
    for (int m = 0; m < R; ++m) p[m * M * CH] = x[m];
  }
}
template <int N, int S, int CH, int NTH = NT>
__device__ __forceinline__ void fft(cf* buf, const cf* tw, int tid) {
  using P = Plan<N>;
  stage<N, P::R1, N, S, CH, NTH>(buf, tw, tid);
  __syncthreads();
  stage<N, P::R2, N / P::R1, S, CH, NTH>(buf, tw, tid);
  __syncthreads();
  if constexpr (P::R3 > 1) {
    stage<N, P::R3, N / (P::R1 * P::R2), S, CH, NTH>(buf, tw, tid);
    __syncthreads();
  }
}
// tw[k] = e^{+2 pi i k / N} from the per-device table (host-built in double precision; tw_offset(N) entries in)
template <int N, int NTH = NT>
__device__ __forceinline__ void twiddles(cf* tw, const cf* __restrict__ twg, int tid) {
  for (int k = tid; k < N; k += NTH) tw[k] = twg[k];
}


// ---- 16-bit row-transformed tensors ("t16": bf16 handles on the one-part route, np = 5).  T (between the forward row and column pass) and T'
// (between the inverse column and row pass) are half of the transform passes' HBM traffic as complex fp32.  Here they are complex FP16 in BLOCK
// FLOATING POINT: the work group that writes a tile -- (image, row, 64 channels, every kx) forward, (image, kx, channel block, every row) inverse --
// takes the tile's largest |component| m, multiplies by the power of two s with m s in [2^14, 2^15) (exact) and rounds to fp16 (11 significant
// bits, normal down to 2^-29 of the tile's maximum); 1 / s goes to one device word per tile, which the reading pass multiplies back in (exact).
// The layer's input and output are bf16 tensors (8 bits), the spectra on this route carry 11 bits already.
__device__ __forceinline__ float bfp_scale(float m) {
  int e = 0;
  if (!(m > 1.0e-30f && m < 3.0e38f)) return 1.f;      // zero / vanishing tiles (below 2^-100: their fp16 image is zero either way), inf, NaN: no scaling
  (void)frexpf(m, &e);      // m in [2^(e-1), 2^e)
  return ldexpf(1.f, 15 - e);      // e > -100: the scale and its inverse are normal fp32 numbers
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, static_cast<_Float16>(a)) | ((unsigned)__builtin_bit_cast(unsigned short, static_cast<_Float16>(b)) << 16);
}
__device__ __forceinline__ cf unpack_h2(unsigned u, float s) {
  return cf{static_cast<float>(__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu))) * s, static_cast<float>(__builtin_bit_cast(_Float16, (unsigned short)(u >> 16))) * s};
}
// the same in ONE instruction per component: v_fma_mix_f32 reads an fp16 half of a register as an fp32 operand (f16 -> f32 is exact, the scale a power of two,
// + 0).  _s: the scale is wave-uniform (a scalar register), _v: per thread.  The register row kernels are bound by their vector-ALU issue slots.
__device__ __forceinline__ cf unpack_h2_mix_s(unsigned u, float s) {
  cf r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(u), "s"(s));
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(u), "s"(s));
  return r;
}
__device__ __forceinline__ cf unpack_h2_mix_v(unsigned u, float s) {
  cf r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(u), "v"(s));
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(u), "v"(s));
  return r;
}
// the work group's maximum, known to every thread (one barrier; red[] must not be in use by a stash of the same tile)
template <int NTH>
__device__ __forceinline__ float block_max_all(float m, float* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < NTH / 64; ++w) r = fmaxf(r, red[w]);
  return r;
}
__device__ __forceinline__ cf bf16pair(unsigned bits) { return cf{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)}; }
// index, in float4 = two channels, of channel pair v (0..31) of 64-channel block cblk in T[kx][c/16][b][y][16]
__device__ __forceinline__ size_t t_fwd_index(int k, int cblk, int v, int b, int y, int B, int H, int C) {
  return ((((size_t)k * (C >> 4) + cblk * 4 + (v >> 3)) * B + b) * H + y) * 8 + (v & 7);
}
// Z = FFT(x_c + i x_{c+1}):  X_c[k] = (Z[k] + conj Z[-k]) / 2,  X_{c+1}[k] = (Z[k] - conj Z[-k]) / (2i)
// Returns the largest |component| this thread stored (the fp16-part GEMM of fp32 handles scales the spectra by it, see Fp16Scale).
template <int NX, int NTH = NT>
__device__ __forceinline__ float rows_fwd_store(const cf* buf, cf* __restrict__ T, int tid, int cblk, int b, int y, int B, int H, int C) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1;
  float4* dst = reinterpret_cast<float4*>(T);
  float m = 0.f;
  for (int t = tid; t < NXH * CH; t += NTH) {
    const int k = t / CH, v = t % CH;
    const cf zk = buf[pos<NX>(k) * CH + v], zn = buf[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
    const float4 o = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    dst[t_fwd_index(k, cblk, v, b, y, B, H, C)] = o;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  return m;
}

// The same tile as complex fp16 in block floating point (t16, bfp_scale above): T16[kx][c/16][b][y][16] of 4 bytes, tinv[(b C/64 + cblk) H + y] = 1 / scale.
// Returns the thread's largest |component| (of the fp32 values, for the per-image word of the spectra's scale).  One barrier inside.
template <int NX, int NTH = NT>
__device__ __forceinline__ float rows_fwd_store16(const cf* buf, void* T, float* __restrict__ tinv, float* red2, int tid, int cblk, int b, int y, int B, int H, int C) {
  constexpr int CH = CB / 2, NXH = NX / 2 + 1, K = (NXH * CH + NTH - 1) / NTH;
  uint2* dst = reinterpret_cast<uint2*>(T);
  float4 o[K];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int t = tid + i * NTH, k = t / CH, v = t % CH;
    o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < NXH * CH) {
      const cf zk = buf[pos<NX>(k) * CH + v], zn = buf[pos<NX>(k == 0 ? 0 : NX - k) * CH + v];
      o[i] = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y), 0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
      m = fmaxf(fmaxf(m, fmaxf(fabsf(o[i].x), fabsf(o[i].y))), fmaxf(fabsf(o[i].z), fabsf(o[i].w)));
    }
  }
  const float s = bfp_scale(block_max_all<NTH>(m, red2, tid));
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int t = tid + i * NTH, k = t / CH, v = t % CH;
    if (t < NXH * CH) dst[t_fwd_index(k, cblk, v, b, y, B, H, C)] = make_uint2(pack_h2(o[i].x * s, o[i].y * s), pack_h2(o[i].z * s, o[i].w * s));
  }
  if (tid == 0) tinv[((size_t)b * (C / CB) + cblk) * H + y] = 1.0f / s;      // [image][64-channel block][row]: contiguous over the rows a column thread reads
  return m;
}

// ---- fp32 handles, np = 4 (two FP16 parts per GEMM operand, cgemm_split.hip): fp16 carries 11 significant bits over 2^-24 .. 2^16, so the
// spectra are scaled by a power of two that is exact to apply and to undo.  tmax = the largest |re| or |im| of the row-transformed tensor T of an
// image (found by the row pass that writes T: an atomic max per tile on the image's device word, order independent: deterministic).  A column
// transform is X[ky] = sum over the H rows of T[y] e^{-i phi}: a COMPONENT of X is sum (re cos + im sin), bounded by H * tmax * (|cos| + |sin|)
// <= sqrt(2) * H * tmax.  The column pass scales by 2^k with H * tmax * 2^k < 2^15, so every scaled component is below sqrt(2) * 2^15 = 46341 --
// inside fp16's 65504 by the remaining factor sqrt(2) only: THE TARGET EXPONENT 15 IS THE LARGEST THAT IS SAFE (16 would overflow to inf).
// The inverse row pass multiplies its 1 / (NY NX) by 2^-k and by the inverse of the filter spectra's own scale.  (struct Fp16Scale: kernels.h)
constexpr int kFp16TargetExp = 15;
static_assert(kFp16TargetExp <= 15, "sqrt(2) * 2^kFp16TargetExp must stay below fp16's largest number 65504");
__device__ __forceinline__ int fp16_exp(float tmax, float hf) {      // e with H * tmax < 2^e (0 for an all-zero tensor)
  const float bound = tmax * hf;
  int e = 0;
  if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);
  return e;
}
// 2^k and 2^-k, k clamped to +-126 so that both are normal fp32 numbers whatever the tensor holds (a tensor whose largest entry is below
// 2^-111 would otherwise get an infinite scale and turn its zeros into NaN)
__device__ __forceinline__ int fp16_k(float tmax, float hf) { const int k = kFp16TargetExp - fp16_exp(tmax, hf); return k > 126 ? 126 : (k < -126 ? -126 : k); }
__device__ __forceinline__ float fp16_scale(float tmax, float hf) { return ldexpf(1.f, fp16_k(tmax, hf)); }
__device__ __forceinline__ float fp16_unscale(float tmax, float hf) { return ldexpf(1.f, -fp16_k(tmax, hf)); }
// max |T| the scale of image b derives from: its own word, or (common) the largest of the tensor's nb words
__device__ __forceinline__ float tmax_of(const float* __restrict__ tmax, int b, int nb, int common) {
  if (!common) return tmax[b];
  float m = 0.f;
  for (int i = 0; i < nb; ++i) m = fmaxf(m, tmax[i]);
  return m;
}
// Per tile of a persistent row kernel: every wave leaves its maximum in LDS in front of the barrier that ends the tile anyway; behind it one
// thread folds them and issues ONE atomic max on the image's device word (values are >= 0: their bit patterns order like unsigned integers).
// The atomic returns nothing, so nobody waits for it; red[] is next written a whole tile (several barriers) later.
__device__ __forceinline__ void wave_max_stash(float m, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
}
template <int NTH>
__device__ __forceinline__ void stash_to_word(const float* red, float* dst) {
  if (threadIdx.x == 0) {
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NTH / 64; ++w) m = fmaxf(m, red[w]);
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));
  }
}
// the work group's maximum -> the device word
template <int NTH>
__device__ __forceinline__ void block_max_to(float m, float* dst, float* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NTH / 64; ++w) m = fmaxf(m, red[w]);
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));      // always the atomic: a plain pre-read of the word could be a stale cache line
  }
}
// The row kernels keep NX x 32 complex numbers in LDS: threads per work group chosen so that the work groups the LDS admits fill the CU's 32 waves
template <int NX> constexpr int rows_threads() { return NX >= 128 ? 512 : NX >= 72 ? 384 : 256; }
template <int NY> constexpr int colblk() { return NY > 100 ? 32 : 64; }      // channels per work group of the inverse column kernel (<= 64 KB of LDS)
template <int NY> constexpr int colimg() { return NY > 96 ? 4 : 8; }         // images per work group (LDS: NY * IMG * 16 complex numbers)
// The column kernels hold big LDS tiles (two work groups per CU for the forward pass): they are latency-bound -- load, three barriers, store --
// unless the tile is worked on by many waves.  Threads per work group, so that a CU runs its full 32 waves:
template <int NY> constexpr int colfwd_threads() { return NY * colimg<NY>() * 16 >= 8192 ? 1024 : NY * colimg<NY>() * 16 >= 4096 ? 512 : 256; }
template <int NY> constexpr int colinv_threads() { return NY * colblk<NY>() >= 4096 ? 512 : 256; }
// two FP16 parts of x * scale (round to nearest even; 22 significant bits where the low part is a normal fp16 number)
__device__ __forceinline__ void split8h(const float (&x)[8], float scale, uint4 (&out)[2]) {
  unsigned short h[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = x[e] * scale;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const _Float16 q = static_cast<_Float16>(v);
      h[p][e] = __builtin_bit_cast(unsigned short, q);
      v = v - static_cast<float>(q);
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
    out[p] = make_uint4((unsigned)h[p][0] | ((unsigned)h[p][1] << 16), (unsigned)h[p][2] | ((unsigned)h[p][3] << 16), (unsigned)h[p][4] | ((unsigned)h[p][5] << 16),
                        (unsigned)h[p][6] | ((unsigned)h[p][7] << 16));
}
// ONE fp16 part of x * scale (round to nearest even; 11 significant bits): np = 5, bf16 handles
__device__ __forceinline__ uint4 round8h(const float (&x)[8], float scale) {
  unsigned short h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = __builtin_bit_cast(unsigned short, static_cast<_Float16>(x[e] * scale));
  return make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
}
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], uint4 (&out)[NP]) {
  unsigned short h[NP][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = x[e];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const __bf16 q = static_cast<__bf16>(v);           // round to nearest even
      h[p][e] = __builtin_bit_cast(unsigned short, q);
      v = v - static_cast<float>(q);                      // exact
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
    out[p] = make_uint4((unsigned)h[p][0] | ((unsigned)h[p][1] << 16), (unsigned)h[p][2] | ((unsigned)h[p][3] << 16), (unsigned)h[p][4] | ((unsigned)h[p][5] << 16),
                        (unsigned)h[p][6] | ((unsigned)h[p][7] << 16));
}
#define CFFT_BY_SIZE(N, CALL)                    \
  switch (N) {                                   \
    case 20: CALL(20); break;                    \
    case 24: CALL(24); break;                    \
    case 28: CALL(28); break;                    \
    case 32: CALL(32); break;                    \
    case 36: CALL(36); break;                    \
    case 40: CALL(40); break;                    \
    case 50: CALL(50); break;                    \
    case 60: CALL(60); break;                    \
    case 64: CALL(64); break;                    \
    case 72: CALL(72); break;                    \
    case 96: CALL(96); break;                    \
    case 100: CALL(100); break;                  \
    case 128: CALL(128); break;                  \
    default: CALL(192); break;                   \
  }

// grid of a persistent transform kernel: exactly the work groups the chip holds at once (a larger grid would run its excess as a
// second, mostly idle round), at most one per tile.  The residency comes from the occupancy query, cached per (kernel, device).
int persistent_grid(const void* kernel, int ntiles, int threads, int dyn_lds = 0);      // dyn_lds: the launch's dynamic LDS bytes

// ---- launchers (N = transform length, one of the lengths with a Plan); a.CoutP = output channels the inverse passes transform (Cout
// padded to 64), ldy = channel stride of the product spectra (Cout padded to the GEMM's N tile)
// sc: the fp16 scaling of np = 4 (tmax written by the forward row passes, read by the column pass and the inverse row passes); all null otherwise
// t16 (every launcher): the tile scale words of a 16-bit T / T' (null: complex fp32)
void cfft_rows_fwd(int NX, const ConvArgs& a, int layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16 = nullptr);
void cfft_rows_fwd_merge(int NX, const ConvArgs& a, const FftMerge& m, int in_layout, cf* T, const cf* tw, float* tmax, hipStream_t st, float* t16 = nullptr);      // in_layout 0 / 1: NHWC fp32 / bf16
hipError_t cfft_cols_fwd(int NY, const ConvArgs& a, int np, const cf* T, void* Xs, const cf* tw, int NXH, int MT, const Fp16Scale& sc, hipStream_t st);
// y16_inv (with t16 only): Yf holds complex fp16 products times a power of two (cgemm_split.hip); y16_inv = its inverse.  0: complex fp32
void cfft_cols_inv(int NY, const ConvArgs& a, const cf* Yf, cf* T, const cf* tw, int NXH, int ldy, int pad, hipStream_t st, float* t16 = nullptr, float y16_inv = 0.f);
void cfft_rows_inv(int NX, const ConvArgs& a, int layout, const cf* T, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
// the same pass with the transform in the registers of two threads per channel pair (conv_fft_rows_reg.hip); false: no such kernel for this case
bool cfft_rows_inv_reg(int NX, const ConvArgs& a, int layout, const cf* T, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
bool cfft_rows_inv_fwd_reg(int NX, const ConvArgs& a, const cf* T, cf* Tn, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
bool cfft_rows_fwd_reg(int NX, const ConvArgs& a, int layout, cf* T, float* tmax, hipStream_t st, float* t16);
// 32 x 32 overlap-save windows read straight from the map they are cut from (a.win_map, fp32 NHWC; Cin % 128 == 0): false = no such kernel
bool cfft_rows_fwd_win_reg(int NX, const ConvArgs& a, cf* T, float* tmax, hipStream_t st);
bool cfft_rows_fwd_win_reg_supported(int NX, int Cin);
bool cfft_rows_fwd_merge_reg(int NX, const ConvArgs& a, const FftMerge& m, int in_layout, cf* T, float* tmax, hipStream_t st, float* t16);
bool cfft_cols_inv_reg(int NY, const ConvArgs& a, const cf* Yf, cf* T, int NXH, int ldy, int pad, hipStream_t st, float* t16, float y16_inv = 0.f);
void cfft_rows_inv_fwd(int NX, const ConvArgs& a, const cf* T, cf* Tn, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
// conv_fft_rows_fused.hip: inverse rows + epilogue + 2x2 max pool + forward rows of the pooled map (NXO points, twiddles two) / + branch merge + forward rows.
// false: no kernel for this case
bool cfft_rows_inv_pool_fwd(int NXI, int NXO, const ConvArgs& a, const cf* T, cf* Tn, const cf* twi, const cf* two, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
bool cfft_rows_inv_merge_fwd(int NX, const ConvArgs& a, const FftMerge& m, const cf* T, cf* Tn, const cf* tw, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
// conv_fft_rows_reg.hip: the model's geometry in registers; t16n != null: bf16 handles (16-bit T' in, 16-bit T + its scale words t16n out, bf16 coarse branches)
bool cfft_rows_inv_merge_fwd_reg(int NX, const ConvArgs& a, const FftMerge& m, const cf* T, cf* Tn, int pad, float norm, const Fp16Scale& sc, hipStream_t st, float* t16n = nullptr);
bool cfft_rows_inv_merge_fwd_reg_supported(int NX, const ConvArgs& a, const FftMerge& m, int pad);
bool cfft_rows_inv_pool_fwd_supported(int NXI, int NXO, int Cout);
// conv_fft_rows_mfma.hip: the 96-point inverse row pass of bf16 handles (16-bit T', planar bf16 output) as a matrix product on the matrix cores; false: no kernel for this case
bool cfft_rows_inv_mfma(int NX, const ConvArgs& a, int layout, const cf* T, int pad, float norm, const Fp16Scale& sc, hipStream_t st);
bool cfft_rows_inv_merge_fwd_supported(int NX, const ConvArgs& a, const FftMerge& m);
}  // namespace cfft
}  // namespace jcm
