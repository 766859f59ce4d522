// Stand-alone check + timing of the channel GEMM (joint-cnn-mrf_amd/csrc/cgemm_split.hip) outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I joint-cnn-mrf_amd/csrc tools/cgemm_probe.hip -o tools/cgemm_probe
//   tools/cgemm_probe <np> <F> <B> <Cin> <Cout> [iters]
// Operands are a hash of their index (the same on host and device), split on the device into the kernel's tile-major layouts by
// two naive packing kernels; sampled outputs are compared with a float64 host sum of the SAME fp32 values.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../joint-cnn-mrf_amd/csrc/cgemm_split.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__host__ __device__ inline float hval(unsigned long long i, unsigned salt) {
  unsigned long long z = i * 0x9E3779B97F4A7C15ull + salt * 0xD1B54A32D192ED03ull;
  z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
  return (float)((double)(z & 0xffffff) / 8388608.0 - 1.0);      // [-1, 1), 24 random bits: a full fp32 mantissa
}
__host__ __device__ inline float xval(size_t f, int b, int ci, int c, int B, int Cin) { return hval(((f * B + b) * Cin + ci) * 2 + c, 1); }
__host__ __device__ inline float wval(size_t f, int ci, int co, int c, int Cin, int Cout) { return hval(((f * Cin + ci) * Cout + co) * 2 + c, 2) * 0.05f; }

// np = 4: two FP16 parts of the value scaled into fp16's range (what the transform passes do, conv_fft.hip)
#define PARTS(np) ((np) == 4 ? 2 : (np) == 5 ? 1 : (np))
#define SX(np) ((np) >= 4 ? 1024.f : 1.f)
#define SW(np) ((np) >= 4 ? 16384.f : 1.f)
__device__ inline void split_store(float v, int np, __bf16* dst, size_t pstride) {
  for (int p = 0; p < PARTS(np); ++p) {
    if (np >= 4) {
      const _Float16 q = static_cast<_Float16>(v);
      reinterpret_cast<_Float16*>(dst)[p * pstride] = q;
      v = v - static_cast<float>(q);
    } else {
      const __bf16 q = static_cast<__bf16>(v);
      dst[p * pstride] = q;
      v = v - static_cast<float>(q);
    }
  }
}
// one thread per (f, row, ci)
__global__ void pack_x(__bf16* xs, int np, int F, int B, int Cin, int MT, int mtiles) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)F * mtiles * MT * Cin;
  if (i >= n) return;
  const int ci = (int)(i % Cin);
  const int row = (int)((i / Cin) % (mtiles * MT));
  const size_t f = i / ((size_t)Cin * mtiles * MT);
  const int mt = row / MT, r = row % MT, kc = ci / 16, kg = (ci % 16) / 8, e = ci % 8, KC = Cin / 16;
  for (int c = 0; c < 2; ++c) {
    const float v = row < B ? xval(f, row, ci, c, B, Cin) * SX(np) : 0.f;
    const int npp = PARTS(np);
    __bf16* dst = np == 5 ? xs + (((((f * mtiles + mt) * (KC / 2) + kc / 2) * 8 + (size_t)(c * 2 + (kc & 1)) * 2 + kg) * MT + r) * 8 + e)
                          : xs + (((((f * mtiles + mt) * KC + kc) * (4 * npp) + (size_t)(c * npp) * 2 + kg) * MT + r) * 8 + e);
    split_store(v, np, dst, (size_t)2 * MT * 8);
  }
}
// ntl-column tiles; f32: unsplit fp32 layout [f][nt][kc][c][kg][4-channel half][col][4]
__global__ void pack_w(void* ws, int np, int f32, int ntl, int F, int Cin, int Cout, int CoutP) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)F * Cin * CoutP;
  if (i >= n) return;
  const int co = (int)(i % CoutP);
  const int ci = (int)((i / CoutP) % Cin);
  const size_t f = i / ((size_t)CoutP * Cin);
  const int nt = co / ntl, cn = co % ntl, kc = ci / 16, kg = (ci % 16) / 8, e = ci % 8, KC = Cin / 16, ntiles = CoutP / ntl;
  for (int c = 0; c < 2; ++c) {
    const float v = co < Cout ? wval(f, ci, co, c, Cin, Cout) * SW(np) : 0.f;
    if (f32) {
      static_cast<float*>(ws)[((((((f * ntiles + nt) * KC + kc) * 2 + c) * 2 + kg) * 2 + e / 4) * ntl + cn) * 4 + e % 4] = v;
    } else {
      const int npp = PARTS(np);
      __bf16* dst = np == 5 ? static_cast<__bf16*>(ws) + (((((f * ntiles + nt) * (KC / 2) + kc / 2) * 8 + (size_t)(c * 2 + (kc & 1)) * 2 + kg) * ntl + cn) * 8 + e)
                            : static_cast<__bf16*>(ws) + (((((f * ntiles + nt) * KC + kc) * (4 * npp) + (size_t)(c * npp) * 2 + kg) * ntl + cn) * 8 + e);
      split_store(v, np, dst, (size_t)2 * ntl * 8);
    }
  }
}

// leaves NaN patterns in LDS and in the vector registers of every CU it visits (a wave that reads LDS or registers it never
// wrote sees what the previous occupant left: its own kind when the kernel runs alone, this when it shares the chip)
__global__ __launch_bounds__(256) void noise_kernel(float* sink, int rounds) {
  extern __shared__ unsigned nlds[];
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) nlds[i] = 0x7fc00000u + i;
  float v[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) v[i] = __uint_as_float(0x7fc00000u + i + threadIdx.x);
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 96; ++i) { v[i] = v[i] * 1.0001f + (float)r; acc += v[i]; }
    __syncthreads();
    acc += __uint_as_float(nlds[(threadIdx.x * 17 + r) & 16383]);
  }
  if (acc == 12345.f) sink[0] = acc;
}

// a victim: keeps a known pattern in 33 KB of LDS for a while and counts words that changed under it (somebody else's out-of-bounds LDS write)
__global__ __launch_bounds__(512) void lds_guard_kernel(unsigned long long* bad, int rounds) {
  __shared__ unsigned g[33 * 256];
  for (int i = threadIdx.x; i < 33 * 256; i += 512) g[i] = 0xabc00000u ^ (i * 2654435761u);
  __syncthreads();
  unsigned long long n = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < 33 * 256; i += 512) n += g[i] != (0xabc00000u ^ (i * 2654435761u));
    __builtin_amdgcn_s_sleep(20);
  }
  if (n) atomicAdd(bad, n);
}

// producer -> consumer over a kernel boundary on one stream: the producer overwrites a buffer with a pattern that depends on the round,
// the consumer (another kernel, same stream) counts words that are not the pattern.  Same access shapes as the column / row passes:
// 8-byte stores of a wave = 512 contiguous bytes, 16-byte loads.
__global__ __launch_bounds__(512) void chain_write(float2* buf, size_t n, unsigned round) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = make_float2(__uint_as_float((unsigned)i * 2654435761u + round), __uint_as_float((unsigned)i ^ round));
}
__global__ __launch_bounds__(384) void chain_check(const float4* buf, size_t n2, unsigned round, unsigned long long* bad) {
  unsigned long long nb = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = buf[i];
    const unsigned i0 = (unsigned)(2 * i), i1 = i0 + 1;
    nb += __float_as_uint(v.x) != i0 * 2654435761u + round;
    nb += __float_as_uint(v.y) != (i0 ^ round);
    nb += __float_as_uint(v.z) != i1 * 2654435761u + round;
    nb += __float_as_uint(v.w) != (i1 ^ round);
  }
  if (nb) atomicAdd(bad, nb);
}

int main(int argc, char** argv) {
  const int np = argc > 1 ? atoi(argv[1]) : 2, F = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 256;
  const int Cin = argc > 4 ? atoi(argv[4]) : 512, Cout = argc > 5 ? atoi(argv[5]) : 512, iters = argc > 6 ? atoi(argv[6]) : 5;
  const int ntl = jcm::cgemm_split_ntile(np, Cout), f32 = jcm::cgemm_split_w_fp32(np) ? 1 : 0;
  const int CoutP = (Cout + ntl - 1) / ntl * ntl;
  const int MT = jcm::cgemm_split_mtile(np, B, Cout), mtiles = (B + MT - 1) / MT;
  const size_t xbytes = (size_t)F * mtiles * MT * Cin * 4 * PARTS(np), wbytes = jcm::cgemm_split_w_bytes(np, F, Cin, Cout), ybytes = (size_t)F * B * CoutP * 8;
  void *xs, *ws, *y;
  CK(hipMalloc(&xs, xbytes)); CK(hipMalloc(&ws, wbytes)); CK(hipMalloc(&y, ybytes));
  CK(hipMemset(y, 0xff, ybytes));
  {
    const size_t nx = (size_t)F * mtiles * MT * Cin, nw = (size_t)F * Cin * CoutP;
    hipLaunchKernelGGL(pack_x, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, (__bf16*)xs, np, F, B, Cin, MT, mtiles);
    hipLaunchKernelGGL(pack_w, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, ws, np, f32, ntl, F, Cin, Cout, CoutP);
    CK(hipDeviceSynchronize());
  }
  CK(jcm::cgemm_split(xs, ws, y, np, F, B, Cin, Cout, CoutP, 0));
  CK(hipDeviceSynchronize());
  // ---- check sampled outputs
  double worst = 0, scale = 0;
  int bad = 0;
  for (int s = 0; s < 400; ++s) {
    const size_t f = (size_t)((s * 7919u) % F);
    const int b = s < 8 ? (s & 1 ? B - 1 : 0) : (int)((s * 104729u) % B);
    const int co = s < 8 ? (s & 2 ? Cout - 1 : 0) : (int)((s * 1299709u) % Cout);
    double yr = 0, yi = 0, mag = 0;
    for (int ci = 0; ci < Cin; ++ci) {
      const double xr = xval(f, b, ci, 0, B, Cin), xi = xval(f, b, ci, 1, B, Cin), wr = wval(f, ci, co, 0, Cin, Cout), wi = wval(f, ci, co, 1, Cin, Cout);
      yr += xr * wr - xi * wi;
      yi += xr * wi + xi * wr;
      mag += fabs(xr * wr) + fabs(xi * wi);
    }
    float2 got;
    CK(hipMemcpy(&got, (char*)y + ((f * B + b) * CoutP + co) * 8, 8, hipMemcpyDeviceToHost));
    got.x /= SX(np) * SW(np); got.y /= SX(np) * SW(np);
    const double err = fmax(fabs(got.x - yr), fabs(got.y - yi));
    const double rms = sqrt((double)Cin) * 0.05 * 0.577 * 0.577 * 1.414;      // typical |y|
    if (!(err <= (np == 5 ? 2e-3 : np >= 3 ? 6e-6 : 1e-4) * rms)) { if (bad < 5) printf("MISMATCH f=%zu b=%d co=%d got (%g,%g) want (%g,%g)\n", f, b, co, got.x, got.y, yr, yi); ++bad; }
    worst = fmax(worst, err / rms);
    scale = rms;
  }
  printf("np=%d F=%d B=%d Cin=%d Cout=%d MT=%d: worst error / typical |y| = %.3g (%d bad of 400; |y| ~ %.3g)\n", np, F, B, Cin, Cout, MT, worst, bad, scale);
  // ---- two streams at once: the same product into two buffers, repeatedly; every result must equal the one computed alone
  if (argc > 7) {
    void* y2;
    CK(hipMalloc(&y2, ybytes));
    std::vector<char> ref(ybytes), got(ybytes);
    CK(hipMemcpy(ref.data(), y, ybytes, hipMemcpyDeviceToHost));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int wrong = 0;
    for (int it = 0; it < 10; ++it) {
      CK(hipMemsetAsync(y, 0xff, ybytes, s1)); CK(hipMemsetAsync(y2, (argv[7][0] == 'g' || argv[7][0] == 'p') ? 0 : 0xff, argv[7][0] == 'p' ? 4096 : ybytes, s2));
      for (int k = 0; k < 3; ++k) {
        CK(jcm::cgemm_split(xs, ws, y, np, F, B, Cin, Cout, CoutP, s1));
        if (argv[7][0] == 'n') hipLaunchKernelGGL(noise_kernel, dim3(4096), dim3(256), 64 * 1024, s2, (float*)y2, 40);
        else if (argv[7][0] == 'g') hipLaunchKernelGGL(lds_guard_kernel, dim3(8192), dim3(512), 0, s2, (unsigned long long*)y2, 300);
        else if (argv[7][0] == 'p') {      // a chain of 6 producer / consumer pairs beside the GEMM; the first 8 bytes of y2 count bad words
          const size_t n = (ybytes - 4096) / 8;
          for (int r = 0; r < 6; ++r) {
            hipLaunchKernelGGL(chain_write, dim3(2048), dim3(512), 0, s2, (float2*)((char*)y2 + 4096), n, (unsigned)(it * 100 + k * 10 + r));
            hipLaunchKernelGGL(chain_check, dim3(1024), dim3(384), 0, s2, (const float4*)((char*)y2 + 4096), n / 2, (unsigned)(it * 100 + k * 10 + r), (unsigned long long*)y2);
          }
        }
        else CK(jcm::cgemm_split(xs, ws, y2, np, F, B, Cin, Cout, CoutP, s2));
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), y, ybytes, hipMemcpyDeviceToHost));
      size_t d1 = 0, d2 = 0;
      for (size_t i = 0; i < ybytes / 8; ++i) d1 += memcmp(&got[i * 8], &ref[i * 8], 8) != 0;
      CK(hipMemcpy(got.data(), y2, ybytes, hipMemcpyDeviceToHost));
      if (argv[7][0] == 'g' || argv[7][0] == 'p') d2 = *reinterpret_cast<unsigned long long*>(got.data());      // LDS words the guard kernel saw change
      else if (argv[7][0] != 'n') for (size_t i = 0; i < ybytes / 8; ++i) d2 += memcmp(&got[i * 8], &ref[i * 8], 8) != 0;
      printf("  concurrent run %d: %zu / %zu outputs differ on stream 1, %zu on stream 2\n", it, d1, ybytes / 8, d2);
      wrong += d1 + d2 != 0;
    }
    return wrong ? 3 : 0;
  }
  // ---- timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) CK(jcm::cgemm_split(xs, ws, y, np, F, B, Cin, Cout, CoutP, 0));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) CK(jcm::cgemm_split(xs, ws, y, np, F, B, Cin, Cout, CoutP, 0));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double nprod = np == 3 ? 6 : np == 5 ? 1 : 3;
  const double flop32 = 8.0 * B * Cin * (double)CoutP * F, bytes = (double)xbytes * B / (mtiles * MT) + wbytes + ybytes;
  printf("  %.3f ms  | %.1f TF fp32-equivalent, %.1f TF executed bf16 (%.1f %% of 2.5 PF) | %.2f GB algorithmic -> %.2f TB/s\n", ms, flop32 / ms / 1e9,
         flop32 * nprod / ms / 1e9, flop32 * nprod / ms / 1e9 / 25.0, bytes / 1e9, bytes / ms / 1e9);
  return bad ? 2 : 0;
}
