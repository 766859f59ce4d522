// FFT realisation of the spatial model's 81 pairwise convolutions (main.py:83-87).
//
// The VALID true convolution of a 120x180 prior with a 60x90 likelihood is read off a
// 120x180 *circular* convolution without any extra padding: with the likelihood in the
// top-left corner of a zero 120x180 frame,
//     circ[59+y, 89+x] = sum_{u<60,v<90} L[u,v] * A[y+59-u, x+89-v] = Cpre[y,x],  y<=60, x<=90,
// because the prior index never wraps on that window.  Per image: 10 forward R2C transforms
// (one per heat-map channel, not one per pair), 81 spectrum products against prior spectra
// computed once at jcm_finalize, 81 inverse C2R transforms.  ~76 MFLOP and ~23 MB of HBM
// traffic per image instead of 4.86 GFLOP of direct multiply-adds.
// Transforms: rocFFT through the hipFFT API (batched 2-D plans, cached per batch size);
// padding, spectrum product and the fused resize+bias+log+sum epilogue are kernels here.
#include <hipfft/hipfft.h>

#include <map>

#include "kernels.h"

namespace jcm {

constexpr int F_H = 120, F_W = 180, F_WC = F_W / 2 + 1;      // real frame, complex row length
constexpr int F_HW = F_H * F_W, F_HWC = F_H * F_WC;
constexpr int FM_H = 60, FM_W = 90, FM_HW = FM_H * FM_W;

struct SmFft {
  std::map<int, hipfftHandle> fwd, inv;   // batch -> plan (2-D R2C / C2R)
  std::map<int, hipfftHandle> col, row;   // batch -> plan (1-D C2C length 120 / 1-D C2R length 180)
  hipStream_t stream = nullptr;
};

static const char* fft_err(hipfftResult r) {
  switch (r) {
    case HIPFFT_SUCCESS: return "HIPFFT_SUCCESS";
    case HIPFFT_INVALID_PLAN: return "HIPFFT_INVALID_PLAN";
    case HIPFFT_ALLOC_FAILED: return "HIPFFT_ALLOC_FAILED";
    case HIPFFT_INVALID_VALUE: return "HIPFFT_INVALID_VALUE";
    case HIPFFT_INTERNAL_ERROR: return "HIPFFT_INTERNAL_ERROR";
    case HIPFFT_EXEC_FAILED: return "HIPFFT_EXEC_FAILED";
    case HIPFFT_SETUP_FAILED: return "HIPFFT_SETUP_FAILED";
    case HIPFFT_INVALID_SIZE: return "HIPFFT_INVALID_SIZE";
    default: return "HIPFFT error";
  }
}

SmFft* sm_fft_create(hipStream_t st) {
  SmFft* f = new SmFft();
  f->stream = st;
  return f;
}

void sm_fft_destroy(SmFft* f) {
  if (!f) return;
  for (auto& kv : f->fwd) hipfftDestroy(kv.second);
  for (auto& kv : f->inv) hipfftDestroy(kv.second);
  for (auto& kv : f->col) hipfftDestroy(kv.second);
  for (auto& kv : f->row) hipfftDestroy(kv.second);
  delete f;
}

static const char* get_plan(SmFft* f, std::map<int, hipfftHandle>& cache, hipfftType type, int batch, hipfftHandle* out) {
  auto it = cache.find(batch);
  if (it != cache.end()) { *out = it->second; return nullptr; }
  int n[2] = {F_H, F_W};
  hipfftHandle p;
  hipfftResult r = hipfftPlanMany(&p, 2, n, nullptr, 1, 0, nullptr, 1, 0, type, batch);
  if (r != HIPFFT_SUCCESS) return fft_err(r);
  r = hipfftSetStream(p, f->stream);
  if (r != HIPFFT_SUCCESS) return fft_err(r);
  cache[batch] = p;
  *out = p;
  return nullptr;
}

// real [n][120][180] -> complex [n][120][91]
const char* sm_fft_r2c(SmFft* f, const float* in, float2* out, int n) {
  hipfftHandle p;
  if (const char* e = get_plan(f, f->fwd, HIPFFT_R2C, n, &p)) return e;
  hipfftResult r = hipfftExecR2C(p, const_cast<float*>(in), reinterpret_cast<hipfftComplex*>(out));
  return r == HIPFFT_SUCCESS ? nullptr : fft_err(r);
}

// complex [n][120][91] -> real [n][120][180] (unnormalised; the input may be overwritten)
const char* sm_fft_c2r(SmFft* f, float2* in, float* out, int n) {
  hipfftHandle p;
  if (const char* e = get_plan(f, f->inv, HIPFFT_C2R, n, &p)) return e;
  hipfftResult r = hipfftExecC2R(p, reinterpret_cast<hipfftComplex*>(in), out);
  return r == HIPFFT_SUCCESS ? nullptr : fft_err(r);
}

// ---- split inverse: columns (1-D C2C, length 120) then only the 61 needed rows (1-D C2R, 180) ----
static const char* get_plan_1d(SmFft* f, std::map<int, hipfftHandle>& cache, int n, hipfftType type, int batch, hipfftHandle* out) {
  auto it = cache.find(batch);
  if (it != cache.end()) { *out = it->second; return nullptr; }
  hipfftHandle p;
  hipfftResult r = hipfftPlanMany(&p, 1, &n, nullptr, 1, 0, nullptr, 1, 0, type, batch);
  if (r != HIPFFT_SUCCESS) return fft_err(r);
  r = hipfftSetStream(p, f->stream);
  if (r != HIPFFT_SUCCESS) return fft_err(r);
  cache[batch] = p;
  *out = p;
  return nullptr;
}
// in place, [n][120] complex, unnormalised inverse
const char* sm_fft_cols(SmFft* f, float2* data, int n) {
  hipfftHandle p;
  if (const char* e = get_plan_1d(f, f->col, F_H, HIPFFT_C2C, n, &p)) return e;
  hipfftResult r = hipfftExecC2C(p, reinterpret_cast<hipfftComplex*>(data), reinterpret_cast<hipfftComplex*>(data), HIPFFT_BACKWARD);
  return r == HIPFFT_SUCCESS ? nullptr : fft_err(r);
}
// [n][91] complex -> [n][180] real, unnormalised
const char* sm_fft_rows(SmFft* f, float2* in, float* out, int n) {
  hipfftHandle p;
  if (const char* e = get_plan_1d(f, f->row, F_W, HIPFFT_C2R, n, &p)) return e;
  hipfftResult r = hipfftExecC2R(p, reinterpret_cast<hipfftComplex*>(in), out);
  return r == HIPFFT_SUCCESS ? nullptr : fft_err(r);
}

// out[n][c][r] = in[n][r][c0 + c] for c < C_out: a (column-pruning) transpose of [n][R][C] complex
// matrices through 32x32 LDS tiles, coalesced on both sides.
__global__ __launch_bounds__(256) void sm_transpose_kernel(const float2* __restrict__ in, float2* __restrict__ out, int R, int C,
                                                           int c0, int C_out, int tiles_r, int tiles_c) {
  __shared__ float2 tile[32][33];
  const int n = blockIdx.x / (tiles_r * tiles_c);
  const int t = blockIdx.x % (tiles_r * tiles_c);
  const int tr = t / tiles_c, tc = t % tiles_c;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;       // 32 x 8
  const float2* src = in + (size_t)n * R * C;
  float2* dst = out + (size_t)n * C_out * R;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = tr * 32 + ly + 8 * k, c = tc * 32 + lx;
    if (r < R && c < C_out) tile[ly + 8 * k][lx] = src[(size_t)r * C + c0 + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tc * 32 + ly + 8 * k, r = tr * 32 + lx;
    if (r < R && c < C_out) dst[(size_t)c * R + r] = tile[lx][ly + 8 * k];
  }
}
hipError_t sm_transpose(const float2* in, float2* out, int N, int R, int C, int c0, int C_out, hipStream_t st) {
  const int tiles_r = (R + 31) / 32, tiles_c = (C_out + 31) / 32;
  hipLaunchKernelGGL(sm_transpose_kernel, dim3(N * tiles_r * tiles_c), dim3(256), 0, st, in, out, R, C, c0, C_out, tiles_r, tiles_c);
  return hipGetLastError();
}

__device__ __forceinline__ float softplus5f(float x) {
  const float z = 5.0f * x;
  const float thr = 13.942385f;
  float s;
  if (z > thr) s = z;
  else if (z < -thr) s = expf(z);
  else s = log1pf(expf(z));
  return 0.2f * s;
}

// frame[b][c][120][180]: sp(bn(h[b,y,x,c])) on the top-left 60x90, zero elsewhere (sc == nullptr: raw copy).
// The C-channel map is read from two tensors -- channels [0,Ca) from hm [B,5400,Ca], the rest from extra
// [B,5400,C-Ca] -- which is tf.concat([hm, torso], axis=3) of main.py:528 without materialising it.
__global__ void sm_pad_frame_kernel(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld, const float* __restrict__ sc,
                                    const float* __restrict__ sh, float* __restrict__ frame, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = i % F_W;
    int64_t r = i / F_W;
    const int y = r % F_H; r /= F_H;
    const int c = r % C;
    const int64_t b = r / C;
    float v = 0.f;
    if (y < FM_H && x < FM_W) {
      const int64_t pix = (b * FM_H + y) * FM_W + x;
      const float hv = c < Ca ? hm[pix * Ca + c] : extra[pix * extra_ld + (c - Ca)];
      v = sc ? softplus5f(hv * sc[c] + sh[c]) : hv;
    }
    frame[i] = v;
  }
}
hipError_t sm_pad_frame(const float* hm, int Ca, const float* extra, const float* sc, const float* sh, float* frame, int B, int C,
                        hipStream_t st, int extra_ld) {
  if (extra_ld <= 0) extra_ld = C - Ca;
  const int64_t total = (int64_t)B * C * F_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_pad_frame_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, hm, Ca, extra, extra_ld, sc, sh, frame, C, total);
  return hipGetLastError();
}

// spec[b][p] = lhat[b][cond[p]] * phat[p] * scale       (scale = 1/(120*180): hipFFT is unnormalised)
// two complex values (16 bytes) per thread; F_HWC = 120*91 is even
__global__ void sm_spec_mul_kernel(const float4* __restrict__ lhat, const float4* __restrict__ phat, const int* __restrict__ cond,
                                   float4* __restrict__ spec, int C, int P, float scale, int64_t total) {
  constexpr int HALF = F_HWC / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = i % HALF;
    int64_t r = i / HALF;
    const int p = r % P;
    const int64_t b = r / P;
    const float4 l = lhat[(b * C + cond[p]) * HALF + k];
    const float4 q = phat[(int64_t)p * HALF + k];
    float4 o;
    o.x = (l.x * q.x - l.y * q.y) * scale;
    o.y = (l.x * q.y + l.y * q.x) * scale;
    o.z = (l.z * q.z - l.w * q.w) * scale;
    o.w = (l.z * q.w + l.w * q.z) * scale;
    spec[i] = o;
  }
}
hipError_t sm_spec_mul(const float2* lhat, const float2* phat, const int* cond, float2* spec, int B, int C, int P, hipStream_t st) {
  static_assert(F_HWC % 2 == 0, "two complex values per thread");
  const int64_t total = (int64_t)B * P * (F_HWC / 2);
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_spec_mul_kernel, dim3((int)(g > 32768 ? 32768 : g)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(lhat), reinterpret_cast<const float4*>(phat), cond,
                     reinterpret_cast<float4*>(spec), C, P, 1.0f / (float)F_HW, total);
  return hipGetLastError();
}

// TF-1.x bilinear 61x91 -> 60x90 (main.py:89) sampled from the VALID window of a circular
// convolution frame: Cpre[y][x] = frame[59+y][89+x].
__device__ __forceinline__ float resize_from_frame(const float* __restrict__ fr, int oy, int ox) {
  const float sy = 61.0f / 60.0f, sx = 91.0f / 90.0f;
  const float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);
  const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  const int yhi = min(ylo + 1, 60), xhi = min(xlo + 1, 90);
  const float ty = fy - (float)ylo, tx = fx - (float)xlo;
  const float* w = fr + 59 * F_W + 89;
  const float tl = w[ylo * F_W + xlo], tr = w[ylo * F_W + xhi];
  const float bl = w[yhi * F_W + xlo], br = w[yhi * F_W + xhi];
  const float top = tl + (tr - tl) * tx;
  const float bot = bl + (br - bl) * tx;
  return top + (bot - top) * ty;
}

// E[b,pix,j] = log(frame_lik[b][j][pix] + d) + sum over the C-1 pairs of j, graph order, of
//              log(R(cfull[b][p]) + spb[p][pix] + d)                      (main.py:117-123)
__global__ void sm_finish_fft_kernel(const float* __restrict__ frame, const float* __restrict__ cfull, const float* __restrict__ spb,
                                     float* __restrict__ logits, int K, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % FM_HW;
    int64_t r = i / FM_HW;
    const int j = r % K;
    const int64_t b = r / K;
    const int oy = pix / FM_W, ox = pix - oy * FM_W;
    float e = logf(frame[((b * C + j) * F_H + oy) * F_W + ox] + 1e-6f);
    const int PJ = C - 1, P = K * PJ;
    for (int q = 0; q < PJ; ++q) {
      const int p = j * PJ + q;
      const float cv = resize_from_frame(cfull + (b * P + p) * F_HW, oy, ox);
      e += logf((cv + spb[(size_t)p * FM_HW + pix]) + 1e-6f);
    }
    logits[(b * FM_HW + pix) * K + j] = e;
  }
}
hipError_t sm_finish_fft(const float* frame, const float* cfull, const float* spbias, float* logits, int B, int K, int C, hipStream_t st) {
  const int64_t total = (int64_t)B * K * FM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_finish_fft_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, frame, cfull, spbias, logits, K, C, total);
  return hipGetLastError();
}

// the same epilogue on the split path: rows[b][p] is [61][180] = frame rows 59..119, window = columns 89..179
__device__ __forceinline__ float resize_from_rows(const float* __restrict__ rw, int oy, int ox) {
  const float sy = 61.0f / 60.0f, sx = 91.0f / 90.0f;
  const float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);
  const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  const int yhi = min(ylo + 1, 60), xhi = min(xlo + 1, 90);
  const float ty = fy - (float)ylo, tx = fx - (float)xlo;
  const float* w = rw + 89;
  const float tl = w[ylo * F_W + xlo], tr = w[ylo * F_W + xhi];
  const float bl = w[yhi * F_W + xlo], br = w[yhi * F_W + xhi];
  const float top = tl + (tr - tl) * tx;
  const float bot = bl + (br - bl) * tx;
  return top + (bot - top) * ty;
}
__global__ void sm_finish_rows_kernel(const float* __restrict__ frame, const float* __restrict__ rows, const float* __restrict__ spb,
                                      float* __restrict__ logits, float* __restrict__ tsave, int K, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % FM_HW;
    int64_t r = i / FM_HW;
    const int j = r % K;
    const int64_t b = r / K;
    const int oy = pix / FM_W, ox = pix - oy * FM_W;
    float e = logf(frame[((b * C + j) * F_H + oy) * F_W + ox] + 1e-6f);
    const int PJ = C - 1, P = K * PJ;
    for (int q = 0; q < PJ; ++q) {
      const int p = j * PJ + q;
      const float cv = resize_from_rows(rows + (b * P + p) * (61 * F_W), oy, ox);
      const float tv = (cv + spb[(size_t)p * FM_HW + pix]) + 1e-6f;
      if (tsave) tsave[(b * P + p) * FM_HW + pix] = tv;      // training: the log's argument is the backward's denominator
      e += logf(tv);
    }
    logits[(b * FM_HW + pix) * K + j] = e;
  }
}
hipError_t sm_finish_rows(const float* frame, const float* rows, const float* spbias, float* logits, float* tsave, int B, int K, int C,
                          hipStream_t st) {
  const int64_t total = (int64_t)B * K * FM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_finish_rows_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, st, frame, rows, spbias, logits, tsave, K, C, total);
  return hipGetLastError();
}

__global__ void sm_resize_frame_kernel(const float* __restrict__ cfull, float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = i % FM_HW;
    const int64_t b = i / FM_HW;
    out[i] = resize_from_frame(cfull + b * F_HW, pix / FM_W, pix % FM_W);
  }
}
hipError_t sm_resize_frame(const float* cfull, float* out, int B, hipStream_t st) {
  const int64_t total = (int64_t)B * FM_HW;
  int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(sm_resize_frame_kernel, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, st, cfull, out, total);
  return hipGetLastError();
}

}  // namespace jcm
