// Training-step kernels that are not convolutions: batch-statistics
// BatchNorm forward/backward, ReLU / max-pool / bilinear-merge backward, the soft-label spatial
// cross-entropy and its gradient, global-norm, Adam / momentum updates.  All tensors fp32 NHWC;
// per-channel reductions accumulate in double (products are formed in fp32, as the reference's fp32 graph forms them).
#include <cstdlib>

#include "kernels.h"

namespace jcm {

// ------------------------------------------------------------------------------------------------
// Per-channel column reductions over a [N, C] row-major matrix.
//   thread -> (row lane, channel), channel fastest so a wave reads contiguous memory;
//   every block writes one double partial per (quantity, channel); a second kernel folds the blocks.
// Op::Q quantities per element.
// ------------------------------------------------------------------------------------------------
constexpr int RED_THREADS = 256;
constexpr int RED_MAX_BLOCKS = 1024;

template <class T>
struct OpStats {              // sum x, sum x^2
  static constexpr int Q = 2;
  const T* x;
  __device__ void operator()(size_t i, int, double* q) const {
    const float v = static_cast<float>(x[i]);
    q[0] += (double)v;
    q[1] += (double)(v * v);
  }
};
template <class T>
struct OpBnBwd {              // sum dy, sum dy * (r - mean)      (dy optionally pre-scaled)
  static constexpr int Q = 2;
  const T* dy;
  const T* r;
  const float* mean;
  float dy_scale;
  __device__ void operator()(size_t i, int c, double* q) const {
    const float g = static_cast<float>(dy[i]) * dy_scale;
    q[0] += (double)g;
    q[1] += (double)(g * (static_cast<float>(r[i]) - mean[c]));
  }
};
template <class T>
struct OpSum {                // sum x   (bias gradient of a conv: sum of dz)
  static constexpr int Q = 1;
  const T* x;
  __device__ void operator()(size_t i, int, double* q) const { q[0] += (double)static_cast<float>(x[i]); }
};

template <class Op>
__global__ __launch_bounds__(RED_THREADS) void col_reduce_kernel(Op op, size_t N, int C, double* __restrict__ partial) {
  // partial: [gridDim.x][Q][C]
  __shared__ double red[RED_THREADS * Op::Q];
  const int lanes = RED_THREADS / C > 0 ? RED_THREADS / C : 1;   // row lanes per block when C <= 256
  const int tid = threadIdx.x;
  // channels are walked in groups of RED_THREADS when C > RED_THREADS
  for (int c0 = 0; c0 < C; c0 += RED_THREADS) {
    const int cw = C - c0 < RED_THREADS ? C - c0 : RED_THREADS;   // channels handled in this pass
    const int nl = C <= RED_THREADS ? lanes : 1;
    const int c = tid % cw, rl = tid / cw;
    double q[Op::Q];      // double accumulators: the pass is bound by its loads, and sums like d beta = sum dy cancel heavily
#pragma unroll
    for (int k = 0; k < Op::Q; ++k) q[k] = 0.0;
    if (rl < nl) {
      // four rows in flight per thread (independent partial sums, added pairwise at the end): with one row per iteration the pass ran at
      // 1.7-2.7 TB/s -- every load waited for the accumulate behind the previous one (round 5)
      double q1[Op::Q], q2[Op::Q], q3[Op::Q];
#pragma unroll
      for (int k = 0; k < Op::Q; ++k) q1[k] = q2[k] = q3[k] = 0.0;
      const size_t stride = (size_t)gridDim.x * nl;
      size_t row = (size_t)blockIdx.x * nl + rl;
      for (; row + 3 * stride < N; row += 4 * stride) {
        op(row * C + c0 + c, c0 + c, q);
        op((row + stride) * C + c0 + c, c0 + c, q1);
        op((row + 2 * stride) * C + c0 + c, c0 + c, q2);
        op((row + 3 * stride) * C + c0 + c, c0 + c, q3);
      }
      for (; row < N; row += stride) op(row * C + c0 + c, c0 + c, q);
#pragma unroll
      for (int k = 0; k < Op::Q; ++k) q[k] = (q[k] + q1[k]) + (q2[k] + q3[k]);
    }
#pragma unroll
    for (int k = 0; k < Op::Q; ++k) red[k * RED_THREADS + tid] = q[k];
    __syncthreads();
    if (tid < cw) {
#pragma unroll
      for (int k = 0; k < Op::Q; ++k) {
        double s = 0.0;
        for (int l = 0; l < nl; ++l) s += red[k * RED_THREADS + l * cw + tid];
        partial[((size_t)blockIdx.x * Op::Q + k) * C + c0 + tid] = s;
      }
    }
    __syncthreads();
  }
}

template <class Op>
static hipError_t col_reduce(const Op& op, size_t N, int C, double* partial, int* blocks_out, hipStream_t st) {
  const int nl = C <= RED_THREADS ? RED_THREADS / C : 1;
  size_t want = (N + (size_t)nl * 64 - 1) / ((size_t)nl * 64);        // ~64 rows per thread
  int blocks = (int)(want < 1 ? 1 : (want > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : want));
  hipLaunchKernelGGL(col_reduce_kernel<Op>, dim3(blocks), dim3(RED_THREADS), 0, st, op, N, C, partial);
  *blocks_out = blocks;
  return hipGetLastError();
}

size_t train_reduce_scratch_doubles(int C) { return (size_t)RED_MAX_BLOCKS * 2 * C; }

// sum over the `blocks` partials of quantity k for channel c, by the 64 threads of one wave (fixed order: deterministic)
__device__ __forceinline__ double fold_partials(const double* __restrict__ partial, int blocks, int Q, int C, int k, int c) {
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 64) s += partial[((size_t)b * Q + k) * C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}

// fold the block partials: batch mean / variance and the moving-average update of
// tf.contrib.layers.batch_norm(decay=0.9, fused): moving_var gets the Bessel-corrected variance
__global__ void bn_stats_finish_kernel(const double* __restrict__ partial, int blocks, int C, double N, float eps, float decay,
                                       float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ mov_mean,
                                       float* __restrict__ mov_var) {
  const int c = blockIdx.x;                  // one wave per channel
  const double s = fold_partials(partial, blocks, 2, C, 0, c), ss = fold_partials(partial, blocks, 2, C, 1, c);
  if (threadIdx.x != 0) return;
  const double m = s / N;
  double var = ss / N - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (mov_mean) {
    const double unb = N > 1.0 ? var * (N / (N - 1.0)) : var;
    mov_mean[c] = (float)((double)mov_mean[c] * decay + (1.0 - (double)decay) * m);
    mov_var[c] = (float)((double)mov_var[c] * decay + (1.0 - (double)decay) * unb);
  }
}

hipError_t bn_batch_stats(const void* x, bool bf16, size_t N, int C, float eps, float decay, float* mean, float* rstd, float* mov_mean,
                          float* mov_var, double* scratch, hipStream_t st) {
  int blocks = 0;
  hipError_t e = bf16 ? col_reduce(OpStats<__bf16>{static_cast<const __bf16*>(x)}, N, C, scratch, &blocks, st)
                      : col_reduce(OpStats<float>{static_cast<const float*>(x)}, N, C, scratch, &blocks, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, (double)N, eps, decay, mean,
                     rstd, mov_mean, mov_var);
  return hipGetLastError();
}

// y = (r - mean) * rstd * gamma + beta
template <class T>
__global__ void bn_apply_kernel(const T* __restrict__ r, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    y[i] = static_cast<T>((static_cast<float>(r[i]) - mean[c]) * (rstd[c] * gamma[c]) + beta[c]);
  }
}
// the same on four channels per thread (fp32 tensors, C % 4 == 0): 16-byte loads and stores, the per-channel parameters as float4 -- element for element the
// expression above (round 6: the 4-byte form moved 2.0 TB/s on conv2_fullres' 177-MB tensor)
__global__ __launch_bounds__(256) void bn_apply4_kernel(const float4* __restrict__ r, const float4* __restrict__ mean, const float4* __restrict__ rstd,
                                                        const float4* __restrict__ gamma, const float4* __restrict__ beta, float4* __restrict__ y, size_t total4, int C4) {
  // (the channel group is carried along: a 64-bit modulo per element costs more issue slots than the 16 bytes it serves)
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cstep = (int)(stride % (size_t)C4);
  int c = (int)(i0 % (size_t)C4) - cstep;
  for (size_t i = i0; i < total4; i += stride) {
    c += cstep;
    if (c >= C4) c -= C4;
    const float4 rv = r[i], m = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
    typedef float f4n __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4n{(rv.x - m.x) * (rs.x * g.x) + b.x, (rv.y - m.y) * (rs.y * g.y) + b.y, (rv.z - m.z) * (rs.z * g.z) + b.z, (rv.w - m.w) * (rs.w * g.w) + b.w},
                                reinterpret_cast<f4n*>(y + i));      // streaming: read next by another kernel, gigabytes later
  }
}
hipError_t bn_apply(const void* r, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, bool bf16, size_t N,
                    int C, hipStream_t st) {
  const size_t total = N * C;
  if (!bf16 && C % 4 == 0) {
    const size_t t4 = total / 4, g4 = (t4 + 255) / 256;
    hipLaunchKernelGGL(bn_apply4_kernel, dim3((unsigned)(g4 > 16384 ? 16384 : g4)), dim3(256), 0, st, static_cast<const float4*>(r), reinterpret_cast<const float4*>(mean),
                       reinterpret_cast<const float4*>(rstd), reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta), static_cast<float4*>(y), t4, C / 4);
    return hipGetLastError();
  }
  size_t g = (total + 255) / 256;
  const dim3 grid((int)(g > 65536 ? 65536 : g));
  if (bf16)
    hipLaunchKernelGGL(bn_apply_kernel<__bf16>, grid, dim3(256), 0, st, static_cast<const __bf16*>(r), mean, rstd, gamma, beta, static_cast<__bf16*>(y), total, C);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(r), mean, rstd, gamma, beta, static_cast<float*>(y), total, C);
  return hipGetLastError();
}

// bn_apply + the 2x2/2 SAME max pool behind it in one pass (round 6; fp32, C % 4 == 0): a thread owns a pooling window x four channels, normalises the window's
// (up to) four inputs with bn_apply4_kernel's expression, stores them (the pool's backward pass reads y) and their maximum -- the pool kernel's re-read of y is gone
__global__ __launch_bounds__(256) void bn_apply_pool4_kernel(const float4* __restrict__ r, const float4* __restrict__ mean, const float4* __restrict__ rstd,
                                                             const float4* __restrict__ gamma, const float4* __restrict__ beta, float4* __restrict__ y, float4* __restrict__ p,
                                                             int H, int W, int C4, int Ho, int Wo, size_t total) {
  typedef float f4n __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C4);
    size_t q = i / (size_t)C4;
    const int ox = (int)(q % (size_t)Wo); q /= (size_t)Wo;
    const int oy = (int)(q % (size_t)Ho);
    const size_t b = q / (size_t)Ho;
    const float4 m = mean[c], rs = rstd[c], g = gamma[c], be = beta[c];
    const int iy = oy * 2, ix = ox * 2;
    const bool hx = ix + 1 < W, hy = iy + 1 < H;
    auto one = [&](int yy, int xx) __attribute__((always_inline)) {
      const size_t o = ((b * H + yy) * W + xx) * C4 + c;
      const float4 rv = r[o];
      const float4 v = make_float4((rv.x - m.x) * (rs.x * g.x) + be.x, (rv.y - m.y) * (rs.y * g.y) + be.y, (rv.z - m.z) * (rs.z * g.z) + be.z, (rv.w - m.w) * (rs.w * g.w) + be.w);
      __builtin_nontemporal_store(f4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f4n*>(y + o));
      return v;
    };
    float4 mx = one(iy, ix);
    auto upd = [&](const float4& v) __attribute__((always_inline)) { mx = make_float4(fmaxf(mx.x, v.x), fmaxf(mx.y, v.y), fmaxf(mx.z, v.z), fmaxf(mx.w, v.w)); };
    if (hx) upd(one(iy, ix + 1));
    if (hy) upd(one(iy + 1, ix));
    if (hx && hy) upd(one(iy + 1, ix + 1));
    p[i] = mx;
  }
}
// false: not this case (bf16 / C % 4) -- the caller runs bn_apply and max_pool_2x2
bool bn_apply_pool(const void* r, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, void* p, bool bf16, int B, int H, int W, int C,
                   hipStream_t st) {
  static const bool fused = [] { const char* e = std::getenv("JCM_BN_POOL"); return !e || std::atoi(e) != 0; }();      // JCM_BN_POOL=0: the two kernels (A/B arm)
  if (!fused || bf16 || C % 4) return false;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * Ho * Wo * (C / 4), g4 = (total + 255) / 256;
  hipLaunchKernelGGL(bn_apply_pool4_kernel, dim3((unsigned)(g4 > 16384 ? 16384 : g4)), dim3(256), 0, st, static_cast<const float4*>(r), reinterpret_cast<const float4*>(mean),
                     reinterpret_cast<const float4*>(rstd), reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta), static_cast<float4*>(y),
                     static_cast<float4*>(p), H, W, C / 4, Ho, Wo, total);
  return true;
}

// sums[0][c] = sum dy, sums[1][c] = sum dy*(r-mean)  ->  dgamma = sums[1]*rstd, dbeta = sums[0]
__global__ void bn_bwd_finish_kernel(const double* __restrict__ partial, int blocks, int C, const float* __restrict__ rstd,
                                     float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;                  // one wave per channel
  const double s = fold_partials(partial, blocks, 2, C, 0, c), sx = fold_partials(partial, blocks, 2, C, 1, c);
  if (threadIdx.x != 0) return;
  sums[c] = (float)s;
  sums[C + c] = (float)sx;
  if (dgamma) dgamma[c] = (float)(sx * (double)rstd[c]);
  if (dbeta) dbeta[c] = (float)s;
}
hipError_t bn_bwd_reduce(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, size_t N, int C,
                         float* sums, float* dgamma, float* dbeta, double* scratch, hipStream_t st) {
  int blocks = 0;
  hipError_t e = bf16 ? col_reduce(OpBnBwd<__bf16>{static_cast<const __bf16*>(dy), static_cast<const __bf16*>(r), mean, dy_scale}, N, C, scratch, &blocks, st)
                      : col_reduce(OpBnBwd<float>{static_cast<const float*>(dy), static_cast<const float*>(r), mean, dy_scale}, N, C, scratch, &blocks, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, rstd, sums, dgamma, dbeta);
  return hipGetLastError();
}

// dr = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)),  xhat = (r-mean)*rstd;  dz = relu ? dr*(r>0) : dr
template <class T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, float dy_scale, const T* __restrict__ r,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums, float invN, int relu, T* __restrict__ dz, size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const float rv = static_cast<float>(r[i]);
    const float rs = rstd[c];
    const float xc = rv - mean[c];
    const float g = static_cast<float>(dy[i]) * dy_scale;
    const float d = (gamma[c] * rs) * (g - sums[c] * invN - xc * (rs * rs) * (sums[C + c] * invN));
    dz[i] = static_cast<T>((relu && !(rv > 0.f)) ? 0.f : d);
  }
}
// four channels per thread (fp32 tensors, C % 4 == 0): element for element the expression above
__device__ __forceinline__ float bn_bwd_one(float dyv, float dy_scale, float rv, float mean, float rs, float gamma, float s0, float s1, float invN, int relu) {
  const float xc = rv - mean;
  const float g = dyv * dy_scale;
  const float d = (gamma * rs) * (g - s0 * invN - xc * (rs * rs) * (s1 * invN));
  return (relu && !(rv > 0.f)) ? 0.f : d;
}
// SUM (round 6): the column sums of dz -- the bias gradient of the convolution in front -- are taken while dz is written instead of by a pass that re-reads it:
// 256 % C4 == 0 (the launcher checks), so a thread stays on ONE channel group; per-thread double sums, folded over the 256 / C4 threads of a group in a fixed
// order, one partial per (block, channel) in col_reduce_kernel's layout (Q = 1): col_sum_finish_kernel folds the blocks.
template <bool SUM>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float4* __restrict__ dy, float dy_scale, const float4* __restrict__ r, const float4* __restrict__ mean,
                                                            const float4* __restrict__ rstd, const float4* __restrict__ gamma, const float4* __restrict__ sums, float invN,
                                                            int relu, float4* __restrict__ dz, size_t total4, int C4, double* __restrict__ partial) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cstep = (int)(stride % (size_t)C4);
  int c = (int)(i0 % (size_t)C4) - cstep;
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  for (size_t i = i0; i < total4; i += stride) {
    c += cstep;
    if (c >= C4) c -= C4;
    const float4 g = dy[i], rv = r[i], m = mean[c], rs = rstd[c], ga = gamma[c], s0 = sums[c], s1 = sums[C4 + c];
    typedef float f4n __attribute__((ext_vector_type(4)));
    const f4n o = f4n{bn_bwd_one(g.x, dy_scale, rv.x, m.x, rs.x, ga.x, s0.x, s1.x, invN, relu), bn_bwd_one(g.y, dy_scale, rv.y, m.y, rs.y, ga.y, s0.y, s1.y, invN, relu),
                      bn_bwd_one(g.z, dy_scale, rv.z, m.z, rs.z, ga.z, s0.z, s1.z, invN, relu), bn_bwd_one(g.w, dy_scale, rv.w, m.w, rs.w, ga.w, s0.w, s1.w, invN, relu)};
    __builtin_nontemporal_store(o, reinterpret_cast<f4n*>(dz + i));
    if constexpr (SUM) { q0 += (double)o[0]; q1 += (double)o[1]; q2 += (double)o[2]; q3 += (double)o[3]; }
  }
  if constexpr (SUM) {
    __shared__ double red[4][256];
    const int tid = threadIdx.x;
    red[0][tid] = q0; red[1][tid] = q1; red[2][tid] = q2; red[3][tid] = q3;
    __syncthreads();
    if (tid < C4) {      // cstep == 0: thread tid + l C4 holds channel group tid
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double sacc = 0.0;
        for (int l = 0; l < 256 / C4; ++l) sacc += red[j][l * C4 + tid];
        partial[(size_t)blockIdx.x * (4 * C4) + 4 * tid + j] = sacc;
      }
    }
  }
}
hipError_t bn_bwd_apply(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma,
                        const float* sums, size_t N, int C, int relu, void* dz, hipStream_t st) {
  const size_t total = N * C;
  if (!bf16 && C % 4 == 0) {
    const size_t t4 = total / 4, g4 = (t4 + 255) / 256;
    hipLaunchKernelGGL(bn_bwd_apply4_kernel<false>, dim3((unsigned)(g4 > 16384 ? 16384 : g4)), dim3(256), 0, st, static_cast<const float4*>(dy), dy_scale, static_cast<const float4*>(r),
                       reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(rstd), reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(sums),
                       (float)(1.0 / (double)N), relu, static_cast<float4*>(dz), t4, C / 4, nullptr);
    return hipGetLastError();
  }
  size_t g = (total + 255) / 256;
  const dim3 grid((int)(g > 65536 ? 65536 : g));
  const float invN = (float)(1.0 / (double)N);
  if (bf16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, grid, dim3(256), 0, st, static_cast<const __bf16*>(dy), dy_scale, static_cast<const __bf16*>(r), mean,
                       rstd, gamma, sums, invN, relu, static_cast<__bf16*>(dz), total, C);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(dy), dy_scale, static_cast<const float*>(r), mean,
                       rstd, gamma, sums, invN, relu, static_cast<float*>(dz), total, C);
  return hipGetLastError();
}

__global__ void col_sum_finish_kernel(const double* __restrict__ partial, int blocks, int C, float* __restrict__ out) {
  const int c = blockIdx.x;                  // one wave per channel
  const double s = fold_partials(partial, blocks, 1, C, 0, c);
  if (threadIdx.x == 0) out[c] = (float)s;
}
hipError_t col_sum(const void* x, bool bf16, size_t N, int C, float* out, double* scratch, hipStream_t st);
// bn_bwd_apply + col_sum of its output in one pass over the tensors (fp32, C % 4 == 0, 256 % (C / 4) == 0; otherwise the two calls)
hipError_t bn_bwd_apply_colsum(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma, const float* sums, size_t N,
                               int C, int relu, void* dz, float* colsum, double* scratch, hipStream_t st) {
  static const bool fused = [] { const char* e = std::getenv("JCM_BN_COLSUM"); return !e || std::atoi(e) != 0; }();      // JCM_BN_COLSUM=0: the two passes (A/B arm)
  if (!fused || bf16 || C % 4 || 256 % (C / 4)) {
    if (hipError_t e = bn_bwd_apply(dy, dy_scale, r, bf16, mean, rstd, gamma, sums, N, C, relu, dz, st); e != hipSuccess) return e;
    return col_sum(dz, bf16, N, C, colsum, scratch, st);
  }
  const size_t t4 = N * C / 4, g4 = (t4 + 255) / 256;
  const int blocks = (int)(g4 > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : g4);
  hipLaunchKernelGGL(bn_bwd_apply4_kernel<true>, dim3(blocks), dim3(256), 0, st, static_cast<const float4*>(dy), dy_scale, static_cast<const float4*>(r),
                     reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(rstd), reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(sums),
                     (float)(1.0 / (double)N), relu, static_cast<float4*>(dz), t4, C / 4, scratch);
  hipLaunchKernelGGL(col_sum_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, colsum);
  return hipGetLastError();
}
hipError_t col_sum(const void* x, bool bf16, size_t N, int C, float* out, double* scratch, hipStream_t st) {
  int blocks = 0;
  hipError_t e = bf16 ? col_reduce(OpSum<__bf16>{static_cast<const __bf16*>(x)}, N, C, scratch, &blocks, st)
                      : col_reduce(OpSum<float>{static_cast<const float*>(x)}, N, C, scratch, &blocks, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(col_sum_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 2x2 stride-2 SAME max-pool backward: the gradient of a window goes to its first maximum in
// row-major window order (TF MaxPoolGrad).  x [B,H,W,C] is the pool input, dy [B,Ho,Wo,C].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  p[0] = static_cast<__bf16>(v.x); p[1] = static_cast<__bf16>(v.y); p[2] = static_cast<__bf16>(v.z); p[3] = static_cast<__bf16>(v.w);
}
// one thread = one pooling window x four channels: the window's (up to) four inputs are read once, 16 bytes each (round 4: one thread per INPUT
// element, each re-deriving its window's maximum from four scalar loads: 1.65 TB/s)
template <class T>
__global__ void max_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H, int W,
                                    int C, int Ho, int Wo, size_t total) {
  const int C4 = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho;
    const size_t b = r / Ho;
    const T* xb = x + b * (size_t)H * W * C;
    T* db = dx + b * (size_t)H * W * C;
    float4 v[4];
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
      in[k] = yy < H && xx < W;
      v[k] = in[k] ? ld4(xb + ((size_t)yy * W + xx) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 g = ld4(dy + ((b * Ho + oy) * Wo + ox) * C + c);
    // first maximum in row-major window order, per channel (k = 0 is always inside the map)
    int bx = 0, by = 0, bz = 0, bw = 0;
    float mx = v[0].x, my = v[0].y, mz = v[0].z, mw = v[0].w;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if (in[k]) {
        if (v[k].x > mx) { mx = v[k].x; bx = k; }
        if (v[k].y > my) { my = v[k].y; by = k; }
        if (v[k].z > mz) { mz = v[k].z; bz = k; }
        if (v[k].w > mw) { mw = v[k].w; bw = k; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (in[k]) {
        const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
        st4(db + ((size_t)yy * W + xx) * C + c, make_float4(bx == k ? g.x : 0.f, by == k ? g.y : 0.f, bz == k ? g.z : 0.f, bw == k ? g.w : 0.f));
      }
    }
  }
}
hipError_t max_pool_bwd(const void* x, const void* dy, void* dx, bool bf16, int B, int H, int W, int C, hipStream_t st) {
  if (C % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);      // windows x channel quads
  size_t g = (total + 255) / 256;
  const dim3 grid((int)(g > 65536 ? 65536 : g));
  if (bf16)
    hipLaunchKernelGGL(max_pool_bwd_kernel<__bf16>, grid, dim3(256), 0, st, static_cast<const __bf16*>(x), static_cast<const __bf16*>(dy),
                       static_cast<__bf16*>(dx), H, W, C, (H + 1) / 2, (W + 1) / 2, total);
  else
    hipLaunchKernelGGL(max_pool_bwd_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(x), static_cast<const float*>(dy),
                       static_cast<float*>(dx), H, W, C, (H + 1) / 2, (W + 1) / 2, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward of a layer whose output goes through the 2x2/2 max pool (conv1, conv2), given the POOLED gradient dp (round 6): the pool's backward
// pass -- dy = dp at the window's first maximum of y, 0 elsewhere -- is formed in registers by both BatchNorm backward kernels instead of being written by
// max_pool_bwd_kernel and read twice.  A thread owns a pooling window x four channels (256 % (C / 4) == 0: it stays on one channel group).
//   reduce: sum dy = sum over windows of dp; sum dy (r - mean) = sum over windows of dp (r at the maximum - mean)   (products in fp32, sums in double, as OpBnBwd)
//   apply:  dz of the window's (up to) four inputs by bn_bwd_one, and the column sums of dz (the bias gradient) as bn_bwd_apply4_kernel<true> takes them
// ------------------------------------------------------------------------------------------------
struct PoolWin { float4 y[4], r[4]; bool in[4]; int k[4]; };      // k[j]: the window slot of channel j's first maximum
__device__ __forceinline__ void pool_win_load(PoolWin& w, const float4* __restrict__ y, const float4* __restrict__ r, size_t b, int oy, int ox, int c, int H, int W, int C4) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int yy = 2 * oy + (s >> 1), xx = 2 * ox + (s & 1);
    w.in[s] = yy < H && xx < W;
    const size_t o = ((b * H + yy) * W + xx) * C4 + c;
    w.y[s] = w.in[s] ? y[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    w.r[s] = w.in[s] ? r[o] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // first maximum in row-major window order, per channel (max_pool_bwd_kernel's rule; slot 0 is always inside the map)
  float mx = w.y[0].x, my = w.y[0].y, mz = w.y[0].z, mw = w.y[0].w;
  w.k[0] = w.k[1] = w.k[2] = w.k[3] = 0;
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    if (w.in[s]) {
      if (w.y[s].x > mx) { mx = w.y[s].x; w.k[0] = s; }
      if (w.y[s].y > my) { my = w.y[s].y; w.k[1] = s; }
      if (w.y[s].z > mz) { mz = w.y[s].z; w.k[2] = s; }
      if (w.y[s].w > mw) { mw = w.y[s].w; w.k[3] = s; }
    }
  }
}
__device__ __forceinline__ float f4_get(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
// folds NQ double sums per thread over the 256 / C4 threads of a channel group (fixed order) into partial[block][NQ][C] (col_reduce_kernel's layout)
template <int NQ>
__device__ __forceinline__ void group_fold(const double (&q)[NQ][4], int C4, double* __restrict__ partial) {
  __shared__ double red[NQ * 4][256];
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NQ; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[k * 4 + j][tid] = q[k][j];
  __syncthreads();
  if (tid < C4) {
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double sacc = 0.0;
        for (int l = 0; l < 256 / C4; ++l) sacc += red[k * 4 + j][l * C4 + tid];
        partial[((size_t)blockIdx.x * NQ + k) * (4 * C4) + 4 * tid + j] = sacc;
      }
  }
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_kernel(const float4* __restrict__ dp, const float4* __restrict__ y, const float4* __restrict__ r,
                                                                   const float4* __restrict__ mean, int H, int W, int C4, int Ho, int Wo, size_t total,
                                                                   double* __restrict__ partial) {
  double q[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C4);
    size_t t = i / (size_t)C4;
    const int ox = (int)(t % (size_t)Wo); t /= (size_t)Wo;
    const int oy = (int)(t % (size_t)Ho);
    const size_t b = t / (size_t)Ho;
    PoolWin w;
    pool_win_load(w, y, r, b, oy, ox, c, H, W, C4);
    const float4 g = dp[i], m = mean[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = f4_get(g, j), mj = f4_get(m, j);
      float rk = f4_get(w.r[0], j);
#pragma unroll
      for (int s = 1; s < 4; ++s) rk = w.k[j] == s ? f4_get(w.r[s], j) : rk;
      q[0][j] += (double)gj;
      q[1][j] += (double)(gj * (rk - mj));
    }
  }
  group_fold<2>(q, C4, partial);
}
__global__ __launch_bounds__(256) void bn_bwd_apply_pooled_kernel(const float4* __restrict__ dp, const float4* __restrict__ y, const float4* __restrict__ r,
                                                                  const float4* __restrict__ mean, const float4* __restrict__ rstd, const float4* __restrict__ gamma,
                                                                  const float4* __restrict__ sums, float invN, int relu, float4* __restrict__ dz, int H, int W, int C4, int Ho,
                                                                  int Wo, size_t total, double* __restrict__ partial) {
  typedef float f4n __attribute__((ext_vector_type(4)));
  double q[1][4] = {{0.0, 0.0, 0.0, 0.0}};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C4);
    size_t t = i / (size_t)C4;
    const int ox = (int)(t % (size_t)Wo); t /= (size_t)Wo;
    const int oy = (int)(t % (size_t)Ho);
    const size_t b = t / (size_t)Ho;
    PoolWin w;
    pool_win_load(w, y, r, b, oy, ox, c, H, W, C4);
    const float4 g = dp[i], m = mean[c], rs = rstd[c], ga = gamma[c], s0 = sums[c], s1 = sums[C4 + c];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (w.in[s]) {
        const int yy = 2 * oy + (s >> 1), xx = 2 * ox + (s & 1);
        const f4n o = f4n{bn_bwd_one(w.k[0] == s ? g.x : 0.f, 1.0f, w.r[s].x, m.x, rs.x, ga.x, s0.x, s1.x, invN, relu),
                          bn_bwd_one(w.k[1] == s ? g.y : 0.f, 1.0f, w.r[s].y, m.y, rs.y, ga.y, s0.y, s1.y, invN, relu),
                          bn_bwd_one(w.k[2] == s ? g.z : 0.f, 1.0f, w.r[s].z, m.z, rs.z, ga.z, s0.z, s1.z, invN, relu),
                          bn_bwd_one(w.k[3] == s ? g.w : 0.f, 1.0f, w.r[s].w, m.w, rs.w, ga.w, s0.w, s1.w, invN, relu)};
        __builtin_nontemporal_store(o, reinterpret_cast<f4n*>(dz + ((b * H + yy) * W + xx) * C4 + c));
        q[0][0] += (double)o[0]; q[0][1] += (double)o[1]; q[0][2] += (double)o[2]; q[0][3] += (double)o[3];
      }
    }
  }
  group_fold<1>(q, C4, partial);
}
// false: not this case (bf16, C % 4, 256 % (C / 4)) -- nothing launched; the caller takes max_pool_bwd + the plain kernels
bool bn_bwd_pooled(const void* dp, const void* y, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma, int B, int H, int W, int C, float* sums,
                   float* dgamma, float* dbeta, void* dz, float* colsum, double* scratch, hipStream_t st, hipError_t* err) {
  static const bool fused = [] { const char* e = std::getenv("JCM_BN_POOLBWD"); return !e || std::atoi(e) != 0; }();      // JCM_BN_POOLBWD=0: max_pool_bwd + the plain kernels
  *err = hipSuccess;
  if (!fused || bf16 || C % 4 || 256 % (C / 4)) return false;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)B * Ho * Wo * C4, g4 = (total + 255) / 256, N = (size_t)B * H * W;
  const int blocks = (int)(g4 > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : g4);
  hipLaunchKernelGGL(bn_bwd_reduce_pooled_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float4*>(dp), static_cast<const float4*>(y), static_cast<const float4*>(r),
                     reinterpret_cast<const float4*>(mean), H, W, C4, Ho, Wo, total, scratch);
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, rstd, sums, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_pooled_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float4*>(dp), static_cast<const float4*>(y), static_cast<const float4*>(r),
                     reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(rstd), reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(sums),
                     (float)(1.0 / (double)N), 1, static_cast<float4*>(dz), H, W, C4, Ho, Wo, total, scratch);
  hipLaunchKernelGGL(col_sum_finish_kernel, dim3(C), dim3(64), 0, st, scratch, blocks, C, colsum);
  *err = hipGetLastError();
  return true;
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the TF-1.x bilinear resize [B,h,w,C] -> [B,H,W,C] (main.py:58,67), scaled by `scale`
// (the 1/3 of the branch merge): gathers, for every source pixel, the output pixels that read it.
// The index / weight arithmetic is the forward's, in fp32, so both sides agree bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(int o, float s, int n, int* lo, int* hi, float* t) {
  const float f = __fmul_rn((float)o, s);
  const int l = (int)floorf(f);
  *lo = l;
  *hi = min(l + 1, n - 1);
  *t = f - (float)l;
}
template <class T>
__global__ void resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int h, int w, int H, int W, int C, float sy,
                                  float sx, float scale, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    size_t r = i / C;
    const int ix = r % w; r /= w;
    const int iy = r % h;
    const size_t b = r / h;
    // outputs whose lo or hi index can equal iy: o*sy in [iy-1, iy+1)
    int oy0 = (int)floorf((float)(iy - 1) / sy) - 1, oy1 = (int)ceilf((float)(iy + 1) / sy) + 1;
    int ox0 = (int)floorf((float)(ix - 1) / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
    oy0 = max(oy0, 0); oy1 = min(oy1, H - 1);
    ox0 = max(ox0, 0); ox1 = min(ox1, W - 1);
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int ylo, yhi; float ty;
      bil_src(oy, sy, h, &ylo, &yhi, &ty);
      float wy = 0.f;
      if (ylo == iy) wy += 1.f - ty;
      if (yhi == iy) wy += ty;
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int xlo, xhi; float tx;
        bil_src(ox, sx, w, &xlo, &xhi, &tx);
        float wx = 0.f;
        if (xlo == ix) wx += 1.f - tx;
        if (xhi == ix) wx += tx;
        if (wx == 0.f) continue;
        acc += wy * wx * static_cast<float>(dy[((b * H + oy) * W + ox) * C + c]);
      }
    }
    dx[i] = static_cast<T>(acc * scale);
  }
}
// four channels per thread (fp32, C % 4 == 0): the tap arithmetic -- divisions, floors, the candidate loops -- is shared by the four, the sums run in the same order
__global__ __launch_bounds__(256) void resize_bwd4_kernel(const float4* __restrict__ dy, float4* __restrict__ dx, int h, int w, int H, int W, int C4, float sy, float sx,
                                                          float scale, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C4);
    size_t r = i / (size_t)C4;
    const int ix = (int)(r % (size_t)w); r /= (size_t)w;
    const int iy = (int)(r % (size_t)h);
    const size_t b = r / (size_t)h;
    int oy0 = (int)floorf((float)(iy - 1) / sy) - 1, oy1 = (int)ceilf((float)(iy + 1) / sy) + 1;
    int ox0 = (int)floorf((float)(ix - 1) / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
    oy0 = max(oy0, 0); oy1 = min(oy1, H - 1);
    ox0 = max(ox0, 0); ox1 = min(ox1, W - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy0; oy <= oy1; ++oy) {
      int ylo, yhi; float ty;
      bil_src(oy, sy, h, &ylo, &yhi, &ty);
      float wy = 0.f;
      if (ylo == iy) wy += 1.f - ty;
      if (yhi == iy) wy += ty;
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int xlo, xhi; float tx;
        bil_src(ox, sx, w, &xlo, &xhi, &tx);
        float wx = 0.f;
        if (xlo == ix) wx += 1.f - tx;
        if (xhi == ix) wx += tx;
        if (wx == 0.f) continue;
        const float4 g = dy[((b * H + oy) * W + ox) * C4 + c];
        const float wgt = wy * wx;
        acc.x += wgt * g.x; acc.y += wgt * g.y; acc.z += wgt * g.z; acc.w += wgt * g.w;
      }
    }
    dx[i] = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
  }
}
hipError_t resize_bilinear_bwd(const void* dy, void* dx, bool bf16, int B, int h, int w, int H, int W, int C, float scale, hipStream_t st) {
  if (!bf16 && C % 4 == 0) {
    const size_t t4 = (size_t)B * h * w * (C / 4), g4 = (t4 + 255) / 256;
    hipLaunchKernelGGL(resize_bwd4_kernel, dim3((unsigned)(g4 > 65536 ? 65536 : g4)), dim3(256), 0, st, static_cast<const float4*>(dy), static_cast<float4*>(dx), h, w, H, W, C / 4,
                       (float)h / (float)H, (float)w / (float)W, scale, t4);
    return hipGetLastError();
  }
  const size_t total = (size_t)B * h * w * C;
  size_t g = (total + 255) / 256;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const dim3 grid((int)(g > 65536 ? 65536 : g));
  if (bf16)
    hipLaunchKernelGGL(resize_bwd_kernel<__bf16>, grid, dim3(256), 0, st, static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), h, w, H, W, C, sy, sx, scale, total);
  else
    hipLaunchKernelGGL(resize_bwd_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(dy), static_cast<float*>(dx), h, w, H, W, C, sy, sx, scale, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// softmax_cross_entropy (main.py:220-240): one block per (image, joint) map of HW pixels.
//   loss[b*K+k] = -sum_px t * log_softmax(z);  dz (+)= gscale * (softmax(z) * sum(t) - t)
// logits [B,HW,K]; target [B,HW,Kt] (first K channels used); dz [B,HW,ldz] (channels >= K untouched).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s) sh[tid] = is_max ? fmaxf(sh[tid], sh[tid + s]) : sh[tid] + sh[tid + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, const float* __restrict__ t, int HW, int K, int Kt,
                                                         float gscale, float* __restrict__ loss, float* __restrict__ dz, int ldz,
                                                         int accumulate) {
  __shared__ float sh[256];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  const float* zb = z + (size_t)b * HW * K + k;
  const float* tb = t + (size_t)b * HW * Kt + k;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) m = fmaxf(m, zb[(size_t)i * K]);
  m = block_reduce(m, sh, true);
  float se = 0.f, st = 0.f, stz = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float zv = zb[(size_t)i * K] - m, tv = tb[(size_t)i * Kt];
    se += expf(zv);
    st += tv;
    stz += tv * zv;
  }
  se = block_reduce(se, sh, false);
  st = block_reduce(st, sh, false);
  stz = block_reduce(stz, sh, false);
  const float lse = logf(se);
  if (threadIdx.x == 0) loss[blockIdx.x] = st * lse - stz;
  if (dz) {
    float* db = dz + (size_t)b * HW * ldz + k;
    const float inv = 1.f / se;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float p = expf(zb[(size_t)i * K] - m) * inv;
      // TF-1.x's softmax_cross_entropy_with_logits kernel emits backprop = softmax - labels (the registered gradient
      // multiplies it by the upstream gradient): NOT the exact derivative p * sum(t) - t when a target map does not sum to
      // one (FLIC blobs clipped by the map border, data.py:171-183).  The reference trains with TF's, so does this.
      const float g = gscale * (p - tb[(size_t)i * Kt]);
      db[(size_t)i * ldz] = accumulate ? db[(size_t)i * ldz] + g : g;
    }
  }
}
hipError_t softmax_ce(const float* logits, const float* target, int B, int HW, int K, int Kt, float gscale, float* loss, float* dz,
                      int ldz, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(B * K), dim3(256), 0, st, logits, target, HW, K, Kt, gscale, loss, dz, ldz, accumulate);
  return hipGetLastError();
}

// out[0] (+)= scale * sum(ce[0:n]), out[1] (+)= scale * sum(ce[n:2n])   (the two loss means of eval_error, main.py:275-283)
__global__ void loss_means_kernel(const float* __restrict__ ce, int n, float scale, float* __restrict__ out, int first) {
  if (threadIdx.x >= 2 || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += ce[threadIdx.x * n + i];
  out[threadIdx.x] = (first ? 0.f : out[threadIdx.x]) + (float)(s * scale);
}
hipError_t loss_means_accumulate(const float* ce, int n, float scale, float* out, bool first, hipStream_t st) {
  hipLaunchKernelGGL(loss_means_kernel, dim3(1), dim3(64), 0, st, ce, n, scale, out, first ? 1 : 0);
  return hipGetLastError();
}

// softmax backward over the pixels of each (image, joint) map: dz += p * (g - sum_px p*g)
// p [B,HW,K] probabilities, g [B,HW,ldg] gradient w.r.t. p (first K channels), dz [B,HW,ldz].
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g, int HW, int K, int ldg,
                                                          float* __restrict__ dz, int ldz) {
  __shared__ float sh[256];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  const float* pb = p + (size_t)b * HW * K + k;
  const float* gb = g + (size_t)b * HW * ldg + k;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) s += pb[(size_t)i * K] * gb[(size_t)i * ldg];
  s = block_reduce(s, sh, false);
  float* db = dz + (size_t)b * HW * ldz + k;
  for (int i = threadIdx.x; i < HW; i += 256) db[(size_t)i * ldz] += pb[(size_t)i * K] * (gb[(size_t)i * ldg] - s);
}
hipError_t softmax_bwd(const float* p, const float* g, int B, int HW, int K, int ldg, float* dz, int ldz, hipStream_t st) {
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(B * K), dim3(256), 0, st, p, g, HW, K, ldg, dz, ldz);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// optimizer side: sum of squares (double), scale, Adam / momentum (tf.train semantics)
// ------------------------------------------------------------------------------------------------
// 16-byte loads and four double partial sums per thread where the range is 16-byte aligned (round 6: 4-byte loads moved 2.2 TB/s), scalar head / tail otherwise;
// the association is fixed by (grid, thread, lane of the float4): deterministic
__device__ __forceinline__ double sumsq_range(const float* __restrict__ x, size_t n, size_t first, size_t stride) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if ((reinterpret_cast<size_t>(x) & 15) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const size_t n4 = n >> 2;
    for (size_t i = first; i < n4; i += stride) {
      const float4 v = x4[i];
      s0 += (double)v.x * (double)v.x; s1 += (double)v.y * (double)v.y; s2 += (double)v.z * (double)v.z; s3 += (double)v.w * (double)v.w;
    }
    for (size_t i = (n4 << 2) + first; i < n; i += stride) { const double v = x[i]; s0 += v * v; }
  } else {
    for (size_t i = first; i < n; i += stride) { const double v = x[i]; s0 += v * v; }
  }
  return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partial) {
  __shared__ double sh[256];
  sh[threadIdx.x] = sumsq_range(x, n, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out, int accumulate) {
  // one wave: lane l adds partial[l], partial[l + 64], ... then a butterfly over the lanes -- a fixed association (deterministic)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) *out = (accumulate ? *out : 0.0) + s;
}
// *out (+)= sum x^2 ; scratch: 1024 doubles
hipError_t sum_squares(const float* x, size_t n, double* out, int accumulate, double* scratch, hipStream_t st) {
  size_t g = (n + 256 * 16 - 1) / (256 * 16);
  const int blocks = (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, x, n, scratch);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, scratch, blocks, out, accumulate);
  return hipGetLastError();
}

// *out = sum of x^2 over the chunks (w[k] + start[k], len[k]) with flag[k] != 0: the weight-decay term over every conv weight tensor in two
// launches (round 4: two launches per tensor, 0.7 ms per training step).  scratch: 1024 doubles
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(float* const* __restrict__ w, const int64_t* __restrict__ start, const int* __restrict__ len,
                                                           const int* __restrict__ flag, int nchunks, double* __restrict__ partial) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int k = blockIdx.x; k < nchunks; k += gridDim.x) {
    if (!flag[k]) continue;
    s += sumsq_range(w[k] + start[k], (size_t)len[k], threadIdx.x, 256);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
hipError_t sum_squares_chunks(float* const* w, const int64_t* start, const int* len, const int* flag, int nchunks, double* out, double* scratch, hipStream_t st) {
  const int blocks = nchunks < 1024 ? (nchunks < 1 ? 1 : nchunks) : 1024;
  hipLaunchKernelGGL(sumsq_chunks_kernel, dim3(blocks), dim3(256), 0, st, w, start, len, flag, nchunks, scratch);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, scratch, blocks, out, 0);
  return hipGetLastError();
}

// g *= clip / max(sqrt(*sumsq), clip)      (tf.clip_by_global_norm)
// then Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t * m / (sqrt(v) + eps)
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            const double* __restrict__ sumsq, float clip, float lr_t, float b1, float b2, float eps) {
  float s = 1.f;
  if (sumsq) {
    const float norm = (float)sqrt(*sumsq);
    s = clip / fmaxf(norm, clip);
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * s;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}
hipError_t adam_update(float* w, const float* g, float* m, float* v, size_t n, const double* sumsq, float clip, float lr_t, float b1,
                       float b2, float eps, hipStream_t st) {
  size_t gr = (n + 255) / 256;
  hipLaunchKernelGGL(adam_kernel, dim3((int)(gr > 4096 ? 4096 : gr)), dim3(256), 0, st, w, g, m, v, n, sumsq, clip, lr_t, b1, b2, eps);
  return hipGetLastError();
}
// tf.train.MomentumOptimizer: acc = mom*acc + g; w -= lr*acc
__global__ void momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ acc, size_t n,
                                const double* __restrict__ sumsq, float clip, float lr, float mom) {
  float s = 1.f;
  if (sumsq) {
    const float norm = (float)sqrt(*sumsq);
    s = clip / fmaxf(norm, clip);
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float a = mom * acc[i] + g[i] * s;
    acc[i] = a;
    w[i] = w[i] - lr * a;
  }
}
hipError_t momentum_update(float* w, const float* g, float* acc, size_t n, const double* sumsq, float clip, float lr, float mom,
                           hipStream_t st) {
  size_t gr = (n + 255) / 256;
  hipLaunchKernelGGL(momentum_kernel, dim3((int)(gr > 4096 ? 4096 : gr)), dim3(256), 0, st, w, g, acc, n, sumsq, clip, lr, mom);
  return hipGetLastError();
}

// ---- the optimizer over every trainable tensor in ONE launch: a chunk table (tensor pointer, offset of the chunk in the
// tensor, offset of the tensor in the flat gradient / slot buffers, length) replaces one launch per tensor (218 of them)
__global__ void adam_chunks_kernel(float* const* __restrict__ wptr, const int64_t* __restrict__ cstart, const int64_t* __restrict__ coff,
                                   const int* __restrict__ clen, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   const double* __restrict__ sumsq, float clip, float lr_t, float b1, float b2, float eps, int momentum_mode) {
  float s = 1.f;
  if (sumsq) {
    const float norm = (float)sqrt(*sumsq);
    s = clip / fmaxf(norm, clip);
  }
  const int c = blockIdx.x;
  float* w = wptr[c] + cstart[c];
  const int64_t base = coff[c] + cstart[c];
  const int n = clen[c];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float gi = g[base + i] * s;
    if (momentum_mode) {                       // tf.train.MomentumOptimizer: acc = mom*acc + g; w -= lr*acc   (b1 = momentum)
      const float a = b1 * m[base + i] + gi;
      m[base + i] = a;
      w[i] = w[i] - lr_t * a;
    } else {
      const float mi = b1 * m[base + i] + (1.f - b1) * gi;
      const float vi = b2 * v[base + i] + (1.f - b2) * gi * gi;
      m[base + i] = mi;
      v[base + i] = vi;
      w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
  }
}
hipError_t optimizer_chunks(float* const* wptr, const int64_t* cstart, const int64_t* coff, const int* clen, int nchunks, const float* g,
                            float* m, float* v, const double* sumsq, float clip, float lr_t, float b1, float b2, float eps, int momentum_mode,
                            hipStream_t st) {
  hipLaunchKernelGGL(adam_chunks_kernel, dim3(nchunks), dim3(256), 0, st, wptr, cstart, coff, clen, g, m, v, sumsq, clip, lr_t, b1, b2, eps,
                     momentum_mode);
  return hipGetLastError();
}

// out [N, ldo] bf16 = in [N, ldi] fp32 in the first ldi columns, zeros beyond (the logits gradient for the bf16 kernels)
__global__ void cast_pad_bf16_kernel(const float* __restrict__ in, int ldi, __bf16* __restrict__ out, int ldo, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % ldo;
    const size_t n = i / ldo;
    out[i] = static_cast<__bf16>(c < ldi ? in[n * ldi + c] : 0.f);
  }
}
hipError_t cast_pad_bf16(const float* in, int ldi, void* out, int ldo, size_t N, hipStream_t st) {
  const size_t total = N * ldo;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((int)(g > 65536 ? 65536 : g)), dim3(256), 0, st, in, ldi, static_cast<__bf16*>(out), ldo, total);
  return hipGetLastError();
}

// the same in fp32 (the logits gradient widened to the 64-channel blocks of the forward transforms)
__global__ void pad_channels_f32_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % ldo;
    const size_t n = i / ldo;
    out[i] = c < ldi ? in[n * ldi + c] : 0.f;
  }
}
// ---- overlap-save windows (kernels.h): one thread per (window pixel, 4 channels) -- float4 where C % 4 == 0, 32-bit index arithmetic
template <int VEC>
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ map, float* __restrict__ win, int H, int W, int C, int WS, int TY, int TX, int valid_only,
                                                            unsigned npix) {
  const int cv = C / VEC;      // channel groups per pixel
  const unsigned pix = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);      // one wave per window pixel: its lanes walk the channel groups
  if (pix >= npix) return;
  unsigned r = pix;
  const int wx = (int)(r % (unsigned)WS); r /= (unsigned)WS;
  const int wy = (int)(r % (unsigned)WS); r /= (unsigned)WS;
  const int tx = (int)(r % (unsigned)TX); r /= (unsigned)TX;
  const int ty = (int)(r % (unsigned)TY);
  const unsigned b = r / (unsigned)TY;
  const int V = WS - 8, y = ty * V - 4 + wy, x = tx * V - 4 + wx;
  const bool halo = wy < 4 || wy >= WS - 4 || wx < 4 || wx >= WS - 4;
  const bool in = y >= 0 && y < H && x >= 0 && x < W && !(valid_only && halo);
  const size_t src = (((size_t)b * H + (in ? y : 0)) * W + (in ? x : 0)) * C, dst = (size_t)pix * C;
  for (int g = threadIdx.x & 63; g < cv; g += 64) {
    if constexpr (VEC == 4) {
      const float4 v = in ? reinterpret_cast<const float4*>(map + src)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(win + dst)[g] = v;
    } else {
      win[dst + g] = in ? map[src + g] : 0.f;
    }
  }
}
template <int VEC>
__global__ __launch_bounds__(256) void window_scatter_kernel(const float* __restrict__ val, float* __restrict__ map, int H, int W, int C, int WS, int TY, int TX, unsigned npix) {
  const int cv = C / VEC;
  const unsigned pix = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);      // over the MAP's pixels: every one lies in exactly one valid region
  if (pix >= npix) return;
  unsigned r = pix;
  const int x = (int)(r % (unsigned)W); r /= (unsigned)W;
  const int y = (int)(r % (unsigned)H);
  const unsigned b = r / (unsigned)H;
  const int V = WS - 8, ty = y / V, tx = x / V;
  const size_t src = ((((size_t)b * TY + ty) * TX + tx) * V + (y - ty * V)) * V + (x - tx * V);
  for (int g = threadIdx.x & 63; g < cv; g += 64) {
    if constexpr (VEC == 4) reinterpret_cast<float4*>(map + (size_t)pix * C)[g] = reinterpret_cast<const float4*>(val + src * C)[g];
    else map[(size_t)pix * C + g] = val[src * C + g];
  }
}
hipError_t window_gather_f32(const float* map, float* win, int B, int H, int W, int C, int WS, int TY, int TX, int valid_only, hipStream_t st) {
  const size_t npix = (size_t)B * TY * TX * WS * WS;
  if (npix >= (1ull << 32)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((npix + 3) / 4)), blk(256);
  if (C % 4 == 0) hipLaunchKernelGGL(window_gather_kernel<4>, grid, blk, 0, st, map, win, H, W, C, WS, TY, TX, valid_only, (unsigned)npix);
  else hipLaunchKernelGGL(window_gather_kernel<1>, grid, blk, 0, st, map, win, H, W, C, WS, TY, TX, valid_only, (unsigned)npix);
  return hipGetLastError();
}
hipError_t window_scatter_f32(const float* val, float* map, int B, int H, int W, int C, int WS, int TY, int TX, hipStream_t st) {
  const size_t npix = (size_t)B * H * W;
  if (npix >= (1ull << 32)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((npix + 3) / 4)), blk(256);
  if (C % 4 == 0) hipLaunchKernelGGL(window_scatter_kernel<4>, grid, blk, 0, st, val, map, H, W, C, WS, TY, TX, (unsigned)npix);
  else hipLaunchKernelGGL(window_scatter_kernel<1>, grid, blk, 0, st, val, map, H, W, C, WS, TY, TX, (unsigned)npix);
  return hipGetLastError();
}

hipError_t pad_channels_f32(const float* in, int ldi, float* out, int ldo, size_t N, hipStream_t st) {
  const size_t total = N * ldo;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(pad_channels_f32_kernel, dim3((int)(g > 65536 ? 65536 : g)), dim3(256), 0, st, in, ldi, out, ldo, total);
  return hipGetLastError();
}

__global__ void cast_bf16_f32_kernel(const __bf16* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = static_cast<float>(in[i]);
}
hipError_t cast_bf16_f32(const void* in, float* out, size_t n, hipStream_t st) {
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)(g > 65536 ? 65536 : g)), dim3(256), 0, st, static_cast<const __bf16*>(in), out, n);
  return hipGetLastError();
}

// out[i] = a[i] * s
__global__ void scale_copy_kernel(const float* __restrict__ a, float s, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] * s;
}
hipError_t scale_copy(const float* a, float s, float* out, size_t n, hipStream_t st) {
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(scale_copy_kernel, dim3((int)(g > 65536 ? 65536 : g)), dim3(256), 0, st, a, s, out, n);
  return hipGetLastError();
}

}  // namespace jcm
