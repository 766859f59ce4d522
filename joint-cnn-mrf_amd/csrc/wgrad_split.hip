// Weight gradient on the bf16 matrix cores with fp32-class accuracy (the bf16x6 operand split of
// conv_split.hip applied to wgrad.hip's dataflow).
//
//   dW[ky][kx][ci][co] = sum_{b,y,x} X[b, y+ky-PAD, x+kx-PAD, ci] * dZ[b, y, x, co]
//
// GEMM per tap: M = ci, N = co, K = pixels.  v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive
// K values of one M (or N) index -- but NHWC keeps the *channels* of a pixel contiguous, not the
// pixels of a channel.  gfx950's LDS transpose read closes the gap: the strip is staged exactly as it
// lies in memory, [pixel][32 channels] (64-byte rows; LDS-DMA, no VGPR round trip), and
// ds_read_b64_tr_b16 hands every lane the 4 consecutive pixels of its channel: in each 16-lane group,
// lane t supplies the address of the 8-byte piece (row t/4, columns 4(t%4)..+3) of a 4 x 16 block
// and receives column t of it.  The kx tap shift is a row offset of the read (an immediate).
//
// Both operands arrive pre-split into three bf16 parts (split_parts: a = a0 + a1 + a2 exactly); the
// six products with part indices p + q <= 2 accumulate into the tap's fp32 accumulator.  A workgroup
// owns one tap row (ky, all kx) of a 64(ci) x 64(co) tile like wgrad_kernel; K is split over
// workgroups and the partial tiles are reduced by wgrad_reduce.
#include "kernels.h"

namespace jcm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WS_PW = 32;          // output pixels per strip = two k16 steps
constexpr int WS_XPX = 48;         // staged input pixels per strip (>= 32 + KS - 1, a multiple of the 16-pixel DMA piece)
constexpr int WS_ROW = 64;         // bytes per staged row: 32 bf16 channels
// NP = operand parts: 3 = fp32 operands split by split_parts (six products), 1 = plain bf16 operands (one product)
__host__ __device__ constexpr int ws_xbytes(int np) { return np * 2 * WS_XPX * WS_ROW; }     // [part][half][pixel][32 ch]
__host__ __device__ constexpr int ws_zbytes(int np) { return np * 2 * WS_PW * WS_ROW; }
__host__ __device__ constexpr int ws_buf(int np) { return ws_xbytes(np) + ws_zbytes(np); }

template <int OFF>
__device__ __forceinline__ u32x2 tr_read_imm(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ bf16x8 pack_op(u32x2 lo, u32x2 hi) {
  u32x4 r = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, r);
}

// The KS shifted activation operands of one (k16 step S, part PA).  The operand of tap kx is rows kx..kx+3 and kx+4..kx+7
// (plus 8*kh) of the strip, so the 4-row blocks at offsets 0 .. KS+3 serve all taps: KS + 4 transpose reads instead of 2*KS.
template <int KS, int S, int PA, int O>
__device__ __forceinline__ void load_blocks(unsigned abase, u32x2 (&blk)[KS + 4]) {
  if constexpr (O < KS + 4) {
    constexpr int off = (PA * 2 * WS_XPX + 16 * S + O) * WS_ROW;       // + 8*kh rows in the lane address
    blk[O] = tr_read_imm<off>(abase);
    load_blocks<KS, S, PA, O + 1>(abase, blk);
  }
}
// all MFMAs of one (S, PA): KS taps x the gradient parts q <= NP - 1 - PA
template <int KS, int NP, bool F16, int S, int PA>
__device__ __forceinline__ void taps(unsigned abase, const bf16x8 (&bz)[NP], f32x16 (&acc)[KS]) {
  u32x2 blk[KS + 4];
  load_blocks<KS, S, PA, 0>(abase, blk);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);       // the MFMAs below must not be hoisted above the wait (the compiler cannot see the reads)
#pragma unroll
  for (int kx = 0; kx < KS; ++kx) {
    const bf16x8 a = pack_op(blk[kx], blk[kx + 4]);
#pragma unroll
    for (int q = 0; q <= NP - 1 - PA; ++q) {
      if constexpr (F16)
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bz[q]), acc[kx], 0, 0, 0);
      else
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bz[q], acc[kx], 0, 0, 0);
    }
  }
}

template <int NP, int S, int Q>
__device__ __forceinline__ void load_z(unsigned zbase, u32x2 (&lo)[NP], u32x2 (&hi)[NP]) {
  if constexpr (Q < NP) {
    constexpr int off = (Q * 2 * WS_PW + 16 * S) * WS_ROW;
    lo[Q] = tr_read_imm<off>(zbase);
    hi[Q] = tr_read_imm<off + 4 * WS_ROW>(zbase);
    load_z<NP, S, Q + 1>(zbase, lo, hi);
  }
}
template <int KS, int NP, bool F16, int S, int PA>
__device__ __forceinline__ void parts_desc(unsigned abase, const bf16x8 (&bz)[NP], f32x16 (&acc)[KS]) {   // smallest products first
  if constexpr (PA >= 0) {
    taps<KS, NP, F16, S, PA>(abase, bz, acc);
    parts_desc<KS, NP, F16, S, PA - 1>(abase, bz, acc);
  }
}
template <int KS, int NP, bool F16, int S>
__device__ __forceinline__ void k16_step(unsigned abase, unsigned zbase, f32x16 (&acc)[KS]) {
  bf16x8 bz[NP];
  {
    u32x2 lo[NP], hi[NP];
    load_z<NP, S, 0>(zbase, lo, hi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NP; ++q) bz[q] = pack_op(lo[q], hi[q]);
  }
  parts_desc<KS, NP, F16, S, NP - 1>(abase, bz, acc);
}

template <int KS, int NP, int NBUF, bool F16>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const __bf16* __restrict__ xp, const __bf16* __restrict__ zp,
                                                              float* __restrict__ partial, int B, int H, int W, int Cin, int Cout, int ldz,
                                                              int n_ci, int n_co, int splits, long xpart, long zpart) {
  constexpr int PAD = (KS - 1) / 2;
  static_assert(WS_PW + KS - 1 <= WS_XPX, "halo fits the staged rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];     // NBUF strip buffers: NBUF - 1 strips of LDS-DMA in flight

  int bid = blockIdx.x;
  const int ky = bid % KS; bid /= KS;
  const int cit = bid % n_ci; bid /= n_ci;
  const int cot = bid % n_co; bid /= n_co;
  const int split = bid;
  const int ci0 = cit * 64, co0 = cot * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wo = wid & 1;

  const int nseg = (W + WS_PW - 1) / WS_PW;
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

  // ---- staging: 10 * NP LDS-DMA pieces of 1 KB per strip (16 pixels x 64 B), dealt round-robin to the 4 waves.
  // A row of the image is one buffer descriptor, so pixels left / right of it read as zeros (SAME padding).
  const int dpix = lane >> 2, dchunk = lane & 3;
  auto stage = [&](long s, int bufsel) {
    const long row = r0 + s / nseg;
    const int seg = (int)(s % nseg);
    const int b = (int)(row / nvalid), y = ylo + (int)(row % nvalid);
    const int yi = y + ky - PAD;
    const int px0 = seg * WS_PW;
    char* base = smem + bufsel * ws_buf(NP);
    constexpr int PER_WAVE = (10 * NP + 3) / 4;         // every wave issues the same number of DMAs per strip (the wait counts
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {                //  below are immediates): surplus slots repeat a piece, which is harmless
      const int d = (wid + 4 * i) % (10 * NP);          // wave-uniform piece index
      if (d < 6 * NP) {                                     // X: [part][half][3 groups of 16 pixels]
        const int grp = d % 3, half = (d / 3) % 2, part = d / 6;
        const __bf16* rowp = xp + part * xpart + ((size_t)b * H + yi) * W * Cin;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(rowp), 0, W * Cin * 2, 0x00020000);
        const int pix = px0 - PAD + 16 * grp + dpix;
        const unsigned voff = (unsigned)((pix * Cin + ci0 + 32 * half) * 2 + dchunk * 16);
        char* dst = base + ((part * 2 + half) * WS_XPX + 16 * grp) * WS_ROW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
      } else {                                          // dZ: [part][half][2 groups of 16 pixels]
        const int e = d - 6 * NP;
        const int grp = e % 2, half = (e / 2) % 2, part = e / 4;
        const __bf16* rowp = zp + part * zpart + ((size_t)b * H + y) * W * ldz;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(rowp), 0, W * ldz * 2, 0x00020000);
        const int pix = px0 + 16 * grp + dpix;
        const unsigned voff = (unsigned)((pix * ldz + co0 + 32 * half) * 2 + dchunk * 16);
        char* dst = base + ws_xbytes(NP) + ((part * 2 + half) * WS_PW + 16 * grp) * WS_ROW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
      }
    }
  };

  // ---- per-lane read addresses: group g = lane>>4 -> (column block g&1, k half g>>1); lane t of the group supplies the
  // piece (row t/4, columns 4(t%4)..+3) of the 4 x 16 block
  const int g = lane >> 4, t = lane & 15;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_off = (unsigned)((8 * (g >> 1) + (t >> 2)) * WS_ROW + (16 * (g & 1) + 4 * (t & 3)) * 2);
  const unsigned a_lane = lds0 + (unsigned)(wi * WS_XPX * WS_ROW) + lane_off;
  const unsigned z_lane = lds0 + (unsigned)(ws_xbytes(NP) + wo * WS_PW * WS_ROW) + lane_off;

  constexpr int D = NBUF - 1;                               // prefetch distance in strips
  constexpr int PER_WAVE_DMA = (10 * NP + 3) / 4;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nstrip) stage(d, d);
  int buf = 0;
  for (long s = 0; s < nstrip; ++s) {
    // this wave's pieces of strip s have landed: at most the D-1 younger strips may still be in flight (near the end fewer
    // strips are outstanding than that, so drain completely there)
    if (s + D <= nstrip) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PER_WAVE_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // ... and everyone's; every wave is past its reads of strip s-1's buffer
    if (s + D < nstrip) stage(s + D, (buf + D) % NBUF);     // = the buffer strip s-1 used
    const unsigned ab = a_lane + (unsigned)(buf * ws_buf(NP)), zb = z_lane + (unsigned)(buf * ws_buf(NP));
    k16_step<KS, NP, F16, 0>(ab, zb, acc);
    k16_step<KS, NP, F16, 1>(ab, zb, acc);
    buf = (buf + 1) % NBUF;
  }

  // ---- partial tile store: D col = lane&31 -> co, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> ci
  float* out = partial + (size_t)split * KS * KS * Cin * Cout;
  const int h = lane >> 5, l31 = lane & 31;
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

// fp16 parts of x * S: out[0] = fp16(xS), out[1] = fp16(xS - out[0])   (conv_split.hip, fp16x3)
__global__ void split_parts16_kernel(const float* __restrict__ x, _Float16* __restrict__ out, size_t n, const float* __restrict__ scale) {
  const float S = scale ? scale[0] : 1.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i] * S;
    const _Float16 h0 = static_cast<_Float16>(v);
    out[i] = h0;
    out[n + i] = static_cast<_Float16>(v - static_cast<float>(h0));
  }
}

// max |x| per block -> scratch; then S = 2^floor(log2(2^14 / max)) (clamped), scale = {S, 1/S}
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ scratch) {
  __shared__ float sh[256];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + k]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scratch[blockIdx.x] = sh[0];
}
__global__ void pow2_scale_kernel(const float* __restrict__ scratch, int nb, float* __restrict__ scale) {   // one wave
  float m = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) m = fmaxf(m, scratch[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (threadIdx.x != 0) return;
  int e = 0;                                   // m = f * 2^e, f in [0.5, 1)
  if (m > 0.f && isfinite(m)) (void)frexpf(m, &e);
  else e = 14;                                 // all zeros (or not finite): S = 1
  int k = 14 - e;                              // m * 2^k in [2^13, 2^14)
  k = k > 60 ? 60 : (k < -60 ? -60 : k);
  scale[0] = ldexpf(1.0f, k);
  scale[1] = ldexpf(1.0f, -k);
}

}  // namespace

hipError_t split_parts16(const float* x, void* out, size_t n, const float* scale, hipStream_t st) {
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(split_parts16_kernel, dim3((int)(g > 65536 ? 65536 : g)), dim3(256), 0, st, x, static_cast<_Float16*>(out), n, scale);
  return hipGetLastError();
}

hipError_t pow2_scale_of(const float* x, size_t n, float* scale, float* scratch, hipStream_t st) {
  size_t g = (n + 256 * 16 - 1) / (256 * 16);
  const int nb = (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
  hipLaunchKernelGGL(amax_kernel, dim3(nb), dim3(256), 0, st, x, n, scratch);
  hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(64), 0, st, scratch, nb, scale);
  return hipGetLastError();
}

bool wgrad_split_supported(int ks, int Cin, int ldz) { return (ks == 9 || ks == 5) && Cin % 8 == 0 && ldz % 8 == 0; }

namespace {
template <int KS, int NP, bool F16 = false>
hipError_t launch_w(const void* xp, const void* zp, float* partial, int splits, int B, int H, int W, int Cin, int Cout, int ldz, hipStream_t st) {
  const int n_ci = (Cin + 63) / 64, n_co = (Cout + 63) / 64;
  const int blocks = KS * n_ci * n_co * splits;
  const long xpart = (long)B * H * W * Cin, zpart = (long)B * H * W * ldz;
  constexpr int NBUF = NP == 1 ? 4 : 2;      // a bf16 strip is 18 MFMAs per wave: three strips of DMA in flight cover the HBM/L2 latency
  hipLaunchKernelGGL((wgrad_split_kernel<KS, NP, NBUF, F16>), dim3(blocks), dim3(256), NBUF * ws_buf(NP), st, static_cast<const __bf16*>(xp),
                     static_cast<const __bf16*>(zp), partial, B, H, W, Cin, Cout, ldz, n_ci, n_co, splits, xpart, zpart);
  return hipGetLastError();
}
}  // namespace

// fp16x3: xp / zp are split_parts16 images (two fp16 parts each, dz pre-scaled by its power-of-two scale)
hipError_t wgrad_split16(const void* xp, const void* zp, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                         hipStream_t st) {
  if (!wgrad_split_supported(ks, Cin, ldz)) return hipErrorInvalidValue;
  return ks == 9 ? launch_w<9, 2, true>(xp, zp, partial, splits, B, H, W, Cin, Cout, ldz, st)
                 : launch_w<5, 2, true>(xp, zp, partial, splits, B, H, W, Cin, Cout, ldz, st);
}

// x, dz bf16 NHWC themselves (bf16 training): one product per k16 step, fp32 accumulate
hipError_t wgrad_bf16(const void* x, const void* dz, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                      hipStream_t st) {
  if (!wgrad_split_supported(ks, Cin, ldz)) return hipErrorInvalidValue;
  return ks == 9 ? launch_w<9, 1>(x, dz, partial, splits, B, H, W, Cin, Cout, ldz, st) : launch_w<5, 1>(x, dz, partial, splits, B, H, W, Cin, Cout, ldz, st);
}

}  // namespace jcm
