// Stride-1 SAME convolution (9x9, 5x5) in the FREQUENCY domain: every such layer of an fp32 handle, the wide 9x9 layers (conv4_*, conv5:
// 256/512 -> 512 channels) of a bf16 handle.
//
// A 9x9 layer with 512 x 512 channels is 229 GFLOP per image as a direct convolution.  With the maps transformed once the layer is, for
// every frequency, one complex matrix product over the channels:  Y[f][b][co] = sum_ci X[f][b][ci] * Wf[f][ci][co].  The transform is a
// CIRCULAR convolution of size NY x NX >= (H + pad) x (W + pad), pad = (k-1)/2: the SAME output y in [0, H) reads inputs y-pad .. y+pad,
// so the wrap-around only has to land in the zero rows H .. NY-1 -- which H + pad rows guarantee (a full linear convolution would need
// H + k - 1).  60x90 maps: 64 x 96 transforms, 64 * 49 = 3136 frequencies, 8*Cin*Cout*3136 = 6.6 GFLOP per image for conv5, 35x fewer
// than the direct form.  In fp32 the route is MORE accurate than the fp32 MFMA accumulation chain it replaces.
//
//   rows_fwd        (image, row, 64 channels)        : NHWC fp32 / bf16 (or planar bf16); two adjacent channels = one complex number
//                                                      z = x_c + i x_{c+1}; complex FFT along x in LDS; X_c, X_{c+1} through the Hermitian
//                                                      symmetry -> T[kx][c/16][b][y][16]   (a column work group's input is contiguous)
//   cols_fwd_split  (8 images, kx, 16 channels)      : FFT along y, then the spectra are SPLIT into bf16 parts and written in the exact
//                                                      LDS image of the channel GEMM: Xs[f][m-tile][c/16][re|im][part][k-half][row][8]
//   cgemm_split     (cgemm_split.hip)                : the channel GEMM on the bf16 matrix cores, one complex product per frequency
//   cols_inv        (image, kx, 64 channels)         : inverse along ky, rows pad .. pad+H-1 kept -> T[b][y][kx][co]
//   rows_inv        (image, row, 64 channels)        : Z = Y_c + i Y_{c+1} (Hermitian extension), inverse complex FFT along kx, columns
//                                                      pad .. pad+W-1, 1/(NY NX), bias, ReLU, folded BatchNorm -> NHWC fp32 / bf16 / planar
// The filter spectra (flipped kernel: TF's conv2d is a correlation) are computed once per (layer, map size) at first use, split into the
// same bf16 parts, in the GEMM's tile-major layout.  FFTs: the in-LDS decimation-in-frequency stages of sm_fused.hip, channel-vectorised
// (consecutive lanes = consecutive channels).  Twiddles come from one table per device, built on the host in double precision.
// Reference semantics: conv2d SAME stride 1 + bias + ReLU + BatchNorm (main.py:133-135,156-169).
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "conv_fft_common.h"

namespace jcm {
namespace cfft {

// ---- filter spectra, split: HWIO fp32 [k][k][Cin][Cout] -> Wf[f][ci][co] = sum_{a,b} w[k-1-a][k-1-b][ci][co] e^{-2 pi i (ky a / NY + kx b / NX)}
// (the flipped kernel: TF's conv2d is a correlation; output channels Cout .. CoutP-1 are zero), written as the channel GEMM's operand
// image Ws[f][co/ntl][ci/16][re|im][part][k-half][ntl columns][8 x 16 bit] (cgemm_split.hip).
// A thread owns the 8 consecutive input channels of one 16-byte unit of one output channel and one kx:
// the k row sums of its filters stay in registers, then for one ky after the other the k-term column sum gives their spectra,
// which are split and stored straight from registers -- consecutive threads are consecutive output channels, so a wave's store instruction is
// one contiguous 1-KB run.  No exchange through LDS, no barrier in the loop (round 3: the packer runs once per weight update in the training
// step, where it is 20 % of the step).
// NP = 4: two FP16 parts of the spectrum times 2^k, k from wscale[0] = max over filters of sum |taps| >= |W[f]| (weight_bound_kernel); layout as
// NP = 2; wscale[1] = 2^-k for the inverse row pass.
template <int KS, int NP>
__global__ __launch_bounds__(256) void weight_spectra_split_kernel(const float* __restrict__ w, uint4* __restrict__ Ws, int Cin, int Cout, int CoutP, int ntl, int NY, int NX,
                                                                   int round_bf16, float* __restrict__ wscale) {
  __shared__ cf twy[192][KS - 1], twx[KS];      // e^{-2 pi i ky a / NY} (a = 1..KS-1), e^{-2 pi i kx b / NX} of this block's kx
  const int tid = threadIdx.x;
  const int kx = blockIdx.y;
#pragma unroll 1
  for (int k = tid; k < NY * (KS - 1) + KS; k += 256) {
    const bool isy = k < NY * (KS - 1);
    const int ky = k / (KS - 1), a = isy ? k % (KS - 1) + 1 : k - NY * (KS - 1);
    const int num = isy ? (ky * a) % NY : (kx * a) % NX, den = isy ? NY : NX;
    double sn, cs;
    sincospi(-2.0 * (double)num / (double)den, &sn, &cs);
    cf* dstw = isy ? &twy[ky][a - 1] : &twx[a];
    *dstw = cf{(float)cs, (float)sn};
  }
  __syncthreads();
  constexpr int CPT = 8;                                      // input channels per thread = one 16-byte unit of the layout
  constexpr int NPP = NP == 5 ? 1 : 2;                        // 16-byte units this thread writes per plane (NP = 2: bf16 parts, 4: fp16 parts, 5: one fp16 part)
  static_assert(NP == 2 || NP == 4 || NP == 5, "operand forms of the channel GEMM");
  float wmul = 1.f;
  if constexpr (NP >= 4) {
    int ex = 0;
    const float bound = wscale[0];
    if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &ex);      // bound < 2^ex
    wmul = ldexpf(1.f, 14 - ex);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) wscale[1] = ldexpf(1.f, ex - 14);
  }
  const size_t e = (size_t)blockIdx.x * 256 + tid;
  if (e >= (size_t)(Cin / CPT) * CoutP) return;
  const int co = (int)(e % CoutP), cig = (int)(e / CoutP);
  cf ra[CPT][KS];                                             // row sums of the flipped kernels of input channels CPT cig .. +CPT-1
  const unsigned toff = (unsigned)(cig * CPT * Cout + (co < Cout ? co : Cout - 1));
#pragma unroll
  for (int c = 0; c < CPT; ++c)
#pragma unroll
    for (int a = 0; a < KS; ++a) ra[c][a] = cf{0.f, 0.f};
#pragma unroll 1
  for (int a = 0; a < KS; ++a) {               // a real loop (k x CPT loads in flight, not k x k x CPT and their addresses) ...
    cf sum[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) sum[c] = cf{0.f, 0.f};
#pragma unroll
    for (int b = 0; b < KS; ++b) {
      // uniform tap base + 32-bit lane offset (always in bounds: no branch per load); tap-major so that one address serves the CPT channels
      const float* tap = w + (size_t)((KS - 1 - a) * KS + (KS - 1 - b)) * Cin * Cout;
      const cf tw = twx[b];
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        float wv = tap[toff + (unsigned)(c * Cout)];
        wv = co < Cout ? wv : 0.f;
        if (round_bf16) wv = static_cast<float>(static_cast<__bf16>(wv));       // bf16 handles: the filter the bf16 MFMA kernels multiply with
        sum[c] = sfma(wv, tw, sum[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
      for (int aa = 0; aa < KS; ++aa) ra[c][aa] = aa == a ? sum[c] : ra[c][aa];      // ... with the register array written by selects, not by index
  }
  const int KC = Cin / 16, ntiles = CoutP / ntl;
  const int ci0 = cig * CPT, kc = ci0 >> 4, kg = (ci0 >> 3) & 1, nt = co / ntl, sn = co % ntl;
  // 16-byte units: stride between frequencies, and this thread's units inside one frequency
  // NP = 5: stages of 32 channels, the two planes of the NP = 2 layout = the stage's two 16-channel halves: unit ((c * 2 + half) * 2 + kg) * ntl + column
  const size_t fstride = NP == 5 ? (size_t)ntiles * (KC >> 1) * 8 * ntl : (size_t)ntiles * KC * (4 * NPP) * ntl;
  uint4* dst = Ws + (size_t)kx * NY * fstride +
               (NP == 5 ? ((size_t)nt * (KC >> 1) + (kc >> 1)) * 8 * ntl + ((size_t)(kc & 1) * 2 + kg) * ntl + sn :
                          ((size_t)nt * KC + kc) * (4 * NPP) * ntl + (size_t)kg * ntl + sn);
  // the spectrum of one ky -> this thread's units (split into the operand parts of the layout); streaming stores: nothing on this GPU reads them back soon
  typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
  auto put = [](uint4* q, const uint4& v) __attribute__((always_inline)) { __builtin_nontemporal_store(u32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4n*>(q)); };
  auto emit = [&](uint4* d, const float (&xr)[CPT], const float (&xi)[CPT]) __attribute__((always_inline)) {
    if constexpr (NP == 5) {
      float x8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x8[c] = xr[c % CPT];
      put(d, round8h(x8, wmul));
#pragma unroll
      for (int c = 0; c < 8; ++c) x8[c] = xi[c % CPT];
      put(d + (size_t)4 * ntl, round8h(x8, wmul));
    } else {
      // [re|im][part][k-half][ntl][8 bf16]
      uint4 u[NPP];
      float x8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x8[c] = xr[c % CPT];
      if constexpr (NP == 4) split8h(x8, wmul, u);
      else split8<NPP>(x8, u);
#pragma unroll
      for (int p = 0; p < NPP; ++p) put(d + (size_t)p * 2 * ntl, u[p]);
#pragma unroll
      for (int c = 0; c < 8; ++c) x8[c] = xi[c % CPT];
      if constexpr (NP == 4) split8h(x8, wmul, u);
      else split8<NPP>(x8, u);
#pragma unroll
      for (int p = 0; p < NPP; ++p) put(d + (size_t)(NPP + p) * 2 * ntl, u[p]);
    }
  };
  // W[ky] = sum_a R_a e^{-i th_a}, th_a = 2 pi ky a / NY, and its partner W[NY - ky] = sum_a R_a e^{+i th_a} share C = sum_a R_a cos th_a and
  // S = sum_a R_a sin th_a (real weights on the complex row sums: two FMAs per term and sum instead of four):  W[ky] = C - i S,  W[NY - ky] = C + i S.
  // ky = 0 and (NY even) ky = NY / 2 stand alone.  (5x5 layers: 154 -> 95 us per launch; the 9x9 packer is bound by its 3.1 TB/s of writes either way.)
  for (int ky = 0; ky <= NY / 2; ++ky) {
    const int kp = ky == 0 ? 0 : NY - ky;      // partner (== ky for 0 and NY / 2)
    float cs[KS - 1], sn_[KS - 1];
#pragma unroll
    for (int a = 1; a < KS; ++a) { const cf t = twy[ky][a - 1]; cs[a - 1] = t.x; sn_[a - 1] = -t.y; }      // twy = (cos th, -sin th)
    float xr[CPT], xi[CPT], pr[CPT], pi_[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      cf C = ra[c][0], S = cf{0.f, 0.f};
#pragma unroll
      for (int a = 1; a < KS; ++a) { C = sfma(cs[a - 1], ra[c][a], C); S = sfma(sn_[a - 1], ra[c][a], S); }
      xr[c] = C.x + S.y; xi[c] = C.y - S.x;      // C - i S
      pr[c] = C.x - S.y; pi_[c] = C.y + S.x;     // C + i S
    }
    emit(dst + (size_t)ky * fstride, xr, xi);
    if (kp != ky) emit(dst + (size_t)kp * fstride, pr, pi_);
  }
}

// max over (ci, co) of sum_taps |w|: a bound of |W[f][ci][co]| for every frequency (np = 4)
template <int TAPS>
__global__ __launch_bounds__(256) void weight_bound_kernel(const float* __restrict__ w, size_t pairs, int round_bf16, float* __restrict__ wmax) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (size_t)gridDim.x * 256) {
    // every tap's load goes out before the first add (the taps are a compile-time count: round 6 -- with a run-time count the loads went out one by one behind the sum)
    float v[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) v[t] = w[(size_t)t * pairs + e];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      float x = v[t];
      if (round_bf16) x = static_cast<float>(static_cast<__bf16>(x));
      sum += fabsf(x);
    }
    m = fmaxf(m, sum);
  }
  block_max_to<256>(m * 1.0001f, wmax, red, threadIdx.x);      // (a hair of slack for the rounding of this sum)
}

struct Sizes { int NY, NX; };
// transform lengths with a radix plan (all even: the row pass packs two real rows into one complex transform and needs a Nyquist bin)
static const int kLens[] = {20, 24, 28, 32, 36, 40, 50, 60, 64, 72, 96, 100, 128, 192};
static bool pick(int need, int* n) {
  for (int v : kLens)
    if (v >= need) { *n = v; return true; }
  return false;
}
static int tw_offset(int n) {
  int off = 0;
  for (int v : kLens) {
    if (v == n) return off;
    off += v;
  }
  return -1;
}
// circular convolution of size >= (H + pad) x (W + pad); one size per map for every kernel size (pad of the 9x9 layers), so that two
// consecutive layers can hand the row-transformed tensor over.  The old limit H + k - 1 <= 192 is kept.
// circ (ConvArgs::circ): overlap-save windows -- the H x W input IS the transform (H, W must be transform lengths, at least one valid row / column)
static bool sizes_of(int H, int W, int ks, Sizes* s, int circ = 0) {
  if (circ) {
    s->NY = H; s->NX = W;
    return (ks == 9 || ks == 5) && H > 8 && W > 8 && tw_offset(H) >= 0 && tw_offset(W) >= 0;
  }
  return (ks == 9 || ks == 5) && H + ks - 1 <= 192 && W + ks - 1 <= 192 && pick(H + 4, &s->NY) && pick(W + 4, &s->NX);
}
static int pad64(int c) { return (c + CB - 1) / CB * CB; }
// JCM_FFT_REG=0 (environment, read once): the LDS kernels for every pass -- the A/B arm of the register kernels (conv_fft_rows_reg.hip)
static bool fft_reg_on() {
  static const bool on = [] { const char* e = std::getenv("JCM_FFT_REG"); return !e || std::atoi(e) != 0; }();
  return on;
}
static int padn(int c, int n) { return (c + n - 1) / n * n; }

int persistent_grid(const void* kernel, int ntiles, int threads, int dyn_lds) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> cache;      // (kernel, device) -> resident work groups
  int dev = 0;
  (void)hipGetDevice(&dev);
  int resident = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({kernel, dev});
    if (it != cache.end()) resident = it->second;
  }
  if (!resident) {
    int ncu = 256, per_cu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)dyn_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    resident = ncu * per_cu;
    std::lock_guard<std::mutex> lk(mu);
    cache[{kernel, dev}] = resident;
  }
  return resident < ntiles ? resident : ntiles;
}

// e^{+2 pi i k / n} for every supported length, one table per device (built on the host in double precision, uploaded at first use)
static const cf* twiddle_table(int dev) {
  static std::mutex mu;
  static std::map<int, cf*> tabs;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tabs.find(dev);
  if (it != tabs.end()) return it->second;
  std::vector<float> h;
  for (int n : kLens)
    for (int k = 0; k < n; ++k) {
      const double ang = 2.0 * 3.14159265358979323846 * (double)k / (double)n;
      h.push_back((float)std::cos(ang));
      h.push_back((float)std::sin(ang));
    }
  cf* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  tabs[dev] = d;
  return d;
}

}  // namespace cfft

using namespace cfft;

bool conv_fft_supported(const ConvArgs& a, int ks) {
  Sizes s;
  return a.Cin % CB == 0 && a.Cin >= CB && a.Cout >= 1 && a.B >= 1 && sizes_of(a.H, a.W, ks, &s, a.circ);
}
// np: operand form of the channel GEMM (cgemm_split.hip): 4 = fp32 handles, 5 / 2 = bf16 handles
size_t conv_fft_weight_bytes(int H, int W, int ks, int Cin, int Cout, int np, int circ) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s, circ)) return 0;
  return cgemm_split_w_bytes(np, s.NY * (s.NX / 2 + 1), Cin, Cout);
}
hipError_t conv_fft_pack_weights(const float* w_hwio, void* wf, int H, int W, int ks, int Cin, int Cout, int np, bool round_bf16, hipStream_t st, float* wscale, int circ,
                                 const float* bound_from) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s, circ) || Cin % 16 || (np != 2 && np != 4 && np != 5) || (np >= 4 && !wscale) || (np == 5 && Cin % 32)) return hipErrorInvalidValue;
  const int ntl = cgemm_split_ntile(np, Cout), CoutP = padn(Cout, ntl);
  const size_t cpt = 8;      // thread = (8 input channels: one 16-byte unit, output channel); one kx per block
  const dim3 grid((unsigned)(((size_t)Cin / cpt * CoutP + 255) / 256), (unsigned)(s.NX / 2 + 1));
  uint4* dst = static_cast<uint4*>(wf);
  const int rb = round_bf16 ? 1 : 0;
  if (np >= 4 && bound_from) {
    // the bound max sum |taps| of this filter is known: the flipped, transposed filter of a data gradient has the forward filter's
    if (hipError_t e = hipMemcpyAsync(wscale, bound_from, sizeof(float), hipMemcpyDeviceToDevice, st); e != hipSuccess) return e;
  } else if (np >= 4) {
    if (hipError_t e = hipMemsetAsync(wscale, 0, 2 * sizeof(float), st); e != hipSuccess) return e;
    const size_t pairs = (size_t)Cin * Cout;
    const dim3 bgrid((unsigned)((pairs + 255) / 256 > 1024 ? 1024 : (pairs + 255) / 256));
    if (ks == 9) hipLaunchKernelGGL(weight_bound_kernel<81>, bgrid, dim3(256), 0, st, w_hwio, pairs, rb, wscale);
    else hipLaunchKernelGGL(weight_bound_kernel<25>, bgrid, dim3(256), 0, st, w_hwio, pairs, rb, wscale);
  }
#define WS_LAUNCH(KS, NPV) hipLaunchKernelGGL((weight_spectra_split_kernel<KS, NPV>), grid, dim3(256), 0, st, w_hwio, dst, Cin, Cout, CoutP, ntl, s.NY, s.NX, rb, wscale)
  if (ks == 9) {
    if (np == 2) WS_LAUNCH(9, 2); else if (np == 4) WS_LAUNCH(9, 4); else WS_LAUNCH(9, 5);
  } else {
    if (np == 2) WS_LAUNCH(5, 2); else if (np == 4) WS_LAUNCH(5, 4); else WS_LAUNCH(5, 5);
  }
#undef WS_LAUNCH
  return hipGetLastError();
}
// scratch: T (the larger of the two row-transformed tensors) + the split activation spectra Xs + the product spectra Yf
namespace {
struct Plan3 { size_t t_bytes, xs_bytes, yf_bytes, sc_fwd_bytes, sc_inv_bytes; int MT, NXH, F, ldy, inv_cb; };
Plan3 plan_of(const ConvArgs& a, const Sizes& s, int np) {
  Plan3 p;
  p.NXH = s.NX / 2 + 1;
  p.F = s.NY * p.NXH;
  p.ldy = padn(a.Cout, cgemm_split_ntile(np, a.Cout));      // complex numbers per row of the product spectra: whole N tiles, and whole
  if (p.ldy < pad64(a.Cout)) p.ldy = pad64(a.Cout);          // 64-channel blocks of the inverse passes (columns no tile writes are never stored)
  p.MT = cgemm_split_mtile(np, a.B, a.Cout);
  const size_t cop = pad64(a.Cout), cmax = (size_t)a.Cin > cop ? a.Cin : cop;
  const size_t bp = (size_t)(a.B + p.MT - 1) / p.MT * p.MT;
  p.t_bytes = (size_t)a.B * p.NXH * a.H * cmax * sizeof(cf);
  p.xs_bytes = (size_t)p.F * bp * a.Cin * 4 * cgemm_split_parts(np);
  p.yf_bytes = (size_t)p.F * a.B * p.ldy * sizeof(cf);
  // tile scale words of the 16-bit T / T' (np = 5, Fp16Scale::t16): one per (image, row, 64 input channels) / (image, kx, inv_cb output channels)
  p.inv_cb = s.NY > 100 ? 32 : 64;      // colblk<NY>() of the inverse column pass
  p.sc_fwd_bytes = (size_t)a.B * a.H * (a.Cin / CB) * sizeof(float);
  p.sc_inv_bytes = (size_t)a.B * p.NXH * (cop / p.inv_cb) * sizeof(float);
  return p;
}
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}  // namespace
size_t conv_fft_workspace_bytes(const ConvArgs& a, int ks, int np) {
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s, a.circ)) return 0;
  const Plan3 p = plan_of(a, s, np);
  return align256(p.t_bytes) + align256(p.xs_bytes) + align256(p.yf_bytes) + align256(p.sc_fwd_bytes) + align256(p.sc_inv_bytes);
}
// Can layer L (a, ks) hand its output to layer L+1 (kernel size ks_next, same map) in row-transformed form?  Same NX for both kernel
// sizes (always: the size depends on the map only), unpadded channel count, two row buffers in LDS.
bool conv_fft_fusable(const ConvArgs& a, int ks, int ks_next) {
  Sizes s, n;
  return sizes_of(a.H, a.W, ks, &s) && sizes_of(a.H, a.W, ks_next, &n) && s.NX == n.NX && a.Cout % CB == 0;
}
size_t conv_fft_handover_bytes(const ConvArgs& a, int ks) {      // T[kx][c/16][b][y][16] of the next layer
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s)) return 0;
  return (size_t)a.B * (s.NX / 2 + 1) * a.H * a.Cout * sizeof(cf);
}
bool conv_fft_win_gather_supported(int win, int Cin) { return fft_reg_on() && cfft_rows_fwd_win_reg_supported(win, Cin); }
bool conv_fft_win_scatter_supported(int win, int Cout) { return fft_reg_on() && win == 32 && Cout % 64 == 0; }      // rows_inv_reg_kernel<32, 0, false>
// the fused hand-overs across a max pool / the branch merge (FftNext, conv_fft_rows_fused.hip)
bool conv_fft_pool_fusable(const ConvArgs& a, int ks, int ks_next) {
  Sizes s, n;
  return sizes_of(a.H, a.W, ks, &s) && sizes_of((a.H + 1) / 2, (a.W + 1) / 2, ks_next, &n) && cfft_rows_inv_pool_fwd_supported(s.NX, n.NX, a.Cout);
}
size_t conv_fft_pool_handover_bytes(const ConvArgs& a, int ks_next) {      // T[kx][c/16][b][y][16] of the next layer, on the pooled map
  Sizes n;
  if (!sizes_of((a.H + 1) / 2, (a.W + 1) / 2, ks_next, &n)) return 0;
  return (size_t)a.B * (n.NX / 2 + 1) * ((a.H + 1) / 2) * a.Cout * sizeof(cf);
}
// h16: bf16 handles on the one-part route with 16-bit row-transformed tensors -- the register kernel only (the model's geometry)
bool conv_fft_merge_fusable(const ConvArgs& a, int ks, int ks_next, const FftMerge& m, bool h16) {
  Sizes s;
  if (!conv_fft_fusable(a, ks, ks_next) || !sizes_of(a.H, a.W, ks, &s)) return false;
  const bool reg = fft_reg_on() && cfft_rows_inv_merge_fwd_reg_supported(s.NX, a, m, (ks - 1) / 2);
  return h16 ? reg : (reg || cfft_rows_inv_merge_fwd_supported(s.NX, a, m));
}
// a.wp = the split filter spectra of THIS map size and kernel size; `work` = conv_fft_workspace_bytes(a, ks, np) bytes.  g0 / g1: optional
// events recorded around the GEMM (the dominant kernel of the layer) for the roofline record.
// in_layout / out_layout: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar
// t_in (fp32 handles): the row-transformed input left by the previous layer's fused kernel -- the forward row pass is skipped;
// t_next: write the NEXT layer's row-transformed input there instead of the spatial output (conv_fft_fusable() says when that is legal).
// xs (optional): where the split activation spectra of the layer's input live instead of the scratch -- the training step keeps them for the
// weight gradient (wgrad_fft.hip); xs_ready: they are there already (the data gradient after the weight gradient of the same layer): the
// forward transforms are skipped.
hipError_t conv_fft_f32(const ConvArgs& a0, int ks, int np, int in_layout, int out_layout, void* work, const void* t_in, void* t_next, const FftMerge* merge,
                        hipEvent_t g0, hipEvent_t g1, hipStream_t st, void* xs, bool xs_ready, const Fp16Scale* scp, const FftNext* nx) {
  Sizes s;
  if (!conv_fft_supported(a0, ks) || !sizes_of(a0.H, a0.W, ks, &s, a0.circ) || (out_layout == 2 && a0.Cout % 8) || (np != 2 && np != 4 && np != 5) || (np == 5 && a0.Cin % 32)) return hipErrorInvalidValue;
  if (a0.circ && (t_in || t_next || merge || in_layout != 0 || out_layout != 0)) return hipErrorInvalidValue;      // windows: fp32 NHWC in and out, nothing fused
  if (nx && (nx->pool || nx->merge) && (!t_next || (nx->pool && nx->merge))) return hipErrorInvalidValue;
  Fp16Scale sc;
  if (np >= 4) {
    if (!scp || !scp->tmax || !scp->winv || (t_next && !scp->tmax_next)) return hipErrorInvalidValue;
    sc = *scp;
    sc.hf = (float)a0.H;
    sc.nb = a0.B;
  }
  // bf16 handles: the merge hand-over (conv4_fullres -> conv5) in 16-bit form -- t_next / t_in is a complex-fp16 T followed by its tile scale words
  const bool h16 = sc.t16 && ((t_next && nx && nx->merge && !t_in) || (t_in && !t_next));
  if ((t_in || t_next) && !h16 && (in_layout != 0 || out_layout != 0)) return hipErrorInvalidValue;
  if (t_next && a0.Cout % CB) return hipErrorInvalidValue;
  if (sc.t16 && (np != 5 || in_layout == 0 || out_layout == 0 || xs || ((t_in || t_next) && !h16))) return hipErrorInvalidValue;      // 16-bit T / T': bf16 tensors either side, one-part route
  if (merge && in_layout == 2) return hipErrorInvalidValue;      // the merge reads NHWC (fp32 or bf16)
  ConvArgs a = a0;
  a.CoutP = pad64(a.Cout);
  // output row y = row y + pad of the circular convolution (whose size is H + 4 for both kernel sizes); windows: the valid region starts 4 rows into the window
  const int opad = (ks - 1) / 2 + (a0.circ ? 4 : 0);
  const Plan3 p = plan_of(a, s, np);
  char* wk = static_cast<char*>(work);
  cf* T = reinterpret_cast<cf*>(wk);
  void* Xs = xs ? xs : wk + align256(p.t_bytes);
  cf* Yf = reinterpret_cast<cf*>(wk + align256(p.t_bytes) + align256(p.xs_bytes));
  if (sc.t16) {
    sc.t16_fwd = reinterpret_cast<float*>(wk + align256(p.t_bytes) + align256(p.xs_bytes) + align256(p.yf_bytes));
    sc.t16_inv = reinterpret_cast<float*>(wk + align256(p.t_bytes) + align256(p.xs_bytes) + align256(p.yf_bytes) + align256(p.sc_fwd_bytes));
    sc.t16_cb = p.inv_cb;
    // a handed-over 16-bit T carries its scale words behind it (written by the producing layer's fused row kernel)
    if (t_in) sc.t16_fwd = reinterpret_cast<float*>(const_cast<char*>(static_cast<const char*>(t_in)) + align256((size_t)a.B * p.NXH * a.H * a.Cin * 4));
  }
  if (xs_ready && !xs) return hipErrorInvalidValue;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const cf* twb = twiddle_table(dev);
  if (!twb) return hipErrorOutOfMemory;
  const cf* twx = twb + tw_offset(s.NX);
  const cf* twy = twb + tw_offset(s.NY);
  const float norm = 1.0f / (float)(s.NY * s.NX);
  const cf* Tin = t_in ? static_cast<const cf*>(t_in) : T;
  const bool fft_reg = fft_reg_on();
  if (!xs_ready) {
    if (merge && !t_in) {
      if (!(fft_reg && cfft_rows_fwd_merge_reg(s.NX, a, *merge, in_layout, T, sc.tmax, st, sc.t16_fwd))) cfft_rows_fwd_merge(s.NX, a, *merge, in_layout, T, twx, sc.tmax, st, sc.t16_fwd);
    } else if (!t_in) {
      if (a.win_map) { if (!cfft_rows_fwd_win_reg(s.NX, a, T, sc.tmax, st)) return hipErrorInvalidValue; }
      else if (!(fft_reg && cfft_rows_fwd_reg(s.NX, a, in_layout, T, sc.tmax, st, sc.t16_fwd))) cfft_rows_fwd(s.NX, a, in_layout, T, twx, sc.tmax, st, sc.t16_fwd);
    }
    if (hipError_t ce = cfft_cols_fwd(s.NY, a, np, Tin, Xs, twy, p.NXH, p.MT, sc, st); ce != hipSuccess) return ce;
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (g0 && hipEventRecord(g0, st) != hipSuccess) return hipErrorUnknown;
  // 16-bit intermediates (bf16 handles, one-part route): the product spectra travel as complex fp16 too, under the constant shift of cgemm_split.hip
  // (same-box A/B at 256 images: 19.1 -> 18.0 ms per step, profiles/r05_ab_y16.log)
  const float yshift = sc.t16 ? cgemm_split_y16_shift(a.Cin) : 0.f, yinv = yshift != 0.f ? 1.0f / yshift : 0.f;
  if (hipError_t e = cgemm_split(Xs, a.wp, Yf, np, p.F, a.B, a.Cin, a.Cout, p.ldy, st, yshift); e != hipSuccess) return e;
  if (g1 && hipEventRecord(g1, st) != hipSuccess) return hipErrorUnknown;
  ConvArgs ai = a;      // the inverse passes' view: windows keep their valid region only
  if (a0.circ) { ai.H = a0.H - 8; ai.W = a0.W - 8; }
  if (!(fft_reg && p.inv_cb == 64 && cfft_cols_inv_reg(s.NY, ai, Yf, T, p.NXH, p.ldy, opad, st, sc.t16_inv, yinv))) cfft_cols_inv(s.NY, ai, Yf, T, twy, p.NXH, p.ldy, opad, st, sc.t16_inv, yinv);
  if (t_next && nx && nx->pool) {
    Sizes sn;
    if (!sizes_of((a.H + 1) / 2, (a.W + 1) / 2, nx->ks_next, &sn) ||
        !cfft_rows_inv_pool_fwd(s.NX, sn.NX, a, T, static_cast<cf*>(t_next), twx, twb + tw_offset(sn.NX), opad, norm, sc, st))
      return hipErrorInvalidValue;
  } else if (t_next && nx && nx->merge) {
    if (sc.t16) {
      float* t16n = reinterpret_cast<float*>(static_cast<char*>(t_next) + align256((size_t)a.B * p.NXH * a.H * a.Cout * 4));
      if (!(fft_reg && cfft_rows_inv_merge_fwd_reg(s.NX, a, *nx->merge, T, static_cast<cf*>(t_next), opad, norm, sc, st, t16n))) return hipErrorInvalidValue;
    } else if (!(fft_reg && cfft_rows_inv_merge_fwd_reg(s.NX, a, *nx->merge, T, static_cast<cf*>(t_next), opad, norm, sc, st)) &&
               !cfft_rows_inv_merge_fwd(s.NX, a, *nx->merge, T, static_cast<cf*>(t_next), twx, opad, norm, sc, st)) {
      return hipErrorInvalidValue;
    }
  } else if (t_next) {
    if (!(fft_reg && cfft_rows_inv_fwd_reg(s.NX, a, T, static_cast<cf*>(t_next), opad, norm, sc, st))) cfft_rows_inv_fwd(s.NX, a, T, static_cast<cf*>(t_next), twx, opad, norm, sc, st);
  } else if (!((ai.rows_mfma & 1) && cfft_rows_inv_mfma(s.NX, ai, out_layout, T, opad, norm, sc, st)) && !(fft_reg && cfft_rows_inv_reg(s.NX, ai, out_layout, T, opad, norm, sc, st))) {
    if (ai.wout_TX > 0) return hipErrorInvalidValue;      // (the scatter into the map exists in the register kernel only)
    cfft_rows_inv(s.NX, ai, out_layout, T, twx, opad, norm, sc, st);
  }
  return hipGetLastError();
}

// the split spectra of an NHWC fp32 tensor alone (rows + columns forward): xs = conv_fft_xs_bytes() bytes, work = conv_fft_workspace_bytes()
size_t conv_fft_xs_bytes(const ConvArgs& a, int ks, int np) {
  Sizes s;
  if (!sizes_of(a.H, a.W, ks, &s, a.circ)) return 0;
  return align256(plan_of(a, s, np).xs_bytes);
}
hipError_t conv_fft_spectra(const ConvArgs& a0, int ks, int np, void* work, void* xs, hipStream_t st, float* tmax, int common) {
  Sizes s;
  if (!conv_fft_supported(a0, ks) || !sizes_of(a0.H, a0.W, ks, &s, a0.circ) || !xs || (np >= 4 && !tmax)) return hipErrorInvalidValue;
  Fp16Scale sc;
  if (np >= 4) { sc.tmax = tmax; sc.hf = (float)a0.H; sc.nb = a0.B; sc.common = common; }
  ConvArgs a = a0;
  a.CoutP = pad64(a.Cout);
  const Plan3 p = plan_of(a, s, np);
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const cf* twb = twiddle_table(dev);
  if (!twb) return hipErrorOutOfMemory;
  cf* T = static_cast<cf*>(work);
  if (a.win_map) { if (!cfft_rows_fwd_win_reg(s.NX, a, T, sc.tmax, st)) return hipErrorInvalidValue; }
  else cfft_rows_fwd(s.NX, a, 0, T, twb + tw_offset(s.NX), sc.tmax, st);
  if (hipError_t ce = cfft_cols_fwd(s.NY, a, np, T, xs, twb + tw_offset(s.NY), p.NXH, p.MT, sc, st); ce != hipSuccess) return ce;
  return hipGetLastError();
}
// geometry of the spectra for the weight-gradient kernels
bool conv_fft_geometry(int H, int W, int ks, int B, int Cout, int np, int* NY, int* NX, int* MT, int circ) {
  Sizes s;
  if (!sizes_of(H, W, ks, &s, circ)) return false;
  *NY = s.NY; *NX = s.NX; *MT = cgemm_split_mtile(np, B, Cout);
  return true;
}

}  // namespace jcm
