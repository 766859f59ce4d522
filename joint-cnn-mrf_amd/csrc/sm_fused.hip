// The spatial model's 81 pairwise convolutions (main.py:83-87, 117-123) with every transform done in LDS.
//
// (Rounds 1-2 ran these transforms through rocFFT: per forward 10 + 81 spectra per image went to HBM and back four times -- product,
// column pass, pruning transpose, row pass, epilogue -- 10.8 GB per 256 images and 3.3 ms.)  A 120x91 half spectrum is 87 KB: it fits the
// LDS of one CU.  So:
//
//   sm_fwd_spectra_kernel, one work group per (image, channel): softplus(BN(heat map)) -> the 60 nonzero rows of the zero
//     120x180 frame, transformed two rows at a time (z = row_a + i row_b, one 180-point complex FFT, unpacked through the
//     Hermitian symmetry), then the 91 columns (120-point FFTs) -> lhat_t[image][channel][91][120] (0.87 MB per image, the
//     only intermediate that leaves the CU; it stays in the Infinity Cache).
//   sm_inv_finish_kernel, one work group per (image, joint): for each of the joint's 9 pairs, in graph order: spectrum
//     product (prior spectra precomputed at jcm_finalize, [pair][91][120]) -> 91 inverse column FFTs in LDS -> rows 59..119
//     only, again two real rows per complex 180-point inverse FFT -> the 61x91 VALID window -> TF-1.x bilinear 61x91 -> 60x90,
//     + bias, + 1e-6, log, accumulated per pixel in registers; the next pair's spectra are requested while the current
//     pair's transforms run.  Nothing but the logits is written.
//
// FFTs: in-place decimation in frequency along the contiguous axis in TWO stages, 120 = 8 x 15 and 180 = 12 x 15 (the 12- and
// 15-point butterflies are prime-factor compositions of the 3-, 4- and 5-point ones: no twiddles inside a butterfly, one
// twiddle per point between the stages); the output of a transform sits at its digit-reversed position, which the consumer
// folds into its read address (pos120 / pos180).  Row pitches 121 / 181 keep the strided accesses of the transposing steps
// bank-conflict free.  The kernels are VALU-bound (a pair is 0.8 M lane-instructions), not LDS- or HBM-bound.
// Unnormalised transforms, 1/(120*180) folded into the product;
// the imaginary parts of the DC / Nyquist bins of a row are dropped as a C2R transform drops them.
#include <atomic>
#include <cstdlib>

#include "sm_lds_fft.h"

namespace jcm {


using namespace smf;

// lhat_t[b][c][v][u] = sum_{y<60, x<90} lik[b][c][y][x] e^{-2 pi i (u y / 120 + v x / 180)},  v < 91
__global__ __launch_bounds__(NT) void sm_fwd_spectra_kernel(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld,
                                                              const float* __restrict__ sc, const float* __restrict__ sh, float2* __restrict__ lhat_t, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* cb = lds + CB;
  cf* rb = lds + RB;
  const int tid = threadIdx.x;
  const int b = blockIdx.x / C, c = blockIdx.x - b * C;
  make_twiddles(lds, tid);
  for (int i = tid; i < NROWP * PX; i += NT) rb[i] = cf{0.f, 0.f};
  for (int i = tid; i < WC * PU; i += NT) cb[i] = cf{0.f, 0.f};
  __syncthreads();
  float* rbf = reinterpret_cast<float*>(rb);
  for (int pix = tid; pix < MHW; pix += NT) {
    const int y = pix / MW, x = pix - y * MW;
    rbf[((y >> 1) * PX + x) * 2 + (y & 1)] = lik_of(hm, Ca, extra, extra_ld, sc, sh, (int64_t)b * MHW + pix, c);
  }
  __syncthreads();
  fft180<PX, -1, MH / 2>(rb, lds + TW180, tid);
  // Z_i = FFT(row 2i + i row 2i+1):  X_a[k] = (Z[k] + conj Z[-k]) / 2,  X_b[k] = (Z[k] - conj Z[-k]) / (2i)  ->  cb[k][2i], cb[k][2i+1]
  for (int t = tid; t < (MH / 2) * WC; t += NT) {
    const int i = t / WC, k = t - i * WC;
    const cf zk = rb[i * PX + pos180(k)], zn = rb[i * PX + pos180(k == 0 ? 0 : FW - k)];
    cb[k * PU + 2 * i] = cf{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    cb[k * PU + 2 * i + 1] = cf{0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
  }
  __syncthreads();
  fft120<PU, -1, WC>(cb, lds + TW120, tid);
  cf* __restrict__ out = reinterpret_cast<cf*>(lhat_t) + (size_t)blockIdx.x * (WC * FH);
  for (int e = tid; e < WC * FH; e += NT) {
    const int v = e / FH, u = e - v * FH;
    out[e] = cb[v * PU + pos120(u)];
  }
}

// logits[b][pix][j] = log(lik[b][j][pix] + 1e-6) + sum over the C-1 pairs p of joint j, graph order, of
//                     log(R(lik[b][cond p] * prior p)[pix] + spb[p][pix] + 1e-6)                        (main.py:117-123)
//
// PERSISTENT and BALANCED (round 6).  The unit of work is one (image, joint, pair); a work group owns the 135 KB of LDS of its CU, so one runs per CU, and
// rounds 2-5 launched one work group per (image, joint): 64 images = 576 work groups on 256 CUs = 2.25 waves, run as 3 (391 us where 272 would do).
// Now G work groups (one per CU, at most one per (image, joint)) take the U = B K (C-1) units in G contiguous, equal ranges.  A range of >= C-1 units starts
// inside at most one (image, joint) item and ends inside at most one; the sum of an item's log terms is kept IN GRAPH ORDER across the cut:
//   (the ranges: eight XCD shares of whole items, each cut into equal ranges for that XCD's work groups -- see the kernel body)
//   * the work group whose range ENDS inside an item does that HEAD first: e = log(lik + 1e-6) + terms 0 .. q-1, stores e in its partial-sum slot and
//     raises its flag (release at device scope);
//   * the work group whose range STARTS inside the item does that TAIL last: by then the flag of its predecessor (the previous slot of the same XCD's
//     queue: dispatched earlier, and it publishes after at most C-2 pairs) has long been raised; it reads the partial sums (acquire) and goes on adding terms q .. C-2 in order.
// Every logit is therefore the same fp32 sum, term by term, as in the one-work-group-per-item kernel, whatever the batch size cuts where.
// Flags carry the launch's epoch (a counter of the handle), so they are never reset.
__global__ __launch_bounds__(NT) void sm_inv_finish_kernel(const float* __restrict__ hm, int Ca, const float* __restrict__ extra, int extra_ld,
                                                             const float* __restrict__ sc, const float* __restrict__ sh,
                                                             const float2* __restrict__ lhat_t, const float2* __restrict__ phat_t,
                                                             const int* __restrict__ cond, const float* __restrict__ spb, float* __restrict__ logits, int K,
                                                             int C, float* __restrict__ tsave, int nunits, float* __restrict__ part, unsigned* __restrict__ flags,
                                                             unsigned epoch, int perm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* cb = lds + CB;
  cf* rb = lds + RB;
  const int tid = threadIdx.x;
  const int PJ = C - 1;
  make_twiddles(lds, tid);
  // TF-1.x bilinear 61x91 -> 60x90 (main.py:89), the arithmetic of sm_fft.hip, separable: per output row {offset of window rows ylo |
  // yhi << 16 in the row buffer, weight ty}, per output column {offset of window columns xlo | xhi << 16, weight tx}.  Window row yy is
  // frame row 59 + yy = row pair yy >> 1, real / imaginary part; window column xx is frame column 89 + xx.
  for (int t = tid; t < MH + MW; t += NT) {
    const bool isy = t < MH;
    const int o = isy ? t : t - MH;
    const float f = __fmul_rn((float)o, isy ? 61.0f / 60.0f : 91.0f / 90.0f);
    const int lo = (int)floorf(f), hi = min(lo + 1, isy ? 60 : 90);
    const int olo = isy ? ((lo >> 1) * PX) * 2 + (lo & 1) : 2 * pos180(89 + lo);
    const int ohi = isy ? ((hi >> 1) * PX) * 2 + (hi & 1) : 2 * pos180(89 + hi);
    lds[TY + t] = cf{__uint_as_float((unsigned)olo | ((unsigned)ohi << 16)), f - (float)lo};
  }
  constexpr int NG = NT / WC;                       // packing step: thread = (column pk, row-pair group pg)
  const int pk = tid % WC, pg = tid / WC;

  // this work group's units [u0, u1).  Work group i runs on XCD i % 8 (slot i / 8 of that XCD's in-order queue).
  //   perm 2 -- one work group per item, items dealt so that an XCD owns WHOLE IMAGES: work group i takes joint n % K of image (i % 8) + 8 (n / K), n = i / 8.
  //             The 32 work groups an XCD runs at a time are then the 9 joints of 3-4 images, walking their pairs in step: at a step they want 2 of an
  //             image's 10 likelihood spectra and one prior per joint -- ~16 distinct spectra for 64 loads, served by that XCD's L2 instead of the fabric.
  //   perm 0 -- the balanced cut (G a multiple of 8): the ITEMS are split eight ways, one contiguous share per XCD, and an XCD's share of units is cut into
  //             equal ranges for its G / 8 slots.  A range's predecessor is therefore work group i - 8: the previous slot of the SAME XCD queue, placed on
  //             a CU before this one whatever else shares the GPU -- a work group never waits for one that has not been dispatched, so two handles'
  //             kernels (two streams, two processes) cannot hold each other's CUs in a cycle.  No range crosses an XCD's share: slot 0 has no tail, the
  //             last slot no head.
  const int G = gridDim.x, w = (int)blockIdx.x;
  int u0, u1;
  if (perm == 2) {
    const int n = w / 8, b = w % 8 + 8 * (n / K);
    if (b * K >= nunits / PJ) return;      // (the last images of a batch that is not a multiple of 8)
    u0 = (b * K + n % K) * PJ;
    u1 = u0 + PJ;
  } else if (perm == 1) {      // one work group per item, in launch order (the A/B arm of 2)
    u0 = w * PJ;
    u1 = u0 + PJ;
  } else {
    const int nitems = nunits / PJ, xcd = w % 8, slot = w / 8, S = G / 8;
    const int lo = (int)((long long)nitems * xcd / 8) * PJ, hi = (int)((long long)nitems * (xcd + 1) / 8) * PJ;
    u0 = lo + (int)((long long)(hi - lo) * slot / S);
    u1 = lo + (int)((long long)(hi - lo) * (slot + 1) / S);
  }
  const int i0 = u0 / PJ, q0 = u0 - i0 * PJ;                       // first item, first pair of it
  const int i1 = (u1 - 1) / PJ, q1 = (u1 - 1) - i1 * PJ + 1;       // last item, one past its last pair
  // A range holds >= PJ units (host), so it starts inside at most one item (its TAIL: pairs q0 ..) and ends inside at most one (its HEAD: pairs .. q1 - 1),
  // and a range inside ONE item is that whole item.  Processing order: the head first, the whole items, the tail last -- ONE flat loop over the units in
  // that order (everything that places a unit is wave-uniform: scalar registers), so that the next unit's spectra are requested across segment boundaries too.
  const bool tail = q0 > 0, head = q1 < PJ;
  const int f0 = tail ? i0 + 1 : i0;                           // first whole item
  const int nh = head ? q1 : 0, nt = tail ? PJ - q0 : 0;       // units of the head / tail segment
  const int n = u1 - u0, nmid = n - nh - nt;                   // nmid = whole items x PJ
  auto place = [&](int k, int& item, int& qq) __attribute__((always_inline)) {
    if (k < nh) { item = i1; qq = k; }
    else if (k < nh + nmid) { const int m = k - nh; item = f0 + m / PJ; qq = m - (m / PJ) * PJ; }
    else { item = i0; qq = q0 + (k - nh - nmid); }
  };
  cf l[NE], q[NE];
  auto request = [&](int k) __attribute__((always_inline)) {
    int item, qq;
    place(k, item, qq);
    const int b = item / K, p = (item - b * K) * PJ + qq;
    const cf* __restrict__ ls = reinterpret_cast<const cf*>(lhat_t) + ((size_t)b * C + cond[p]) * (WC * FH);
    const cf* __restrict__ qs = reinterpret_cast<const cf*>(phat_t) + (size_t)p * (WC * FH);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int idx = tid + i * NT;
      if (idx < WC * FH) { l[i] = ls[idx]; q[i] = qs[idx]; }
    }
  };
  if (n > 0) request(0);
  float e[NPIX];
  for (int k = 0; k < n; ++k) {
    int item, qq;
    place(k, item, qq);
    const int b = item / K, j = item - b * K, p = j * PJ + qq;
    const bool in_head = k < nh, in_tail = k >= nh + nmid;
    const bool seg_first = in_tail ? k == nh + nmid : qq == 0;
    const bool seg_last = in_head ? k == nh - 1 : qq == PJ - 1;
    // the thread index as this iteration sees it: opaque to the optimiser, so that the pixel coordinates, tap offsets and addresses derived from it are
    // recomputed here (a few integer instructions) instead of being hoisted out of the loop and spilled (62 scratch accesses per unit otherwise)
    int tix = tid;
    asm volatile("" : "+v"(tix));
    if (seg_first) {
      if (in_tail) {
        // the partial sums of the previous slot of this XCD (work group w - 8: its head of this item): wait for its flag, then read them at device scope
        if (tid == 0) {
          while (__hip_atomic_load(flags + (w - 8), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* __restrict__ ps = part + (size_t)(w - 8) * MHW;
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
          const int pix = tix + i * NT;
          e[i] = pix < MHW ? __builtin_nontemporal_load(ps + pix) : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
          const int pix = tix + i * NT;
          e[i] = pix < MHW ? logf(lik_of(hm, Ca, extra, extra_ld, sc, sh, (int64_t)b * MHW + pix, j) + 1e-6f) : 0.f;
        }
      }
    }
    // spectrum product -> cb[v][u] (the previous pair's column buffer was last read before the barrier that closed its packing step)
    constexpr float scale = 1.0f / (float)(FH * FW);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int idx = tid + i * NT;
      if (idx < WC * FH) cb[idx + idx / FH] = scale * cmul(l[i], q[i]);       // [v][u] with pitch 121 = 120 + 1
    }
    __syncthreads();
    fft120<PU, 1, WC>(cb, lds + TW120, tid);
    // rows 59 + 2i (real part) and 60 + 2i (imaginary part) of the frame as ONE complex inverse transform: Z = X_a + i X_b with the
    // Hermitian extension X[180 - k] = conj X[k]: one read of X_a[k], X_b[k] gives Z[k] and Z[180 - k]
    if (pg < NG) {
      const bool edge = pk == 0 || pk == FW / 2;       // DC / Nyquist: real by symmetry; a C2R transform ignores their imaginary parts
      for (int i = pg; i < NROWP; i += NG) {
        cf xa = cb[pk * PU + pos120(59 + 2 * i)];
        cf xb = i < 30 ? cb[pk * PU + pos120(60 + 2 * i)] : cf{0.f, 0.f};
        if (edge) { xa.y = 0.f; xb.y = 0.f; }
        rb[i * PX + pk] = cf{xa.x - xb.y, xa.y + xb.x};
        if (!edge) rb[i * PX + FW - pk] = cf{xa.x + xb.y, xb.x - xa.y};
      }
    }
    __syncthreads();
    if (k + 1 < n) request(k + 1);      // lands behind the row transforms and the epilogue (the registers are free of the column butterflies now)
    fft180<PX, 1, NROWP>(rb, lds + TW180, tid);
    // VALID window Cpre[yy][xx] = frame[59 + yy][89 + xx] -> resize -> + bias, + 1e-6, log
    const float* rbf = reinterpret_cast<const float*>(rb);
    const float* __restrict__ bias = spb + (size_t)p * MHW;
#pragma unroll
    for (int i = 0; i < NPIX; ++i) {
      const int pix = tix + i * NT;
      if (pix < MHW) {
        const int oy = pix / MW, ox = pix - oy * MW;
        const cf cy = lds[TY + oy], cx = lds[TX + ox];
        const unsigned ry = __float_as_uint(cy.x), rx = __float_as_uint(cx.x);
        const unsigned rlo = ry & 0xffffu, rhi = ry >> 16, plo = rx & 0xffffu, phi = rx >> 16;
        const float tl = rbf[rlo + plo], tr = rbf[rlo + phi];
        const float bl = rbf[rhi + plo], br = rbf[rhi + phi];
        const float top = tl + (tr - tl) * cx.y;
        const float bot = bl + (br - bl) * cx.y;
        const float cv = top + (bot - top) * cy.y;
        const float tv = (cv + bias[pix]) + 1e-6f;
        if (tsave) tsave[((size_t)b * (K * PJ) + p) * MHW + pix] = tv;      // training step: the log's argument is the backward pass's denominator
        e[i] += __logf(tv);      // v_log_f32 * ln 2: the argument is a normal number >= 1e-6
      }
    }
    __syncthreads();      // the row buffer is rewritten by the next pair's packing step (its column buffer is free already)
    if (seg_last) {
      if (in_head) {
        float* __restrict__ ps = part + (size_t)w * MHW;
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
          const int pix = tix + i * NT;
          if (pix < MHW) __builtin_nontemporal_store(e[i], ps + pix);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();      // every thread's partial sums are out (and released) before the flag goes up
        if (tid == 0) __hip_atomic_store(flags + w, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
          const int pix = tix + i * NT;
          if (pix < MHW) logits[((size_t)b * MHW + pix) * K + j] = e[i];
        }
      }
    }
  }
}

// work groups of sm_inv_finish_kernel that are resident at once on this device (one per CU: 135 KB of LDS each)
static int sm_inv_resident() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int r = cached[dev & 63].load(std::memory_order_relaxed);
  if (r > 0) return r;
  int ncu = 256, per_cu = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sm_inv_finish_kernel, NT, (size_t)LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
  r = ncu * per_cu;
  cached[dev & 63].store(r, std::memory_order_relaxed);
  return r;
}
size_t sm_fused_scratch_bytes() { return (size_t)sm_inv_resident() * (MHW * sizeof(float) + 64); }      // partial sums + one flag word (a cache line) per work group

hipError_t sm_fused_forward(const float* hm, int Ca, const float* extra, int extra_ld, const float* sc, const float* sh, const float2* phat_t,
                            const int* cond, const float* spbias, float2* lhat_t, float* logits, int B, int K, int C, hipStream_t st, float* tsave, void* scratch,
                            unsigned epoch) {
  if (extra_ld <= 0) extra_ld = C - Ca;
  if (!scratch || epoch == 0) return hipErrorInvalidValue;
  static LdsAttr attr_f, attr_i;
  if (hipError_t e = attr_f.ensure(reinterpret_cast<const void*>(sm_fwd_spectra_kernel), LDS_BYTES); e != hipSuccess) return e;
  if (hipError_t e = attr_i.ensure(reinterpret_cast<const void*>(sm_inv_finish_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(sm_fwd_spectra_kernel, dim3(B * C), dim3(NT), LDS_BYTES, st, hm, Ca, extra, extra_ld, sc, sh, lhat_t, C);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  // Grid: one work group per (image, joint) item when the items fill whole rounds of the resident work groups (or fewer than one), the balanced
  // cut otherwise.  Measured (tools/sm_time.py, round 6): 64 images = 2.25 rounds: 0.369 ms balanced (G = 256) against 0.398 ms with 576 work groups (0.45 ms
  // with round 5's kernel); 256 images = 9.0 rounds: 1.20 ms with 2304 work groups against 1.43 ms with 256 persistent ones -- work groups that walk their
  // units in lockstep issue their spectrum loads in bursts, which costs ~15 % per unit, so the cut has to save more than that.
  const int resident = sm_inv_resident();
  const int items = B * K, rounds = (items + resident - 1) / resident;
  int G = items;
  if (items > resident && (double)rounds * resident > 1.15 * (double)items) G = resident;      // every range then holds >= C - 1 units (the kernel's cut rule)
  if (const char* e = std::getenv("JCM_SM_G")) { const int g = std::atoi(e); if (g > 0 && g <= items && (items % g == 0 || g <= resident)) G = g; }      // (tools/sm_time.py sweeps)
  // the cut needs eight XCD shares of whole slots (G a multiple of 8, every range >= C - 1 units: items / 8 >= G / 8); otherwise one work group per item
  if (G != items && (G % 8 != 0 || items / 8 < G / 8)) G = items;
  static const int perm_env = [] { const char* e = std::getenv("JCM_SM_PERM"); return e ? std::atoi(e) : 2; }();      // (1: items in launch order -- the A/B arm of 2)
  const int perm = G == items ? (perm_env == 2 ? 2 : 1) : 0;
  if (perm == 2) G = 8 * ((B + 7) / 8) * K;
  float* part = static_cast<float*>(scratch);
  unsigned* flags = reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + (size_t)resident * MHW * sizeof(float));
  hipLaunchKernelGGL(sm_inv_finish_kernel, dim3(G), dim3(NT), LDS_BYTES, st, hm, Ca, extra, extra_ld, sc, sh, lhat_t, phat_t, cond, spbias, logits, K, C, tsave, B * K * (C - 1),
                     part, flags, epoch, perm);
  return hipGetLastError();
}


hipError_t sm_fused_spectra(const float* hm, int Ca, const float* extra, int extra_ld, const float* sc, const float* sh, float2* lhat_t, int B, int C,
                            hipStream_t st) {
  if (extra_ld <= 0) extra_ld = C - Ca;
  static LdsAttr attr_f;
  if (hipError_t e = attr_f.ensure(reinterpret_cast<const void*>(sm_fwd_spectra_kernel), LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(sm_fwd_spectra_kernel, dim3(B * C), dim3(NT), LDS_BYTES, st, hm, Ca, extra, extra_ld, sc, sh, lhat_t, C);
  return hipGetLastError();
}

}  // namespace jcm
