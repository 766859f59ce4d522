// libjcm C ABI (include/jcm.h): context, parameter store, weight packing, workspace arena and
// the forward graph of main.py:29-74,94-125,522-531 as a sequence of kernel launches on one
// HIP stream.  No tensor library types cross this boundary -- plain pointers and sizes.
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include <mutex>
#include <vector>

#include "ctx.h"

using namespace jcm;

namespace {
thread_local std::string g_err;
}

namespace jcm {

const char* const kJointNames[10] = {"lsho", "lelb", "lwri", "rsho", "relb", "rwri", "lhip", "rhip", "nose", "torso"};

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

namespace {
struct Chain { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool armed = false; };
Chain g_chain[64];
}  // namespace
// Handles whose OUTERMOST entry point is running on this host thread.  A call is NESTED only when the same thread re-enters the handle (from the
// gradient-ready callback of its running training step); another thread's call on the same handle waits on the handle's mutex until the whole
// outermost call -- callbacks included -- has returned (the de-facto serialisation of rounds 1-4, without the data race on call_depth).
namespace {
thread_local std::vector<jcm_ctx*> t_active;
}
CallOrder::CallOrder(jcm_ctx* ctx) : c(ctx) {
  if (!c) return;
  for (jcm_ctx* a : t_active) nested = nested || a == c;
  if (!nested) {
    hlk = std::unique_lock<std::mutex>(c->call_mu);      // held for the whole outermost call
    t_active.push_back(c);
  }
  ++c->call_depth;                   // (only ever touched by the thread that holds call_mu)
  acquire();
  if (nested) return;                // the outer call's hand-over fields and scale words stay as they are
  c->order = this;
  // what an aborted call may have left behind
  c->fft_t_in = nullptr; c->fft_t_next = nullptr; c->fft_merge = nullptr; c->fft_xs = nullptr; c->fft_xs_ready = false; c->fft_tmax_in = nullptr;
  c->fft_next_pool = 0; c->fft_next_ks = 0; c->fft_next_merge = nullptr; c->fft_t_in_16 = false; c->fft_win_map = nullptr; c->fft_win_scatter = false;
  // the fp16-scale words are reused from the start only BETWEEN calls (a call keeps words of its early layers until its last ones: the training step)
  if (c->fft_block_i > 0 || c->fft_word_i > jcm_ctx::kFftWords - jcm_ctx::kFftWordsPerCall) {
    for (int i = 0; i <= c->fft_block_i && i < (int)c->fft_blocks.size(); ++i)
      (void)hipMemsetAsync(c->fft_blocks[i].p, 0, (size_t)c->fft_blocks[i].cap * sizeof(float), c->stream);
    c->fft_block_i = 0;
    c->fft_word_i = 0;
  }
}
void CallOrder::acquire() {
  if (!c || lk.owns_lock()) return;
  Chain& ch = g_chain[c->device & 63];
  lk = std::unique_lock<std::mutex>(ch.mu);      // held while the call is queuing kernels: two host threads never interleave their launches
  if (c->call_order) {
    if (!ch.ev && hipEventCreateWithFlags(&ch.ev, hipEventDisableTiming) != hipSuccess) ch.ev = nullptr;
    if (ch.ev && ch.armed && ch.last != c->stream) (void)hipStreamWaitEvent(c->stream, ch.ev, 0);
  }
}
void CallOrder::release() {
  if (!c || !lk.owns_lock()) return;
  if (c->call_order) {
    Chain& ch = g_chain[c->device & 63];
    if (ch.ev && hipEventRecord(ch.ev, c->stream) == hipSuccess) { ch.armed = true; ch.last = c->stream; }
  }
  lk.unlock();
}
CallOrder::~CallOrder() {
  if (!c) return;
  release();
  --c->call_depth;
  if (!nested) {
    c->order = nullptr;
    for (size_t i = t_active.size(); i-- > 0;)
      if (t_active[i] == c) { t_active.erase(t_active.begin() + (long)i); break; }
    hlk.unlock();
  }
}

int arena_reserve(jcm_ctx* c, size_t bytes) {
  if (bytes <= c->arena_cap) return JCM_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->arena) HIP_TRY(hipFree(c->arena));
  c->arena = nullptr;
  c->arena_cap = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->arena), bytes));
  c->arena_cap = bytes;
  return JCM_OK;
}

int dev_alloc(jcm_ctx* c, void** p, size_t bytes) {
  HIP_TRY(hipMalloc(p, bytes));
  c->owned.push_back(*p);
  c->param_bytes += bytes;
  return JCM_OK;
}

// the balanced spatial-model kernel's partial sums + flags (sm_fused.hip): allocated and zeroed once per handle; every launch takes the next epoch
int sm_scratch_next(jcm_ctx* c, void** scratch, unsigned* epoch) {
  if (!c->sm_scratch) {
    const size_t n = sm_fused_scratch_bytes();
    JCM_TRY(dev_alloc(c, &c->sm_scratch, n));
    HIP_TRY(hipMemsetAsync(c->sm_scratch, 0, n, c->stream));
  }
  if (++c->sm_epoch == 0) {      // (the counter wrapped: flags of 2^32 launches ago could match -- start over from zeroed flags)
    HIP_TRY(hipMemsetAsync(c->sm_scratch, 0, sm_fused_scratch_bytes(), c->stream));
    c->sm_epoch = 1;
  }
  *scratch = c->sm_scratch;
  *epoch = c->sm_epoch;
  return JCM_OK;
}

const Tensor* find(jcm_ctx* c, const std::string& name) {
  auto it = c->params.find(name);
  return it == c->params.end() ? nullptr : &it->second;
}

// Inference BatchNorm folded to y = x*scale + shift:  scale = gamma*rsqrt(var+eps), shift = beta-mean*scale.
// Buffers are allocated on the first call and rewritten in place afterwards (training refresh).
int fold_bn(jcm_ctx* c, const std::string& scope, int n, float** scale, float** shift) {
  const Tensor* t[4];
  static const char* const kNames[4] = {"gamma", "beta", "moving_mean", "moving_variance"};
  for (int i = 0; i < 4; ++i) {
    const std::string name = scope + "/BatchNorm/" + kNames[i];
    t[i] = find(c, name);
    if (!t[i]) return fail(JCM_ERR_STATE, "missing parameter '" + name + "'");
    if (t[i]->n != (size_t)n) return fail(JCM_ERR_STATE, "parameter '" + name + "' has " + std::to_string(t[i]->n) + " elements, expected " + std::to_string(n));
  }
  if (!*scale) JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(scale), n * sizeof(float)));
  if (!*shift) JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(shift), n * sizeof(float)));
  HIP_TRY(bn_fold(t[0]->d, t[1]->d, t[2]->d, t[3]->d, kBnEps, *scale, *shift, n, c->stream));
  return JCM_OK;
}

int check(jcm_handle h, bool need_final) {
  if (!h) return fail(JCM_ERR_ARG, "null handle");
  if (need_final && !h->finalized) return fail(JCM_ERR_STATE, "jcm_finalize has not been called");
  return JCM_OK;
}

const ConvLayer* conv_of(jcm_ctx* c, const std::string& scope) {
  auto it = c->convs.find(scope);
  return it == c->convs.end() ? nullptr : &it->second;
}

// The launch itself (kernel choice by precision / f32_conv); run_conv_layer brackets it with the timing events.
static int launch_conv_layer(jcm_ctx* c, const ConvLayer* L, const void* wp, const void* x, int B, int H, int W, void* out, bool act_bf16,
                             bool out_f32, int in_planar, int out_planar) {
  ConvArgs a{};
  a.x = x; a.wp = wp; a.bias = L->bias; a.scale = L->scale; a.shift = L->shift; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout; a.relu_bn = L->has_bn ? 1 : 0;
  a.in_planar = in_planar; a.out_planar = out_planar;
  a.hpool = c->conv_hpool;
  c->conv_hpool = 0;
  if ((in_planar || out_planar) && !act_bf16) return fail(JCM_ERR_ARG, "planar activations exist on the bf16 path only");
  if (a.hpool && !act_bf16) return fail(JCM_ERR_STATE, "half pool requested on an fp32 layer");
  if (act_bf16) {
    a.CoutP = L->coutp_bf16;
    if (a.hpool && (out_f32 || L->thin_bf16 || L->ks != 5 || conv_igemm_bf16_bn(L->cout, L->ks) != 128 || !conv5_strip_bf16_supported(a, L->ks)))
      return fail(JCM_ERR_STATE, "half pool requested for a layer that does not run on conv5_strip_bf16_kernel");
    if (L->thin_bf16 && out_f32 && L->wp_kxfold && conv_kxfold_bf16_supported(a, L->ks)) {
      a.wp = L->wp_kxfold;
      HIP_TRY(conv_kxfold_bf16(a, c->stream));
    } else if (L->thin_bf16 && out_f32) {
      HIP_TRY(conv_thin_bf16(a, c->stream));
    }
    else HIP_TRY(conv_igemm_bf16(a, L->ks, out_f32, c->stream));
  } else {
    a.CoutP = L->coutp;
    const bool use_split = L->wp_split && (L->thin ? c->f32_conv == 2 : conv_split_supported(L->ks, L->cin, L->coutp_split, B, H, W, c->split_min_wgs));
    if (use_split) {     // fp16x3: lift this input into the fp16 range by its own power-of-two scale
      HIP_TRY(pow2_scale_of(static_cast<const float*>(x), (size_t)B * H * W * L->cin, c->act_scale, c->scale_scratch, c->stream));
      a.in_scale = c->act_scale;
      a.w_scale = L->wscale;
    }
    if (L->thin && L->wp_split && c->f32_conv == 2) {
      a.wp = L->wp_split;
      a.CoutP = 16;
      HIP_TRY(conv_thin_split16(a, c->stream));
    } else if (L->thin) {
      HIP_TRY(conv_thin_f32(a, c->stream));
    } else if (use_split) {
      a.wp = L->wp_split;
      a.CoutP = L->coutp_split;
      HIP_TRY(conv_split_f32(a, L->ks, 2, c->stream));
    } else {
      HIP_TRY(conv_igemm_f32(a, L->ks, c->stream));
    }
  }
  return JCM_OK;
}

static int pool_get(jcm_ctx* c, hipEvent_t* e) {
  if (!c->event_pool.empty()) {
    *e = c->event_pool.back();
    c->event_pool.pop_back();
    return JCM_OK;
  }
  HIP_TRY(hipEventCreate(e));
  return JCM_OK;
}
int prof_begin(jcm_ctx* c, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!c->profile) return JCM_OK;
  JCM_TRY(pool_get(c, e0));
  if (int r = pool_get(c, e1); r != JCM_OK) { c->event_pool.push_back(*e0); *e0 = nullptr; return r; }
  hipError_t e = hipEventRecord(*e0, c->stream);
  if (e != hipSuccess) {
    prof_end(c, "", *e0, *e1, false);
    *e0 = *e1 = nullptr;
    return fail(JCM_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  return JCM_OK;
}
// ok: the launch went out -> record the closing event and file the pair under `scope`; otherwise return both to the pool
void prof_end(jcm_ctx* c, const std::string& scope, hipEvent_t e0, hipEvent_t e1, bool ok) {
  if (!e0) return;
  if (ok && hipEventRecord(e1, c->stream) == hipSuccess) {
    c->prof[scope].emplace_back(e0, e1);
  } else {
    c->event_pool.push_back(e0);
    c->event_pool.push_back(e1);
  }
}
void prof_release_all(jcm_ctx* c, bool destroy) {
  for (auto& kv : c->prof)
    for (auto& ev : kv.second) { c->event_pool.push_back(ev.first); c->event_pool.push_back(ev.second); }
  c->prof.clear();
  if (destroy) {
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->event_pool.clear();
  }
}

// Does this stride-1 layer run in the frequency domain (conv_fft.hip)?  fp32 handles: inference and the training step (forward and data
// gradient; the filter spectra are recomputed after every update -- refresh_derived invalidates them); bf16 handles: inference only.
bool takes_fft(jcm_ctx* c, const ConvLayer* L, int B, int H, int W) {
  if (!c->conv9_fft || c->f32_conv != 0 || (c->train && c->precision != JCM_PRECISION_F32) || (L->ks != 9 && L->ks != 5) || L->cin == 3 || !L->w_raw) return false;
  // bf16 handles: the wide 9x9 layers only.  Round 5 measured the 5x5 layers of a bf16 handle on this route at B = 256 (HIP events per layer, same box):
  // conv2 (64 -> 128) 1.95 / 0.53 / 0.14 ms on conv5_strip_bf16_kernel against 4.20 / 0.90 / 0.25 ms here (its 128 output channels make the fp32
  // product spectra 6 of its 15 GB); conv3 (128 -> 256) 1.82 / 0.50 / 0.17 against 1.73 / 0.53 / 0.18 ms -- break-even, and the tower's error
  // against the bf16-operand oracle grows from 4.3e-3 to 5.9e-3 of the logit scale: both stay on the strip kernels.
  if (c->precision == JCM_PRECISION_BF16 && (L->ks != 9 || L->thin_bf16 || L->cout % 8)) return false;
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout;
  return conv_fft_supported(a, L->ks);
}
bool fft_spectra_valid(jcm_ctx* c, const std::string& scope, int H, int W, int circ) {
  auto it = c->fft_w.find(scope + (circ ? "@win" : "@") + std::to_string(H) + "x" + std::to_string(W));
  return it != c->fft_w.end() && it->second.valid;
}
// n zeroed device words (one per image of a row-transformed tensor).  Blocks are zeroed when they are created and every time the handle starts
// over at the first one (CallOrder: between calls, in stream order, behind every kernel that read the old words); a word is handed out once per lap.
int fft_new_words(jcm_ctx* c, int n, float** w) {
  if (n < 1) return fail(JCM_ERR_ARG, "fft_new_words: n < 1");
  for (;;) {
    if (c->fft_block_i < (int)c->fft_blocks.size()) {
      jcm_ctx::WordBlock& b = c->fft_blocks[c->fft_block_i];
      if (c->fft_word_i + n <= b.cap) {
        *w = b.p + c->fft_word_i;
        c->fft_word_i += n;
        return JCM_OK;
      }
      ++c->fft_block_i;      // the rest of this block stays unused until the next lap
      c->fft_word_i = 0;
      continue;
    }
    jcm_ctx::WordBlock b;
    b.cap = n > jcm_ctx::kFftWords ? n : jcm_ctx::kFftWords;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.p), (size_t)b.cap * sizeof(float)));
    if (hipError_t e = hipMemsetAsync(b.p, 0, (size_t)b.cap * sizeof(float), c->stream); e != hipSuccess) {
      (void)hipFree(b.p);
      return fail(JCM_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    }
    c->fft_blocks.push_back(b);
  }
}
int run_conv_fft(jcm_ctx* c, const ConvLayer* L, const std::string& scope, const void* x, int B, int H, int W, void* out, int in_layout, int out_layout, int circ) {
  ConvArgs a{};
  a.x = x; a.bias = L->bias; a.scale = L->scale; a.shift = L->shift; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout; a.CoutP = L->cout; a.relu_bn = L->has_bn ? 1 : 0;
  a.circ = circ;
  a.rows_mfma = c->fft_rows_mfma;
  const size_t mark = c->arena_off;
  const int np = fft_np(c);      // operand form of the channel GEMM (cgemm_split.hip)
  void* work = arena_alloc<char>(c, conv_fft_workspace_bytes(a, L->ks, np));
  c->arena_off = mark;                                   // scratch of this layer only: later layers run behind it on the stream
  if (c->dry) { c->fft_t_in = nullptr; c->fft_t_next = nullptr; c->fft_merge = nullptr; c->fft_xs = nullptr; c->fft_xs_ready = false; c->fft_tmax_in = nullptr; c->fft_next_pool = 0; c->fft_next_ks = 0; c->fft_next_merge = nullptr; c->fft_t_in_16 = false; c->fft_win_map = nullptr; c->fft_win_scatter = false; return JCM_OK; }
  // Filter spectra are cached per (layer, map size).  The cache is bounded (JCM_FFT_CACHE_GB, default 64): a caller that walks many
  // image sizes (7.7 GB per size for conv5) makes it drop every spectrum that is not this layer's before it grows past the bound.
  const std::string key = scope + (circ ? "@win" : "@") + std::to_string(H) + "x" + std::to_string(W);
  // (A training handle keeps the spectra of BOTH geometries of a layer -- overlap-save windows for steps of <= 32 images, the whole map for evaluation
  // forwards and larger batches -- so that a loop that alternates training steps and evaluation does not re-pack gigabytes and stall the stream at
  // every flip (round 5 dropped the other geometry here); the cache bound below is what limits the footprint.)
  if (!c->fft_w.count(key)) {
    static const size_t cap = [] { const char* e = std::getenv("JCM_FFT_CACHE_GB"); return (size_t)(e ? std::atoi(e) : 64) << 30; }();
    const size_t need = conv_fft_weight_bytes(H, W, L->ks, L->cin, L->cout, np, circ);
    size_t held = 0;
    for (auto& kv : c->fft_w) held += kv.second.bytes;
    if (held + need > cap && !c->fft_w.empty()) {
      HIP_TRY(hipStreamSynchronize(c->stream));            // earlier layers of this forward may still read theirs
      for (auto& kv : c->fft_w) (void)hipFree(kv.second.p);
      c->fft_w.clear();
    }
  }
  jcm_ctx::FftW& fw = c->fft_w[key];
  if (!fw.p) {
    const size_t wb = (conv_fft_weight_bytes(H, W, L->ks, L->cin, L->cout, np, circ) + 255) & ~size_t(255);
    fw.bytes = wb + 256;      // + the two words of the filter spectra's scale (np = 4)
    if (hipMalloc(&fw.p, fw.bytes) == hipSuccess) {
      fw.wscale = reinterpret_cast<float*>(static_cast<char*>(fw.p) + wb);
    } else {
      const size_t mb = fw.bytes >> 20;
      c->fft_w.erase(key);
      return fail(JCM_ERR_HIP, "out of device memory for the filter spectra of '" + scope + "' (" + std::to_string(mb) + " MB); jcm_set_option(\"conv9_fft\", 0) selects the direct kernels");
    }
  }
  if (!fw.valid) {
    // a data gradient's pseudo-layer ("dgrad:<scope>", jcm_train.hip) holds the flipped, transposed filter of <scope>: the same set of taps per
    // (ci, co) pair, hence the same bound -- taken from the forward spectra of the same geometry when they are valid (always, inside a step)
    const float* bound_from = nullptr;
    if (np >= 4 && scope.compare(0, 6, "dgrad:") == 0) {
      auto it = c->fft_w.find(key.substr(6));
      if (it != c->fft_w.end() && it->second.valid && it->second.wscale) bound_from = it->second.wscale;
    }
    HIP_TRY(conv_fft_pack_weights(L->w_raw, fw.p, H, W, L->ks, L->cin, L->cout, np, c->precision == JCM_PRECISION_BF16, c->stream, fw.wscale, circ, bound_from));
    fw.valid = true;
  }
  a.wp = fw.p;
  hipEvent_t e0 = nullptr, e1 = nullptr, g0 = nullptr, g1 = nullptr;
  JCM_TRY(prof_begin(c, &e0, &e1));
  if (c->profile && (pool_get(c, &g0) != JCM_OK || pool_get(c, &g1) != JCM_OK)) { g0 = g1 = nullptr; }
  const void* t_in = c->fft_t_in;
  void* t_next = c->fft_t_next;
  const FftMerge* mg = static_cast<const FftMerge*>(c->fft_merge);
  void* xs = c->fft_xs;
  const bool xs_ready = c->fft_xs_ready;
  FftNext nx;
  nx.pool = c->fft_next_pool; nx.ks_next = c->fft_next_ks; nx.merge = static_cast<const FftMerge*>(c->fft_next_merge);
  c->fft_next_pool = 0; c->fft_next_ks = 0; c->fft_next_merge = nullptr;
  const bool t_in_16 = c->fft_t_in_16;
  c->fft_t_in_16 = false;
  if (c->fft_win_map) {      // the windows are gathered by the forward row pass
    a.win_map = c->fft_win_map; a.win_B = c->fft_win_B; a.win_H = c->fft_win_H; a.win_W = c->fft_win_W; a.win_TY = c->fft_win_TY; a.win_TX = c->fft_win_TX;
    c->fft_win_map = nullptr;
  }
  if (c->fft_win_scatter) {      // ... and scattered by the inverse row pass (the geometry fields stay valid without a gather)
    a.wout_H = c->fft_win_H; a.wout_W = c->fft_win_W; a.wout_TY = c->fft_win_TY; a.wout_TX = c->fft_win_TX;
    c->fft_win_scatter = false;
  }
  Fp16Scale sc;
  if (np >= 4) {
    // the word of this layer's input: handed over with t_in / ready spectra, or a fresh one for this layer's own row pass
    sc.tmax = c->fft_tmax_in;
    if ((t_in || xs_ready) && !sc.tmax) return fail(JCM_ERR_STATE, "conv_fft '" + scope + "': a handed-over tensor without its scale word");
    if (!sc.tmax) JCM_TRY(fft_new_words(c, B, &sc.tmax));
    if (t_next) JCM_TRY(fft_new_words(c, B, &sc.tmax_next));
    sc.winv = fw.wscale + 1;
    sc.common = c->train ? 1 : 0;      // a handle with training state: one scale per tensor (the weight gradient sums over the images)
    // 16-bit T / T' between the row and column passes: bf16 tensors on both sides of the layer, one-part spectra, nothing handed over or kept
    // ... except the merge hand-over conv4_fullres -> conv5 of jcm_pd_forward, which exists in 16-bit form (rows_inv_merge_fwd_reg_kernel<.., true>)
    sc.t16 = (np == 5 && c->fft_t16 && in_layout != 0 && out_layout != 0 && !xs && (!t_in || t_in_16) && (!t_next || (nx.merge && !t_in))) ? 1 : 0;
  }
  c->fft_t_in = nullptr; c->fft_t_next = nullptr; c->fft_merge = nullptr; c->fft_xs = nullptr; c->fft_xs_ready = false;
  c->fft_tmax_in = sc.tmax_next;      // the next frequency-domain layer takes t_next (and its word)
  c->fft_last_tmax = sc.tmax;
  const hipError_t e = conv_fft_f32(a, L->ks, np, in_layout, out_layout, work, t_in, t_next, mg, g0, g1, c->stream, xs, xs_ready, np >= 4 ? &sc : nullptr, &nx);
  if (g0 && g1 && e == hipSuccess) c->prof[scope + "/gemm"].emplace_back(g0, g1);
  else { if (g0) c->event_pool.push_back(g0); if (g1) c->event_pool.push_back(g1); }
  prof_end(c, scope, e0, e1, e == hipSuccess);
  if (e != hipSuccess) return fail(JCM_ERR_HIP, std::string("conv_fft_f32: ") + hipGetErrorString(e));
  return JCM_OK;
}

// One conv layer.  Activations are fp32, or bf16 when the handle runs the bf16 path (`act_bf16`);
// `out_f32` forces an fp32 result (the logits layer).
int run_conv_layer(jcm_ctx* c, const ConvLayer* L, const std::string& scope, int stride, const void* x, int B, int H, int W, int sub,
                   void* out, bool act_bf16, bool out_f32, int in_planar, int out_planar) {
  if (stride == 2) {
    if (c->dry) return JCM_OK;
    if (!(L->ks == 5 && L->cin == 3 && L->has_bn))
      return fail(JCM_ERR_ARG, "stride-2 kernel exists for 5x5, Cin=3, BN layers only (" + scope + ")");
    HIP_TRY(conv1_5x5s2(static_cast<const float*>(x), L->w_raw, L->bias, L->scale, L->shift, out, act_bf16, B, H, W, sub,
                        L->cout, c->stream));
    return JCM_OK;
  }
  if (stride == 1 && takes_fft(c, L, B, H, W))
    return run_conv_fft(c, L, scope, x, B, H, W, out, act_bf16 ? (in_planar ? 2 : 1) : 0, (act_bf16 && !out_f32) ? (out_planar ? 2 : 1) : 0);
  if (c->dry) return JCM_OK;
  const void* wp = act_bf16 ? L->wp_bf16 : static_cast<const void*>(L->wp);
  if (stride != 1 || !wp) return fail(JCM_ERR_ARG, "no kernel for layer '" + scope + "' with stride " + std::to_string(stride));
  if (!act_bf16 && L->wp_stale) {
    HIP_TRY(pack_weights_f32(L->w_raw, L->wp, L->ks, L->cin, L->cout, L->coutp, c->stream));
    L->wp_stale = false;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  JCM_TRY(prof_begin(c, &e0, &e1));
  const int r = launch_conv_layer(c, L, wp, x, B, H, W, out, act_bf16, out_f32, in_planar, out_planar);
  prof_end(c, scope, e0, e1, r == JCM_OK);
  return r;
}

}  // namespace jcm

namespace jcm {

// fp16x3: {Sw, 1/Sw} with Sw the power of two that brings max|w| just below 2^14 (device scalars, recomputed at every refresh)
static int weight_scale(jcm_ctx* c, const Tensor& w, float** wscale) {
  if (!c->scale_scratch) {
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->scale_scratch), 1024 * sizeof(float)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->act_scale), 2 * sizeof(float)));
  }
  if (!*wscale) JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(wscale), 2 * sizeof(float)));
  HIP_TRY(pow2_scale_of(w.d, w.n, *wscale, c->scale_scratch, c->stream));
  return JCM_OK;
}

int refresh_derived(jcm_ctx* c, bool first) {
  for (auto& kv : c->fft_w) kv.second.valid = false;     // filter spectra follow the weights: recomputed on next use
  // ---- conv layers: every "<scope>/weights" of rank 4
  for (auto& kv : c->params) {
    const std::string& name = kv.first;
    const std::string suffix = "/weights";
    if (name.size() <= suffix.size() || name.compare(name.size() - suffix.size(), suffix.size(), suffix) != 0) continue;
    const Tensor& w = kv.second;
    if (w.shape.size() != 4 || w.shape[0] != w.shape[1]) return fail(JCM_ERR_ARG, "'" + name + "' must be [k,k,Cin,Cout]");
    const std::string scope = name.substr(0, name.size() - suffix.size());
    ConvLayer L;
    if (!first) {
      auto it = c->convs.find(scope);
      if (it == c->convs.end()) return fail(JCM_ERR_STATE, "conv layer '" + scope + "' appeared after jcm_finalize");
      L = it->second;
    }
    L.ks = (int)w.shape[0]; L.cin = (int)w.shape[2]; L.cout = (int)w.shape[3];
    L.w_raw = w.d;
    const Tensor* b = find(c, scope + "/biases");
    if (!b || b->n != (size_t)L.cout) return fail(JCM_ERR_STATE, "missing or mis-sized '" + scope + "/biases'");
    L.bias = b->d;
    L.has_bn = find(c, scope + "/BatchNorm/gamma") != nullptr;
    if (L.has_bn) JCM_TRY(fold_bn(c, scope, L.cout, &L.scale, &L.shift));
    if ((L.ks == 5 || L.ks == 9) && L.cin % 16 == 0 && c->precision == JCM_PRECISION_F32) {
      L.thin = L.ks == 9 && L.cout <= 12;            // logits layer: 4x4x1_16b MFMA kernel, channels padded to 16
      const int bn = L.thin ? 16 : conv_igemm_bn(L.cout);
      L.coutp = (L.cout + bn - 1) / bn * bn;
      const size_t n = (size_t)L.ks * L.ks * L.cin * L.coutp;
      if (!L.wp) JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&L.wp), n * sizeof(float)));
      // after a weight update (training step) the packing waits until a direct kernel reads it: layers on the frequency-domain route never do
      if (first) HIP_TRY(pack_weights_f32(w.d, L.wp, L.ks, L.cin, L.cout, L.coutp, c->stream));
      else L.wp_stale = true;
    }
    if (c->precision == JCM_PRECISION_F32 && c->f32_conv == 2 && (L.ks == 9 || L.ks == 5) && L.cin % 16 == 0 && L.cout % 128 == 0) {
      const int ns = 2;      // operand parts of the direct split kernels: two fp16 parts, three products (fp16x3)
      L.coutp_split = L.cout;
      if (!L.wp_split) JCM_TRY(dev_alloc(c, &L.wp_split, conv_split_weight_bytes(L.ks, L.cin, L.coutp_split, ns)));
      JCM_TRY(weight_scale(c, w, &L.wscale));
      HIP_TRY(pack_weights_split(w.d, L.wp_split, L.ks, L.cin, L.cout, L.coutp_split, ns, c->stream, L.wscale));
    }
    if (c->precision == JCM_PRECISION_F32 && c->f32_conv == 2 && L.ks == 9 && L.cout <= 16 && L.cin % 32 == 0) {   // logits layer, fp16x3
      L.coutp_split = 16;
      if (!L.wp_split) JCM_TRY(dev_alloc(c, &L.wp_split, conv_split_weight_bytes(L.ks, L.cin, 16, 2)));
      JCM_TRY(weight_scale(c, w, &L.wscale));
      HIP_TRY(pack_weights_split(w.d, L.wp_split, L.ks, L.cin, L.cout, 16, 2, c->stream, L.wscale));
    }
    if (L.ks == 5 && L.cin == 3 && L.cout == 64 && L.has_bn && c->precision == JCM_PRECISION_F32) {
      if (!L.wq1_f32) JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&L.wq1_f32), 5 * 16 * 64 * sizeof(float)));
      HIP_TRY(pack_conv1_f32(w.d, L.wq1_f32, c->stream));
      if (!L.wq1_split) JCM_TRY(dev_alloc(c, &L.wq1_split, conv1_split_weight_bytes()));
      HIP_TRY(pack_conv1_split(w.d, L.wq1_split, c->stream));
    }
    if (L.ks == 5 && L.cin == 3 && L.cout == 64 && L.has_bn && c->precision == JCM_PRECISION_BF16) {
      if (!L.wq1_bf16) JCM_TRY(dev_alloc(c, &L.wq1_bf16, 5 * 2 * 64 * 16));
      HIP_TRY(pack_conv1_bf16(w.d, L.wq1_bf16, c->stream));
    }
    if ((L.ks == 5 || L.ks == 9) && c->precision == JCM_PRECISION_BF16 && L.cin != 3) {
      if (L.cin % 32 != 0) return fail(JCM_ERR_ARG, "bf16 path needs Cin % 32 == 0 ('" + scope + "' has " + std::to_string(L.cin) + ")");
      L.thin_bf16 = L.ks == 9 && L.cout <= 16 && !L.has_bn;   // logits layer: 16x16x32 MFMA kernel, fp32 out
      const int bn = L.thin_bf16 ? 16 : conv_igemm_bf16_bn(L.cout, L.ks);
      L.coutp_bf16 = (L.cout + bn - 1) / bn * bn;
      const size_t n = (size_t)L.ks * L.ks * L.cin * L.coutp_bf16;
      if (!L.wp_bf16) JCM_TRY(dev_alloc(c, &L.wp_bf16, n * 2));
      HIP_TRY(pack_weights_bf16(w.d, L.wp_bf16, L.ks, L.cin, L.cout, L.coutp_bf16, c->stream));
      if (L.thin_bf16 && L.cout == 9) {              // the logits layer's second packing: kernel columns folded into N
        if (!L.wp_kxfold) JCM_TRY(dev_alloc(c, &L.wp_kxfold, conv_kxfold_weight_bytes(L.cin)));
        HIP_TRY(pack_weights_kxfold(w.d, L.wp_kxfold, L.cin, c->stream));
      }
    }
    c->convs[scope] = L;
  }
  // ---- spatial model tables (main.py:477-487): pairs in graph order
  if (find(c, "bn_sm/BatchNorm/gamma")) {
    const int P = c->K * (kC - 1);
    JCM_TRY(fold_bn(c, "bn_sm", kC, &c->bn_sm_scale, &c->bn_sm_shift));
    if (first) {
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->sp_energy), (size_t)P * kPrH * kPrW * sizeof(float)));
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->sp_bias), (size_t)P * kHmHW * sizeof(float)));
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->cond), (size_t)P * sizeof(int)));
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->prior_spec_t), (size_t)P * kSpec * sizeof(float2)));
    }
    if (first) {
      std::vector<int> cond(P);
      std::vector<const float*> ep(P), bp(P);
      int p = 0;
      for (int j = 0; j < c->K; ++j) {
        for (int cc = 0; cc < kC; ++cc) {
          if (cc == j) continue;
          const std::string key = std::string(kJointNames[j]) + "_" + kJointNames[cc];
          const Tensor* e = find(c, "energy_" + key);
          const Tensor* bi = find(c, "bias_" + key);
          if (!e || e->n != (size_t)kPrH * kPrW) return fail(JCM_ERR_STATE, "missing or mis-sized 'energy_" + key + "' (want [1,120,180,1])");
          if (!bi || bi->n != (size_t)kHmHW) return fail(JCM_ERR_STATE, "missing or mis-sized 'bias_" + key + "' (want [1,60,90,1])");
          ep[p] = e->d;
          bp[p] = bi->d;
          cond[p++] = cc;
        }
      }
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->energy_ptrs), P * sizeof(float*)));
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->bias_ptrs), P * sizeof(float*)));
      HIP_TRY(hipMemcpyAsync(c->cond, cond.data(), P * sizeof(int), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(c->energy_ptrs, ep.data(), P * sizeof(float*), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(c->bias_ptrs, bp.data(), P * sizeof(float*), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));   // the tables are stack-local
    }
    HIP_TRY(sm_softplus5_multi(c->energy_ptrs, c->sp_energy, P, (int64_t)kPrH * kPrW, c->stream));   // main.py:120
    HIP_TRY(sm_softplus5_multi(c->bias_ptrs, c->sp_bias, P, kHmHW, c->stream));                        // main.py:122
    HIP_TRY(sm_lds_fwd_frames(c->sp_energy, c->prior_spec_t, P, c->stream));      // [pair][91][120]: the layout every consumer reads
    c->has_sm = true;
  }
  if (first) {
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&c->cond0), sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->cond0, 0, sizeof(int), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return JCM_OK;
}

}  // namespace jcm

namespace {

int run_conv(jcm_ctx* c, const std::string& scope, int stride, const void* x, int B, int H, int W, int sub, void* out,
             bool act_bf16, bool out_f32, int in_planar = 0, int out_planar = 0) {
  const ConvLayer* L = conv_of(c, scope);
  if (!L) return fail(JCM_ERR_STATE, "no conv layer '" + scope + "' (set '" + scope + "/weights' and finalize)");
  return run_conv_layer(c, L, scope, stride, x, B, H, W, sub, out, act_bf16, out_f32, in_planar, out_planar);
}

// bf16 handles: does a [B,H,W,Cin] launch of this 9x9 layer take the flattened-strip kernel (which reads / writes the
// planar activation layout at full speed)?
bool takes_strip(const ConvLayer* L, int B, int H, int W) {
  if (!L->wp_bf16 || L->thin_bf16 || conv_igemm_bf16_bn(L->cout, L->ks) != 256) return false;
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout; a.CoutP = L->coutp_bf16;
  a.out_planar = 1;
  return conv_strip_bf16_supported(a, L->ks);
}

// will this bf16 5x5 layer run on conv5_strip_bf16_kernel (which reads and writes either activation layout)?
bool takes_c5strip(const ConvLayer* L, int B, int H, int W) {
  if (!L->wp_bf16 || L->ks != 5 || conv_igemm_bf16_bn(L->cout, L->ks) != 128 || L->cout % 8) return false;
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout; a.CoutP = L->coutp_bf16;
  return conv5_strip_bf16_supported(a, L->ks);
}

// fp32 handles: two consecutive frequency-domain layers on the same map -- the first one's fused inverse/forward row kernel writes the
// second one's row-transformed input (from the arena) and the activation between them never reaches HBM.  Call right before
// run_conv(first); returns the buffer to pass to expect_handover() before run_conv(second), or null.
static void* offer_handover(jcm_ctx* c, const ConvLayer* La, const ConvLayer* Lb, int B, int H, int W) {
  if (c->precision != JCM_PRECISION_F32 || !takes_fft(c, La, B, H, W) || !takes_fft(c, Lb, B, H, W)) return nullptr;
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = La->cin; a.Cout = La->cout;
  if (La->cout != Lb->cin || !conv_fft_fusable(a, La->ks, Lb->ks)) return nullptr;
  void* t = arena_alloc<char>(c, conv_fft_handover_bytes(a, La->ks));
  c->fft_t_next = t;
  return t;
}
// ... with the 2x2 max pool of main.py:47,55,64 between them: La runs on H x W, Lb on the pooled map; the fused kernel (conv_fft_rows_fused.hip) pools
// a row pair in LDS and writes Lb's row-transformed input -- neither La's output nor the pooled map reaches HBM.
static void* offer_pool_handover(jcm_ctx* c, const ConvLayer* La, const ConvLayer* Lb, int B, int H, int W) {
  if (!(c->fft_fuse & 1) || c->precision != JCM_PRECISION_F32 || !takes_fft(c, La, B, H, W) || !takes_fft(c, Lb, B, (H + 1) / 2, (W + 1) / 2)) return nullptr;
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = La->cin; a.Cout = La->cout;
  if (La->cout != Lb->cin || !conv_fft_pool_fusable(a, La->ks, Lb->ks)) return nullptr;
  void* t = arena_alloc<char>(c, conv_fft_pool_handover_bytes(a, Lb->ks));
  c->fft_t_next = t;
  c->fft_next_pool = 1;
  c->fft_next_ks = Lb->ks;
  return t;
}
// model(x, n_joints), main.py:29-74.  x fp32 NHWC; intermediate activations fp32 or bf16.

int pd_forward_impl(jcm_ctx* c, const float* x, int B, int H, int W, float* logits) {
  static const char* const kRes[3] = {"fullres", "halfres", "quarterres"};
  const ConvLayer* L4 = conv_of(c, "conv4_fullres");
  const ConvLayer* L5 = conv_of(c, "conv5");
  if (!L4 || !L5 || !conv_of(c, "conv6")) return fail(JCM_ERR_STATE, "part-detector parameters incomplete");
  const bool bf = c->precision == JCM_PRECISION_BF16;
  const size_t es = bf ? 2 : 4;
  auto act = [&](size_t elems) { return static_cast<void*>(arena_alloc<char>(c, elems * es)); };
  void* x4[3];
  int h4[3], w4[3];
  for (int r = 0; r < 3; ++r) {
    const int sub = 1 << r;
    h4[r] = cdiv2(cdiv2(cdiv2(H / sub)));                        // resize_images(x, [H//2, W//2]) main.py:51,60
    w4[r] = cdiv2(cdiv2(cdiv2(W / sub)));
  }
  const int hh = h4[0], ww = w4[0];
  const ConvLayer* L6 = conv_of(c, "conv6");
  // fp32 handles, model geometry: the full-resolution branch's conv4 hands conv5 the row-transformed MERGED map (conv_fft_rows_fused.hip) -- x1 is never
  // written.  The coarse branches then have to be there first: the branches run half, quarter, full.
  FftMerge mg{nullptr, h4[1], w4[1], nullptr, h4[2], w4[2]};
  bool fuse45 = false;
  // bf16 handles: the same hand-over in 16-bit form (one-part route with 16-bit row-transformed tensors, NHWC bf16 branches)
  const bool h16 = bf && fft_np(c) == 5 && c->fft_t16;
  if ((!bf || h16) && !c->debug_skip && takes_fft(c, L4, B, hh, ww) && takes_fft(c, L5, B, hh, ww) && L4->cout == L5->cin) {
    ConvArgs a{};
    a.B = B; a.H = hh; a.W = ww; a.Cin = L4->cin; a.Cout = L4->cout;
    fuse45 = (c->fft_fuse & 2) && conv_fft_merge_fusable(a, L4->ks, L5->ks, mg, h16);
  }
  // branch outputs survive the per-branch scratch, so carve them first
  void* t45 = nullptr;
  for (int r = 0; r < 3; ++r) {
    if (r == 0 && fuse45) {
      ConvArgs a{};
      a.B = B; a.H = hh; a.W = ww; a.Cin = L4->cin; a.Cout = L4->cout;
      t45 = arena_alloc<char>(c, conv_fft_handover_bytes(a, L4->ks));
      x4[0] = nullptr;
    } else {
      x4[r] = act((size_t)B * h4[r] * w4[r] * L4->cout);
    }
  }
  mg.x2 = x4[1]; mg.x3 = x4[2];
  // bf16: the 9x9 chain (conv3 out -> conv4 -> merge -> conv5 -> conv6 in) runs on planar activations [B][C/8][H*W][8]
  // when conv5 takes the strip kernel; every producer / consumer on that chain handles the layout.
  const int planar = bf && L6->thin_bf16 && L4->cout % 8 == 0 && L5->cout % 8 == 0 && takes_strip(L5, B, h4[0], w4[0]) ? 1 : 0;
  // ... except between two frequency-domain layers: their row passes read and write NHWC in whole 128-byte lines per pixel, while a planar
  // tensor gives a lane only the 4 bytes of its channel pair inside a 16-byte unit (3.3 against 4.8 TB/s measured for the inverse row pass).
  // So with conv5 in the frequency domain the chain conv4 -> merge -> conv5 is NHWC; conv5's OUTPUT stays planar for the logits kernel.
  const int planar45 = planar && !takes_fft(c, L5, B, h4[0], w4[0]) ? 1 : 0;
  static const int kOrder[3] = {1, 2, 0};
  for (int ri = 0; ri < 3; ++ri) {
    const int r = kOrder[ri];
    const size_t mark = c->arena_off;
    const std::string res = kRes[r];
    const int sub = 1 << r;
    const int hin = H / sub, win = W / sub;
    const ConvLayer* L1 = conv_of(c, "conv1_" + res);
    const ConvLayer* L2 = conv_of(c, "conv2_" + res);
    const ConvLayer* L3 = conv_of(c, "conv3_" + res);
    if (!L1 || !L2 || !L3) return fail(JCM_ERR_STATE, "part-detector parameters incomplete (" + res + ")");
    const float* xin = x;
    int xh = H, xw = W, xsub = sub;
    if (H % sub || W % sub) {   // non-integer scale: a real bilinear resize, not sub-sampling
      float* xr = arena_alloc<float>(c, (size_t)B * hin * win * 3);
      if (!c->dry) HIP_TRY(resize_bilinear(x, xr, B, H, W, 3, hin, win, c->stream));
      xin = xr; xh = hin; xw = win; xsub = 1;
    }
    const int h1 = cdiv2(hin), w1 = cdiv2(win);
    const int h2 = cdiv2(h1), w2 = cdiv2(w1);
    void* p1;
    const int sk = c->debug_skip;
    if (sk & 1) { p1 = act((size_t)B * h2 * w2 * L1->cout); }
    else if (!bf && L1->wq1_f32 && xh % (4 * xsub) == 0 && xw % (4 * xsub) == 0) {
      // fp32 path: conv1 + ReLU/BN + pool1 in one fp32-MFMA kernel (the unpooled 240x360x64 map never reaches HBM)
      p1 = act((size_t)B * h2 * w2 * L1->cout);
      // default route (the stride-1 layers run on split operands on the bf16 matrix cores): conv1 too; the exact fp32 MFMA chain otherwise
      const bool split1 = c->conv9_fft && c->f32_conv == 0 && L1->wq1_split;
      if (!c->dry)
        HIP_TRY(split1 ? conv1_mfma_pool_split(xin, L1->wq1_split, L1->bias, L1->scale, L1->shift, static_cast<float*>(p1), B, xh, xw, xsub, c->stream)
                       : conv1_mfma_pool_f32(xin, L1->wq1_f32, L1->bias, L1->scale, L1->shift, static_cast<float*>(p1), B, xh, xw, xsub, c->stream));
    } else if (bf && L1->wq1_bf16 && xh % (4 * xsub) == 0 && xw % (4 * xsub) == 0) {
      // bf16 path: conv1 + ReLU/BN + pool1 in one MFMA kernel; only the pooled map touches HBM
      p1 = act((size_t)B * h2 * w2 * L1->cout);
      if (!c->dry)
        HIP_TRY(conv1_mfma_pool(xin, L1->wq1_bf16, L1->bias, L1->scale, L1->shift, p1, B, xh, xw, xsub, c->stream));  // :44-45,52-53,61-62
    } else {
      void* c1 = act((size_t)B * h1 * w1 * L1->cout);
      JCM_TRY(run_conv(c, "conv1_" + res, 2, xin, B, xh, xw, xsub, c1, bf, false));       // main.py:44,52,61
      p1 = act((size_t)B * h2 * w2 * L1->cout);
      if (!c->dry) HIP_TRY(max_pool_2x2(c1, p1, bf, B, h1, w1, L1->cout, c->stream));       // :45,53,62
    }
    const int h3 = cdiv2(h2), w3 = cdiv2(w2);
    // fp32 handles: conv2 -> pool2 -> conv3 as one hand-over in row-transformed form (the pool inside the fused row kernel)
    void* t23 = (bf || sk) ? nullptr : offer_pool_handover(c, L2, L3, B, h2, w2);
    void* c2 = t23 ? nullptr : act((size_t)B * h2 * w2 * L2->cout);
    // bf16: conv2 -> pool2 -> conv3 on planar activations when both 5x5 layers take the strip kernel (its window rows are then 1-KB
    // contiguous LDS-DMA reads; from NHWC every 16-byte unit of a pixel is a separate cache line).  A planar [B][C/8][H][W][8] tensor IS an
    // NHWC tensor of B*C/8 images with 8 channels: the pooling kernel runs on it unchanged.
    const int pl23 = bf && takes_c5strip(L2, B, h2, w2) && takes_c5strip(L3, B, h3, w3) ? 1 : 0;
    // ... and the pool's horizontal half is taken in conv2's epilogue (even widths): c2 is then the [.., h2, w2 / 2, ..] map of pixel-pair maxima
    const int hp = pl23 && c->bf16_hpool && w2 % 2 == 0 && !(sk & 6) ? 1 : 0;
    c->conv_hpool = hp;
    if (!(sk & 4)) JCM_TRY(run_conv(c, "conv2_" + res, 1, p1, B, h2, w2, 1, c2, bf, false, 0, pl23));     // :46,54,63
    c->conv_hpool = 0;
    void* p2 = t23 ? nullptr : act((size_t)B * h3 * w3 * L2->cout);
    if (!c->dry && !(sk & 2) && !t23) {                                                     // :47,55,64
      if (hp) HIP_TRY(vpool_2x1_bf16(c2, p2, B * (L2->cout / 8), h2, w2 / 2, 8, c->stream));
      else if (pl23) HIP_TRY(max_pool_2x2(c2, p2, bf, B * (L2->cout / 8), h2, w2, 8, c->stream));
      else HIP_TRY(max_pool_2x2(c2, p2, bf, B, h2, w2, L2->cout, c->stream));
    }
    const ConvLayer* L4r = conv_of(c, "conv4_" + res);
    if (!L4r) return fail(JCM_ERR_STATE, "part-detector parameters incomplete (conv4_" + res + ")");
    const int in4 = planar && L3->cout % 8 == 0 && takes_strip(L4r, B, h3, w3) && !takes_fft(c, L4r, B, h3, w3) ? 1 : 0;      // the patch kernels and the row pass read NHWC
    void* t34 = (sk & 24) ? nullptr : offer_handover(c, L3, L4r, B, h3, w3);      // (no hand-over when either side is left out)
    void* c3 = t34 ? nullptr : act((size_t)B * h3 * w3 * L3->cout);
    c->fft_t_in = t23;
    if (!(sk & 8)) JCM_TRY(run_conv(c, "conv3_" + res, 1, p2, B, h3, w3, 1, c3, bf, false, pl23, in4));   // :48,56,65
    c->fft_t_in = t34;
    if (r == 0 && fuse45) { c->fft_t_next = t45; c->fft_next_merge = &mg; }      // conv4_fullres writes conv5's row-transformed (merged) input
    if (!(sk & 16)) JCM_TRY(run_conv(c, "conv4_" + res, 1, c3, B, h3, w3, 1, x4[r], bf, false, in4, planar45));   // :49,57,66
    c->arena_off = mark;
  }
  // conv5 in the frequency domain: its forward row kernel forms ((x1 + up(x2)) + up(x3)) / 3 while it loads the rows (NHWC inputs: fp32, or
  // bf16 on a bf16 handle, where the merged value is rounded to bf16 as the separate merge kernel's output would be) -- unless conv4_fullres
  // handed the row-transformed merged map over already (fuse45, fp32 handles).
  // (Round 5 measured the alternative for bf16 handles -- the merge as its own bandwidth-bound kernel + conv5's register row pass: 20.45 against
  // 20.11 ms per 256-image step with the fused kernel, three interleaved runs each: writing and re-reading the 1.4 GB merged tensor costs more
  // than the fused kernel's slower rows.)
  const bool fuse_merge = !fuse45 && takes_fft(c, L5, B, hh, ww) && !planar45;
  void* merged = fuse45 ? nullptr : fuse_merge ? x4[0] : act((size_t)B * hh * ww * L4->cout);
  if (!c->dry && !fuse45 && !fuse_merge && !(c->debug_skip & 32)) {                        // :58,67,69-70
    if (planar45) HIP_TRY(upsample_merge3_planar(x4[0], x4[1], h4[1], w4[1], x4[2], h4[2], w4[2], merged, B, hh, ww, L4->cout, c->stream));
    else HIP_TRY(upsample_merge3(x4[0], x4[1], h4[1], w4[1], x4[2], h4[2], w4[2], merged, bf, B, hh, ww, L4->cout, c->stream));
  }
  const int sk = c->debug_skip;
  void* t56 = (sk & 96) ? nullptr : offer_handover(c, L5, conv_of(c, "conv6"), B, hh, ww);
  void* c5 = t56 ? nullptr : act((size_t)B * hh * ww * L5->cout);
  if (fuse_merge && !(sk & 32)) c->fft_merge = &mg;
  if (fuse45) { c->fft_t_in = t45; c->fft_t_in_16 = h16; }
  if (!(sk & 32)) JCM_TRY(run_conv(c, "conv5", 1, merged, B, hh, ww, 1, c5, bf, false, planar45, planar));   // :71
  c->fft_t_in = t56;
  if (!(sk & 64)) JCM_TRY(run_conv(c, "conv6", 1, c5, B, hh, ww, 1, logits, bf, true, planar, 0));         // :72
  return JCM_OK;
}

// spatial_model(heat_map), main.py:94-125.
// The 10-channel input is given as channels [0,Ca) of `hm` ([B,5400,Ca]) plus `extra` ([B,5400,10-Ca]): Ca = 10 for
// jcm_sm_forward, Ca = 9 + the torso map inside the tower (the tf.concat of main.py:528 is never materialised).
int sm_forward_impl(jcm_ctx* c, const float* hm, int Ca, const float* extra, int B, float* logits, int extra_ld = 0) {
  if (extra_ld <= 0) extra_ld = kC - Ca;
  if (!c->has_sm) return fail(JCM_ERR_STATE, "spatial-model parameters (bn_sm, energy_*, bias_*) were not set");
  const int P = c->K * (kC - 1);
  if (c->sm_algo == 1) {   // direct convolution
    float* lik = arena_alloc<float>(c, (size_t)B * kC * kHmH * 96);
    float* cpre = arena_alloc<float>(c, (size_t)B * P * kCH * kCW);
    if (c->dry) return JCM_OK;
    HIP_TRY(sm_likelihood(hm, Ca, extra, c->bn_sm_scale, c->bn_sm_shift, lik, B, kC, c->stream, extra_ld));
    HIP_TRY(sm_pair_conv(c->sp_energy, lik, c->cond, cpre, B, P, kC, c->stream));
    HIP_TRY(sm_finish(lik, cpre, c->sp_bias, logits, B, c->K, kC, c->stream));
    return JCM_OK;
  }
  if (c->sm_algo == 3) {   // fused: all transforms in LDS, only the 10 likelihood spectra per image leave the CU
    float2* lhat_t = arena_alloc<float2>(c, (size_t)B * kC * kSpec);
    if (c->dry) return JCM_OK;
    void* scr = nullptr;
    unsigned epoch = 0;
    JCM_TRY(sm_scratch_next(c, &scr, &epoch));
    HIP_TRY(sm_fused_forward(hm, Ca, extra, extra_ld, c->bn_sm_scale, c->bn_sm_shift, c->prior_spec_t, c->cond, c->sp_bias, lhat_t, logits, B, c->K, kC,
                             c->stream, nullptr, scr, epoch));
    return JCM_OK;
  }
  return fail(JCM_ERR_STATE, "sm_algo must be 3 (transforms in LDS) or 1 (direct)");
}

}  // namespace

extern "C" {

int jcm_abi_version(void) { return 1; }

const char* jcm_last_error(void) { return g_err.c_str(); }

int jcm_create(int device, void* stream, jcm_handle* out) {
  if (!out) return fail(JCM_ERR_ARG, "null out pointer");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(JCM_ERR_ARG, "device " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(JCM_ERR_HIP, std::string("libjcm is built for gfx950 (MI355X) only; device reports ") + prop.gcnArchName);
  jcm_ctx* c = new jcm_ctx();
  c->device = device;
  c->stream = static_cast<hipStream_t>(stream);
  *out = c;
  return JCM_OK;
}

int jcm_destroy(jcm_handle h) {
  if (!h) return JCM_OK;
  DeviceGuard g(h->device);
  (void)hipStreamSynchronize(h->stream);
  prof_release_all(h, true);
  if (h->train) train_destroy(h);
  for (auto& kv : h->params) (void)hipFree(kv.second.d);
  for (void* p : h->owned) (void)hipFree(p);
  for (auto& kv : h->fft_w) (void)hipFree(kv.second.p);
  for (auto& b : h->fft_blocks) (void)hipFree(b.p);
  if (h->arena) (void)hipFree(h->arena);
  delete h;
  return JCM_OK;
}

int jcm_set_option(jcm_handle h, const char* key, int64_t value) {
  JCM_TRY(check(h, false));
  const std::string k = key ? key : "";
  if (k == "profile") {   // allowed at any time; switching it on starts a fresh record (events go back to the pool)
    if (value != 0 && !h->profile) {
      DeviceGuard g(h->device);
      (void)hipStreamSynchronize(h->stream);
      prof_release_all(h, false);
    }
    h->profile = value != 0;
    return JCM_OK;
  }
  if (k == "micro_batch") {   // allowed at any time
    if (value < 0) return fail(JCM_ERR_ARG, "micro_batch must be >= 0 (0 = default: 256 bf16 / 64 fp32)");
    h->micro_batch = (int)value;
    return JCM_OK;
  }
  if (k == "debug_skip") {   // allowed at any time; bisecting aid: groups of launches of jcm_pd_forward that are left out (results are then garbage)
    h->debug_skip = (int)value;
    return JCM_OK;
  }
  if (k == "call_order") {   // allowed at any time; 0 = debugging: this handle's calls are not ordered against other handles' on the device
    h->call_order = value != 0;
    return JCM_OK;
  }
  if (k == "conv9_fft") {  // allowed at any time
    h->conv9_fft = value != 0;
    return JCM_OK;
  }
  if (k == "fft_single") {   // allowed at any time (bf16 handles); the filter spectra have another form: the cache is dropped
    if ((value != 0) != (h->fft_single != 0)) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      for (auto& kv : h->fft_w) (void)hipFree(kv.second.p);
      h->fft_w.clear();
    }
    h->fft_single = value != 0;
    return JCM_OK;
  }
  if (k == "fft_windows") {   // allowed at any time (fp32 handles with training state)
    h->fft_win = value != 0;
    return JCM_OK;
  }
  if (k == "fft_fuse") {   // allowed at any time (fp32 handles): bit 0 = conv2 -> pool -> conv3, bit 1 = conv4_fullres -> merge -> conv5 as fused hand-overs
    if (value < 0 || value > 3) return fail(JCM_ERR_ARG, "fft_fuse must be 0..3 (bit 0: pool hand-over, bit 1: merge hand-over)");
    h->fft_fuse = (int)value;
    return JCM_OK;
  }
  if (k == "bf16_hpool") {   // allowed at any time (bf16 handles)
    h->bf16_hpool = value != 0;
    return JCM_OK;
  }
  if (k == "fft_rows_mfma") {   // allowed at any time (bf16 handles with 16-bit row-transformed tensors)
    h->fft_rows_mfma = value != 0;
    return JCM_OK;
  }
  if (k == "fft_t16") {   // allowed at any time (bf16 handles, fft_single = 1)
    h->fft_t16 = value != 0;
    return JCM_OK;
  }
  if (k == "sm_chunk") {  // allowed at any time
    if (value < 1) return fail(JCM_ERR_ARG, "sm_chunk must be >= 1");
    h->sm_chunk = (int)value;
    return JCM_OK;
  }
  if (k == "split_min_wgs") {   // allowed at any time
    if (value < 0) return fail(JCM_ERR_ARG, "split_min_wgs must be >= 0");
    h->split_min_wgs = (int)value;
    return JCM_OK;
  }
  if (k == "sm_algo") {   // allowed at any time
    if (value != 1 && value != 3) return fail(JCM_ERR_ARG, "sm_algo must be 3 (every transform in LDS, default) or 1 (direct sliding-window kernel); the rocFFT routes 0 and 2 were removed in round 5");
    h->sm_algo = (int)value;
    return JCM_OK;
  }
  if (h->finalized) return fail(JCM_ERR_STATE, "options must be set before jcm_finalize");
  if (k == "precision") {
    if (value != JCM_PRECISION_F32 && value != JCM_PRECISION_BF16) return fail(JCM_ERR_ARG, "precision must be 0 (f32) or 1 (bf16)");
    h->precision = (int)value;
  } else if (k == "f32_conv") {
    if (value != 0 && value != 2) return fail(JCM_ERR_ARG, "f32_conv must be 0 (default) or 2 (fp16x3 direct split kernels); 1 (bf16x6) was retired in round 5");
    h->f32_conv = (int)value;
  } else if (k == "n_joints") {
    if (value < 1 || value > 9) return fail(JCM_ERR_ARG, "n_joints must be in [1,9]");
    h->K = (int)value;
  } else {
    return fail(JCM_ERR_ARG, "unknown option '" + k + "'");
  }
  return JCM_OK;
}

int jcm_set_tensor(jcm_handle h, const char* name, const float* data, const int64_t* shape, int ndim) {
  JCM_TRY(check(h, false));
  if (h->finalized) return fail(JCM_ERR_STATE, "parameters must be set before jcm_finalize");
  if (!name || !data || !shape || ndim < 1 || ndim > 4) return fail(JCM_ERR_ARG, "bad set_tensor arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return fail(JCM_ERR_ARG, std::string("non-positive dimension in '") + name + "'");
    n *= (size_t)shape[i];
  }
  Tensor& t = h->params[name];
  if (t.d) { (void)hipFree(t.d); h->param_bytes -= t.n * sizeof(float); }
  t.shape.assign(shape, shape + ndim);
  t.n = n;
  t.d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d), n * sizeof(float)));
  h->param_bytes += n * sizeof(float);
  HIP_TRY(hipMemcpyAsync(t.d, data, n * sizeof(float), hipMemcpyDefault, h->stream));
  order.release();
  HIP_TRY(hipStreamSynchronize(h->stream));   // the caller may free `data` on return
  return JCM_OK;
}

int jcm_finalize(jcm_handle h) {
  JCM_TRY(check(h, false));
  if (h->finalized) return fail(JCM_ERR_STATE, "already finalized");
  DeviceGuard g(h->device);
  CallOrder order(h);
  JCM_TRY(refresh_derived(h, true));
  h->finalized = true;
  return JCM_OK;
}

int jcm_conv_layer(jcm_handle h, const char* scope, int stride, int last_layer, const float* x, int B, int H, int W, float* out) {
  JCM_TRY(check(h, true));
  if (!scope || !x || !out || B < 1 || H < 1 || W < 1) return fail(JCM_ERR_ARG, "bad conv_layer arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  const ConvLayer* L = conv_of(h, scope);
  if (!L) return fail(JCM_ERR_STATE, std::string("no conv layer '") + scope + "'");
  if ((last_layer != 0) == L->has_bn)
    return fail(JCM_ERR_ARG, std::string("last_layer flag disagrees with the BatchNorm parameters stored for '") + scope + "'");
  if (h->precision == JCM_PRECISION_F32)      // (the frequency-domain route of the wide 9x9 layers takes its scratch from the arena)
    return with_arena(h, [&] { return run_conv(h, scope, stride, x, B, H, W, 1, out, false, false); });
  // bf16 handle: the boundary stays fp32 NHWC; the layer runs exactly as inside the tower -- input rounded to bf16 (the
  // activation type of that path), bf16 MFMA kernel, bf16 result (fp32 for the logits layer) -- and is widened back.
  if (stride != 1) return fail(JCM_ERR_ARG, "bf16 handles run the stride-2 first layer fused with its pool inside jcm_pd_forward only");
  jcm_ctx* c = h;
  return with_arena(c, [&] {
    const size_t nin = (size_t)B * H * W * L->cin, nout = (size_t)B * H * W * L->cout;
    void* xb = arena_alloc<char>(c, nin * 2);
    void* ob = last_layer ? nullptr : static_cast<void*>(arena_alloc<char>(c, nout * 2));
    if (!c->dry) HIP_TRY(cast_pad_bf16(x, L->cin, xb, L->cin, (size_t)B * H * W, c->stream));
    JCM_TRY(run_conv(c, scope, 1, xb, B, H, W, 1, last_layer ? static_cast<void*>(out) : ob, true, last_layer != 0));   // (sizes its own scratch in the dry pass)
    if (!c->dry && !last_layer) HIP_TRY(cast_bf16_f32(ob, out, nout, c->stream));
    return (int)JCM_OK;
  });
}

// conv_layer(((x1 + up(x2)) + up(x3)) / 3) (main.py:58,67,69-71) exactly as the tower runs it: on the frequency-domain route the merge is formed by the
// layer's forward row pass (rows_fwd_merge*), otherwise by the merge kernel in front of the layer.
int jcm_conv_layer_merged(jcm_handle h, const char* scope, const float* x1, const float* x2, int H2, int W2, const float* x3, int H3, int W3, int B, int H, int W,
                          float* out) {
  JCM_TRY(check(h, true));
  if (!scope || !x1 || !x2 || !x3 || !out || B < 1 || H < 1 || W < 1 || H2 < 1 || W2 < 1 || H3 < 1 || W3 < 1) return fail(JCM_ERR_ARG, "bad conv_layer_merged arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  jcm_ctx* c = h;
  const ConvLayer* L = conv_of(c, scope);
  if (!L) return fail(JCM_ERR_STATE, std::string("no conv layer '") + scope + "'");
  if (!L->has_bn) return fail(JCM_ERR_ARG, "conv_layer_merged: a layer with BatchNorm parameters is expected (conv5)");
  const bool bf = c->precision != JCM_PRECISION_F32;
  if (bf && L->cin % 8) return fail(JCM_ERR_ARG, "conv_layer_merged: Cin % 8 != 0 on a bf16 handle");
  return with_arena(c, [&] {
    const size_t n1 = (size_t)B * H * W * L->cin, n2 = (size_t)B * H2 * W2 * L->cin, n3 = (size_t)B * H3 * W3 * L->cin, nout = (size_t)B * H * W * L->cout;
    const void *a1 = x1, *a2 = x2, *a3 = x3;
    void* ob = out;
    if (bf) {      // the boundary stays fp32 NHWC (jcm_conv_layer): the three maps are rounded to bf16, the result is widened back
      void* b1 = arena_alloc<char>(c, n1 * 2);
      void* b2 = arena_alloc<char>(c, n2 * 2);
      void* b3 = arena_alloc<char>(c, n3 * 2);
      ob = arena_alloc<char>(c, nout * 2);
      if (!c->dry) {
        HIP_TRY(cast_pad_bf16(x1, L->cin, b1, L->cin, (size_t)B * H * W, c->stream));
        HIP_TRY(cast_pad_bf16(x2, L->cin, b2, L->cin, (size_t)B * H2 * W2, c->stream));
        HIP_TRY(cast_pad_bf16(x3, L->cin, b3, L->cin, (size_t)B * H3 * W3, c->stream));
      }
      a1 = b1; a2 = b2; a3 = b3;
    }
    FftMerge mg{a2, H2, W2, a3, H3, W3};
    const void* in = a1;
    if (takes_fft(c, L, B, H, W)) {
      c->fft_merge = &mg;
    } else {
      void* merged = arena_alloc<char>(c, n1 * (bf ? 2 : 4));
      if (!c->dry) HIP_TRY(upsample_merge3(a1, a2, H2, W2, a3, H3, W3, merged, bf, B, H, W, L->cin, c->stream));
      in = merged;
    }
    JCM_TRY(run_conv(c, scope, 1, in, B, H, W, 1, ob, bf, false));
    if (bf && !c->dry) HIP_TRY(cast_bf16_f32(ob, out, nout, c->stream));
    return (int)JCM_OK;
  });
}

int jcm_max_pool(jcm_handle h, const float* x, int B, int H, int W, int C, float* out) {
  JCM_TRY(check(h, false));
  if (!x || !out || B < 1 || H < 1 || W < 1 || C < 1 || C % 4) return fail(JCM_ERR_ARG, "bad max_pool arguments (C must be a multiple of 4)");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(max_pool_2x2(x, out, false, B, H, W, C, h->stream));
  return JCM_OK;
}

int jcm_resize_bilinear(jcm_handle h, const float* x, int B, int H, int W, int C, int OH, int OW, float* out) {
  JCM_TRY(check(h, false));
  if (!x || !out || B < 1 || H < 1 || W < 1 || C < 1 || OH < 1 || OW < 1) return fail(JCM_ERR_ARG, "bad resize arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(resize_bilinear(x, out, B, H, W, C, OH, OW, h->stream));
  return JCM_OK;
}

int jcm_pd_forward(jcm_handle h, const float* x, int B, int H, int W, float* logits_out) {
  JCM_TRY(check(h, true));
  if (!x || !logits_out || B < 1 || H < 8 || W < 8) return fail(JCM_ERR_ARG, "bad pd_forward arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  return with_arena(h, [&] { return pd_forward_impl(h, x, B, H, W, logits_out); });
}

int jcm_spatial_softmax(jcm_handle h, const float* in, int B, int HW, int K, float* out) {
  JCM_TRY(check(h, false));
  if (!in || !out || B < 1 || HW < 1 || K < 1) return fail(JCM_ERR_ARG, "bad spatial_softmax arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(softmax_argmax(in, out, nullptr, B, HW, 1, K, h->stream));   // one-pass kernel for K = 9 maps, general kernel otherwise
  return JCM_OK;
}

int jcm_conv_mrf(jcm_handle h, const float* A, const float* Bmaps, int B, float* out) {
  JCM_TRY(check(h, true));
  if (!A || !Bmaps || !out || B < 1) return fail(JCM_ERR_ARG, "bad conv_mrf arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  jcm_ctx* c = h;
  return with_arena(c, [&] {
    if (c->sm_algo == 1) {
      float* rev = arena_alloc<float>(c, (size_t)B * kHmH * 96);
      float* cpre = arena_alloc<float>(c, (size_t)B * kCH * kCW);
      if (c->dry) return (int)JCM_OK;
      HIP_TRY(sm_likelihood(Bmaps, 1, nullptr, nullptr, nullptr, rev, B, 1, c->stream));     // reversed, padded copy
      HIP_TRY(sm_pair_conv(A, rev, c->cond0, cpre, B, 1, 1, c->stream));         // main.py:83-87
      HIP_TRY(sm_resize_only(cpre, out, B, c->stream));                          // main.py:89
      return (int)JCM_OK;
    }
    // transforms in LDS (sm_fused.hip / sm_lds.hip), spectra transposed [91][120]: the map in the top-left corner of a zero 120x180 frame, the
    // product, and rows 59..119 of the circular convolution, whose window [59.., 89..] is the VALID true convolution
    float2* lhat = arena_alloc<float2>(c, (size_t)B * kSpec);
    float2* ahat = arena_alloc<float2>(c, kSpec);
    float2* spec = arena_alloc<float2>(c, (size_t)B * kSpec);
    float* cfull = arena_alloc<float>(c, (size_t)B * kFrame);
    if (c->dry) return (int)JCM_OK;
    HIP_TRY(sm_fused_spectra(Bmaps, 1, nullptr, 0, nullptr, nullptr, lhat, B, 1, c->stream));
    HIP_TRY(sm_lds_fwd_frames(A, ahat, 1, c->stream));
    HIP_TRY(sm_spec_mul(lhat, ahat, c->cond0, spec, B, 1, 1, c->stream));        // main.py:83-87 (1 / (120 * 180) inside)
    HIP_TRY(sm_lds_inv_frames(spec, cfull, B, 59, 61, 1.0f, c->stream));
    HIP_TRY(sm_resize_frame(cfull, out, B, c->stream));                          // main.py:89
    return (int)JCM_OK;
  });
}

int jcm_sm_forward(jcm_handle h, const float* hm10, int B, float* logits_out) {
  JCM_TRY(check(h, true));
  if (!hm10 || !logits_out || B < 1) return fail(JCM_ERR_ARG, "bad sm_forward arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  return with_arena(h, [&] { return sm_forward_impl(h, hm10, kC, nullptr, B, logits_out); });
}

int jcm_softmax_argmax(jcm_handle h, const float* logits, int B, int HH, int WW, int K, float* prob, int32_t* coords) {
  JCM_TRY(check(h, false));
  if (!logits || (!prob && !coords) || B < 1 || HH < 1 || WW < 1 || K < 1) return fail(JCM_ERR_ARG, "bad softmax_argmax arguments");
  if (!prob && !(K == 9 && (HH * WW) % 4 == 0 && HH * WW <= 5632))
    return fail(JCM_ERR_ARG, "softmax_argmax without a probability output exists for K = 9 and H*W % 4 == 0, H*W <= 5632 only");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(softmax_argmax(logits, prob, coords, B, HH * WW, WW, K, h->stream));
  return JCM_OK;
}

int jcm_argmax_coords(jcm_handle h, const float* hm, int B, int HH, int WW, int K, int32_t* coords) {
  JCM_TRY(check(h, false));
  if (!hm || !coords || B < 1 || HH < 1 || WW < 1 || K < 1) return fail(JCM_ERR_ARG, "bad argmax arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(argmax_coords(hm, coords, B, HH * WW, WW, K, h->stream));
  return JCM_OK;
}

// The tower of main.py:522-531, optionally with the two cross-entropy terms of main.py:538-539 in inference mode (what
// eval_error runs, main.py:275-283).  The torso channel is `torso` [B,HW,1], or channel K of y [B,HW,K+1] when y is given.
static int forward_impl(jcm_handle h, const float* x, const float* torso, const float* y, int B, int H, int W, int use_sm,
                        float* pd_prob, float* sm_prob, int32_t* pd_coords, int32_t* sm_coords, float* losses) {
  JCM_TRY(check(h, true));
  if (!x || B < 1 || H < 8 || W < 8) return fail(JCM_ERR_ARG, "bad forward arguments");
  const int tld = y ? h->K + 1 : 1;                       // floats per pixel of the tensor the torso channel lives in
  if (y) torso = y + h->K;
  if (use_sm && !torso) return fail(JCM_ERR_ARG, "use_sm needs the torso heat map (y_in[...,K:], main.py:528)");
  if (use_sm && (cdiv2(cdiv2(cdiv2(H))) != kHmH || cdiv2(cdiv2(cdiv2(W))) != kHmW))
    return fail(JCM_ERR_ARG, "the spatial model is defined for 60x90 heat maps (480x720 images) only");
  DeviceGuard g(h->device);
  CallOrder order(h);
  jcm_ctx* c = h;
  const int K = c->K;
  if (use_sm && K + 1 != kC) return fail(JCM_ERR_ARG, "use_sm requires n_joints == 9 (10-channel spatial model)");
  const int hh = cdiv2(cdiv2(cdiv2(H))), ww = cdiv2(cdiv2(cdiv2(W)));
  // Images are independent in inference (moving-statistics BatchNorm, main.py:406), so a large batch -- a rank's
  // share of BASELINE configs[3]'s 2048 -- is walked in micro-batches: the arena is sized for one micro-batch,
  // every launch sequence is the one a batch of that size gets, and the outputs land in the caller's tensors at
  // the image offset.  The arena is sized once, for the largest micro-batch.
  const int mb_opt = c->micro_batch > 0 ? c->micro_batch : (c->precision == JCM_PRECISION_BF16 ? 256 : 64);
  const int mb = B < mb_opt ? B : mb_opt;
  auto body = [&](int b0, int nb) {
    const size_t n = (size_t)nb * hh * ww * K;
    const size_t o = (size_t)b0 * hh * ww * K;
    float* logits = arena_alloc<float>(c, n);
    float* prob = pd_prob ? pd_prob + o : arena_alloc<float>(c, n);
    const size_t mark = c->arena_off;
    JCM_TRY(pd_forward_impl(c, x + (size_t)b0 * H * W * 3, nb, H, W, logits));            // main.py:522
    c->arena_off = mark;
    // spatial_softmax (main.py:523) and the argmax of evaluation.py:15-24 in one pass over the logits
    if (!c->dry) HIP_TRY(softmax_argmax(logits, prob, pd_coords ? pd_coords + (size_t)b0 * 2 * K : nullptr, nb, hh * ww, ww, K, c->stream));
    float* ce = losses ? arena_alloc<float>(c, 2 * (size_t)nb * K) : nullptr;            // per (image, joint) cross entropies
    const float* yb = y ? y + (size_t)b0 * hh * ww * (K + 1) : nullptr;
    if (losses && !c->dry) HIP_TRY(softmax_ce(logits, yb, nb, hh * ww, K, K + 1, 0.f, ce, nullptr, 0, 0, c->stream));     // main.py:538
    if (use_sm) {
      float* sml = arena_alloc<float>(c, n);
      float* smp = sm_prob ? sm_prob + o : nullptr;
      hipEvent_t s0 = nullptr, s1 = nullptr;
      if (!c->dry) JCM_TRY(prof_begin(c, &s0, &s1));
      const int rs = sm_forward_impl(c, prob, K, torso + (size_t)b0 * hh * ww * tld, nb, sml, tld);       // main.py:528,530
      if (!c->dry) prof_end(c, "sm", s0, s1, rs == JCM_OK);      // jcm_profile_read("sm"): the spatial model's kernels (bench.py roofline.sm)
      JCM_TRY(rs);
      if (!c->dry && (smp || sm_coords))
        HIP_TRY(softmax_argmax(sml, smp, sm_coords ? sm_coords + (size_t)b0 * 2 * K : nullptr, nb, hh * ww, ww, K, c->stream));   // main.py:531
      if (losses && !c->dry) HIP_TRY(softmax_ce(sml, yb, nb, hh * ww, K, K + 1, 0.f, ce + (size_t)nb * K, nullptr, 0, 0, c->stream));   // main.py:539
    } else if (losses && !c->dry) {
      HIP_TRY(hipMemcpyAsync(ce + (size_t)nb * K, ce, (size_t)nb * K * sizeof(float), hipMemcpyDeviceToDevice, c->stream));       // main.py:535
    }
    if (losses && !c->dry) HIP_TRY(loss_means_accumulate(ce, nb * K, 1.0f / (float)(B * K), losses, b0 == 0, c->stream));      // reduce_mean, main.py:240
    return (int)JCM_OK;
  };
  // sizing pass on the largest micro-batch, then the real passes (the arena never reallocates mid-graph)
  c->dry = true;
  c->arena_off = 0;
  c->arena_peak = 0;
  int r = body(0, mb);
  c->dry = false;
  if (r != JCM_OK) return r;
  JCM_TRY(arena_reserve(c, c->arena_peak));
  for (int b0 = 0; b0 < B; b0 += mb) {
    c->arena_off = 0;
    JCM_TRY(body(b0, B - b0 < mb ? B - b0 : mb));
  }
  return JCM_OK;
}

int jcm_forward(jcm_handle h, const float* x, const float* torso, int B, int H, int W, int use_sm,
                float* pd_prob, float* sm_prob, int32_t* pd_coords, int32_t* sm_coords) {
  return forward_impl(h, x, torso, nullptr, B, H, W, use_sm, pd_prob, sm_prob, pd_coords, sm_coords, nullptr);
}

int jcm_eval_forward(jcm_handle h, const float* x, const float* y, int B, int H, int W, int use_sm,
                     float* pd_prob, float* sm_prob, int32_t* pd_coords, int32_t* sm_coords, float* losses) {
  if (!y || !losses) return fail(JCM_ERR_ARG, "eval_forward needs the target heat maps y [B,60,90,K+1] and a 2-float loss buffer");
  return forward_impl(h, x, nullptr, y, B, H, W, use_sm, pd_prob, sm_prob, pd_coords, sm_coords, losses);
}

int jcm_window_resize(jcm_handle h, const float* src, int nsrc, int H, int W, int C, const int32_t* windows, int NW,
                      int OH, int OW, float* out) {
  JCM_TRY(check(h, false));
  if (!src || !windows || !out || nsrc < 1 || H < 1 || W < 1 || C < 1 || NW < 1 || OH < 1 || OW < 1)
    return fail(JCM_ERR_ARG, "bad window_resize arguments");
  for (int i = 0; i < NW; ++i) {
    const int32_t* w = windows + i * 5;
    if (w[0] < 0 || w[0] >= nsrc || w[3] < 1 || w[4] < 1) return fail(JCM_ERR_ARG, "bad window " + std::to_string(i));
  }
  DeviceGuard g(h->device);
  CallOrder order(h);
  jcm_ctx* c = h;
  return with_arena(c, [&] {
    int* wdev = arena_alloc<int>(c, (size_t)NW * 5);
    float2* mm = arena_alloc<float2>(c, NW);
    if (c->dry) return (int)JCM_OK;
    HIP_TRY(hipMemcpyAsync(wdev, windows, (size_t)NW * 5 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(window_resize(src, H, W, C, wdev, NW, mm, OH, OW, out, c->stream));
    order.release();
    HIP_TRY(hipStreamSynchronize(c->stream));   // `windows` is caller-owned host memory
    return (int)JCM_OK;
  });
}

int jcm_group_mean(jcm_handle h, const float* in, int n, int G, int64_t M, float* out) {
  JCM_TRY(check(h, false));
  if (!in || !out || n < 1 || G < 1 || M < 1) return fail(JCM_ERR_ARG, "bad group_mean arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(group_mean(in, out, n, G, (size_t)M, h->stream));
  return JCM_OK;
}

int jcm_profile_read(jcm_handle h, const char* scope, double* total_ms, int* launches) {
  JCM_TRY(check(h, false));
  if (!scope || !total_ms || !launches) return fail(JCM_ERR_ARG, "bad profile_read arguments");
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  double tot = 0;
  int n = 0;
  auto it = h->prof.find(scope);
  if (it != h->prof.end()) {
    for (auto& ev : it->second) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
      tot += ms;
      ++n;
      h->event_pool.push_back(ev.first);
      h->event_pool.push_back(ev.second);
    }
    h->prof.erase(it);
  }
  *total_ms = tot;
  *launches = n;
  return JCM_OK;
}

int jcm_conv_kernel_name(jcm_handle h, const char* scope, int B, int H, int W, char* name, int cap) {
  JCM_TRY(check(h, true));
  if (!scope || !name || cap < 2 || B < 1 || H < 1 || W < 1) return fail(JCM_ERR_ARG, "bad conv_kernel_name arguments");
  const ConvLayer* L = conv_of(h, scope);
  if (!L) return fail(JCM_ERR_STATE, std::string("no conv layer '") + scope + "'");
  ConvArgs a{};
  a.B = B; a.H = H; a.W = W; a.Cin = L->cin; a.Cout = L->cout; a.relu_bn = L->has_bn ? 1 : 0;
  const char* k;
  if (L->cin == 3) {
    k = h->precision == JCM_PRECISION_BF16 ? (L->wq1_bf16 ? "conv1_mfma_pool_kernel" : "conv1_5x5s2_kernel") : (L->wq1_f32 ? (h->conv9_fft && h->f32_conv == 0 && L->wq1_split ? "conv1_mfma_pool_split_kernel" : "conv1_mfma_pool_f32_kernel") : "conv1_5x5s2_kernel");
  } else if (h->precision == JCM_PRECISION_BF16) {
    a.CoutP = L->coutp_bf16;
    a.in_planar = 0;
    k = L->thin_bf16 ? (L->wp_kxfold && conv_kxfold_bf16_supported(a, L->ks) ? "conv_kxfold_bf16_kernel" : "conv_thin_bf16_kernel")
        : (conv_igemm_bf16_bn(L->cout, L->ks) == 256 && conv_strip_bf16_supported(a, L->ks)) ? "conv_strip_bf16_kernel"
        : (L->ks == 5 && conv_igemm_bf16_bn(L->cout, L->ks) == 128 && conv5_strip_bf16_supported(a, L->ks)) ? "conv5_strip_bf16_kernel" : "conv_igemm_bf16_kernel";
    if (takes_fft(h, L, B, H, W)) k = "conv_fft(cgemm_split_kernel)";
  } else {
    const bool use_split = L->wp_split && (L->thin ? h->f32_conv == 2 : conv_split_supported(L->ks, L->cin, L->coutp_split, B, H, W, h->split_min_wgs));
    k = L->thin ? (use_split ? "conv_thin_split16_kernel" : "conv_thin_f32_kernel") : use_split ? "conv_split_kernel" : "conv_igemm_f32_kernel";
    if (takes_fft(h, L, B, H, W)) k = "conv_fft(cgemm_split_kernel)";
  }
  std::snprintf(name, (size_t)cap, "%s", k);
  return JCM_OK;
}

int64_t jcm_workspace_bytes(jcm_handle h) { return h ? (int64_t)(h->arena_cap + h->param_bytes) : 0; }

}  // extern "C"
