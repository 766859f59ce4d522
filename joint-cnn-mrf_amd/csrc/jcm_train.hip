// Joint training step behind include/jcm.h (main.py:511-577): forward in
// training mode (batch-statistics BatchNorm + moving-average update), the two soft-label spatial
// cross-entropies + weight decay, the backward pass of the part detector and the spatial model,
// and tf.train-style clip + Adam / momentum.  fp32 throughout; every convolution (forward, data
// gradient, weight gradient) runs on v_mfma_f32_32x32x2_f32.
#include <cmath>
#include <cstring>

#include <cstdlib>
#include "ctx.h"

using namespace jcm;

namespace jcm {

struct BnSave {
  float* mean = nullptr;   // [C] batch mean
  float* rstd = nullptr;   // [C] 1/sqrt(biased var + eps)
};

struct DgradW {
  float* wd = nullptr;     // packed flipped/transposed weights for conv_igemm_f32
  void* wd_split = nullptr;  // the same in two fp16 parts for conv_split_f32 (handles with f32_conv = 2)
  void* wd_bf16 = nullptr;   // bf16 handles: packed for conv_igemm_bf16
  int cinp_bf16 = 0, coutp_bf16 = 0;
  bool stale = true;       // packed before the last weight update
  int cinp = 0;            // dZ channel stride the kernel reads (= Cout rounded up to 16)
  int coutp = 0;           // packed N extent (= Cin rounded up to the kernel's N tile)
};

struct Slot {
  std::string name;
  float* w;
  size_t n, off;
};

struct TrainState {
  std::vector<Slot> slots;             // trainable tensors, sorted by name
  std::map<std::string, size_t> index; // name -> slot
  size_t total = 0;
  float* opt_m = nullptr;              // Adam m / momentum accumulator, flat [total]
  float* opt_v = nullptr;              // Adam v, flat [total]
  float* ones = nullptr;               // [maxC] identity epilogue scale
  float* zeros = nullptr;              // [maxC]
  std::map<std::string, DgradW> dgrad;
  std::map<std::string, BnSave> bn;
  float* scratch_flip = nullptr;       // largest flipped HWIO weight
  void* zs = nullptr;                  // split spectra of the dz the last conv_wgrad saw (frequency-domain route), for the conv_dgrad that follows
  const void* zs_of = nullptr;
  int zs_cin = 0;
  float* zs_tmax = nullptr;            // ... and the word of their fp16 scaling
  double* red = nullptr;               // per-channel reduction scratch
  double* sumsq = nullptr;             // [2]: grad sum of squares, weight sum of squares (l2)
  float* small = nullptr;              // [2*maxC + 64] misc
  // spatial model: per-pair parameter pointers / flat-gradient offsets, graph order
  const float** e_ptr = nullptr;
  const float** b_ptr = nullptr;
  int64_t* e_off = nullptr;
  int64_t* b_off = nullptr;
  // optimizer chunk table (one launch updates every tensor)
  float* gscale = nullptr;             // [2] {S, 1/S}: power-of-two scale of the current layer's gradient (f32_conv = 2)
  float* gscratch = nullptr;           // [1024]
  const void* gscale_of = nullptr;     // the tensor gscale currently describes
  // "these gradients are final" notifications (jcm_train_set_grad_callback): name prefix -> [offset, count) of the flat buffer
  jcm_grad_ready_fn ready_fn = nullptr;
  void* ready_user = nullptr;
  std::map<std::string, std::pair<int64_t, int64_t>> ranges;
  float** ck_w = nullptr;
  int64_t* ck_start = nullptr;
  int64_t* ck_off = nullptr;
  int* ck_len = nullptr;
  int* ck_isw = nullptr;               // the chunk belongs to a '<scope>/weights' tensor (weight decay, main.py:195-205)
  int n_chunks = 0;
  int maxC = 0;
  long step = 0;                       // optimizer updates applied (n_iters, main.py:491)
};

}  // namespace jcm

namespace {

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
bool trainable(const std::string& name) { return !ends_with(name, "moving_mean") && !ends_with(name, "moving_variance"); }

// The weights changed: the packed data-gradient filters are stale.  They are repacked where a layer's data gradient next runs on the direct
// kernels (conv_dgrad) -- layers on the frequency-domain route never read them.
int repack_dgrad(jcm_ctx* c) {
  for (auto& kv : c->train->dgrad) kv.second.stale = true;
  return JCM_OK;
}
int ensure_dgrad_packed(jcm_ctx* c, const std::string& scope) {
  TrainState* t = c->train;
  {
    const ConvLayer* L = conv_of(c, scope);
    DgradW& d = t->dgrad[scope];
    if (!d.stale) return JCM_OK;
    d.stale = false;
    if (d.wd_bf16) {
      HIP_TRY(flip_transpose_weights(L->w_raw, t->scratch_flip, L->ks, L->cin, L->cout, d.cinp_bf16, c->stream));
      HIP_TRY(pack_weights_bf16(t->scratch_flip, d.wd_bf16, L->ks, d.cinp_bf16, L->cin, d.coutp_bf16, c->stream));
      return JCM_OK;
    }
    HIP_TRY(flip_transpose_weights(L->w_raw, t->scratch_flip, L->ks, L->cin, L->cout, d.cinp, c->stream));
    HIP_TRY(pack_weights_f32(t->scratch_flip, d.wd, L->ks, d.cinp, L->cin, d.coutp, c->stream));
    // gradients: bf16 parts (full fp32 range) in mode 1; fp16 parts + a per-tensor power-of-two scale in mode 2
    if (d.wd_split) HIP_TRY(pack_weights_split(t->scratch_flip, d.wd_split, L->ks, d.cinp, L->cin, L->cin, 2, c->stream, L->wscale));
  }
  return JCM_OK;
}

// Every write of the gradients with this name prefix has been enqueued on the stream: tell the host, which can start
// reducing that range on another stream (after an event recorded now) while the backward pass goes on.
void notify_ready(jcm_ctx* c, const std::string& prefix) {
  TrainState* t = c->train;
  if (c->dry || !t->ready_fn) return;
  auto it = t->ranges.find(prefix);
  if (it == t->ranges.end() || it->second.second <= 0) return;
  // user code runs WITHOUT the device's call-chain lock (it may call jcm_* entry points, also of this handle: CallOrder::nested)
  if (c->order) c->order->release();
  t->ready_fn(t->ready_user, it->second.first, it->second.second);
  if (c->order) c->order->acquire();
}

float* grad_of(TrainState* t, float* grads, const std::string& name) {
  auto it = t->index.find(name);
  return it == t->index.end() ? nullptr : grads + t->slots[it->second].off;
}

// Activations and their gradients are fp32, or bf16 on a bf16 handle (mixed precision: fp32 master weights,
// statistics, losses, spatial model and optimizer; bf16 tensors between the layers, bf16 MFMA with fp32 accumulate).
inline bool bf(const jcm_ctx* c) { return c->precision == JCM_PRECISION_BF16; }
inline void* act(jcm_ctx* c, size_t elems) { return arena_alloc<char>(c, elems * (bf(c) ? 2 : 4)); }

// ---- one conv layer in training mode: r = relu(conv + b) [or conv + b], batch stats, y = BN(r)
struct LayerFwd {
  std::string scope;
  const ConvLayer* L = nullptr;
  const void* in = nullptr;    // input activation (stride-1 layers) or the fp32 image (conv1)
  int H = 0, W = 0;            // output map
  void* r = nullptr;
  void* y = nullptr;
  void* xs = nullptr;          // fp32 handles, frequency-domain layers: the split spectra of the input, kept for the weight gradient (wgrad_fft.hip)
  float* xs_tmax = nullptr;    // ... and the device word of their fp16 scaling (np = 4)
  int win = 0, TY = 0, TX = 0; // the layer ran on overlap-save windows (kWin x kWin, TY x TX of them per image): xs are the WINDOWS' spectra
};

// ---- overlap-save windows (fp32 handles; ConvArgs::circ).  At 16 images per GPU every pass of a wide layer is bound by filter-sized
// spectra -- F Cin Cout complex numbers written by the packers, read by the forward and the data-gradient GEMM, written and read as the weight gradient's
// per-frequency products: 6 x 6.6 GB per step for conv5 on the 64 x 96 transform of its 60 x 90 map.  Cut into 32 x 32 windows (24 x 24 valid pixels + a halo
// of 4, 3 x 4 windows per map) the same layer has 544 frequencies instead of 3136 and 192 "images" instead of 16: the filter-sized tensors shrink 5.8x, the
// activation-sized ones (which were 3 % of the traffic) grow 2.1x.
constexpr int kWin = 32, kWinValid = kWin - 8;
static bool takes_windows(jcm_ctx* c, const ConvLayer* L, int B, int H, int W, int* TY, int* TX) {
  if (bf(c) || !c->fft_win || !takes_fft(c, L, B, H, W)) return false;
  int NY = 0, NX = 0, MT = 0;
  if (!conv_fft_geometry(H, W, L->ks, B, L->cout, fft_np(c), &NY, &NX, &MT)) return false;
  static const double fr = [] { const char* e = std::getenv("JCM_WIN_FREQ_RATIO"); return e ? std::atof(e) : 1.5; }();      // (round 6: 1.5 -- the 30 x 45 maps too, 936 frequencies against 544: 24.03 -> 23.87 ms; rounds 3-5: 2)
  if (fr * kWin * (kWin / 2 + 1) > NY * (NX / 2 + 1)) return false;      // at least half the frequencies, or the larger activation spectra eat the gain (30 x 45 maps: 936 -> 544)
  // ... and filters wide enough that their spectra dominate: the windows cost a gather, a scatter and 2.1x the transform work per channel (measured at 16 images:
  // with every 60 x 90 layer on windows the step stayed at 36 ms -- 9.7 ms saved on filter-sized tensors, as much spent on activation-sized ones)
  static const long min_cc = [] { const char* e = std::getenv("JCM_WIN_MIN_CC"); return e ? std::atol(e) : 128l * 256; }();      // (round 6: 128 x 256 = conv3_fullres too, now that the windows are gathered and scattered inside the row passes: 24.20 -> 24.03 ms; 64 x 128: 24.87)
  if ((long)L->cin * L->cout < min_cc) return false;
  // ... and a batch small enough: what the windows save (filter-sized traffic, independent of the batch: 9.7 ms per step) is spent again on activation-sized work
  // that grows with it (4.4 ms at 16 images)
  if (B > 32) return false;
  *TY = (H + kWinValid - 1) / kWinValid;
  *TX = (W + kWinValid - 1) / kWinValid;
  ConvArgs a{};
  a.B = B * *TY * *TX; a.H = kWin; a.W = kWin; a.Cin = L->cin; a.Cout = L->cout; a.circ = 1;
  if (!conv_fft_supported(a, L->ks)) return false;
  // the data gradient runs the same windows through the flipped, transposed filter: a layer with Cin = this layer's Cout padded to 64
  ConvArgs d = a;
  d.Cin = (L->cout + 63) / 64 * 64; d.Cout = L->cin;
  return conv_fft_supported(d, L->ks);
}

// the convolution half: r = relu(conv + b) (or conv + b), the input spectra kept for the weight gradient where the layer runs in the frequency domain
int conv_train_fwd_conv(jcm_ctx* c, LayerFwd& f, int stride, const void* x, int B, int Hin, int Win, int sub) {
  TrainState* t = c->train;
  f.L = conv_of(c, f.scope);
  if (!f.L) return fail(JCM_ERR_STATE, "no conv layer '" + f.scope + "'");
  f.in = x;
  f.H = stride == 2 ? cdiv2(Hin / sub) : Hin;
  f.W = stride == 2 ? cdiv2(Win / sub) : Win;
  const size_t N = (size_t)B * f.H * f.W;
  f.r = f.L->has_bn ? act(c, N * f.L->cout) : static_cast<void*>(arena_alloc<float>(c, N * f.L->cout));   // the logits stay fp32
  ConvLayer L = *f.L;
  L.scale = t->ones;       // epilogue = relu(z + b) * 1 + 0
  L.shift = t->zeros;
  if (int TY = 0, TX = 0; stride == 1 && takes_windows(c, &L, B, Hin, Win, &TY, &TX)) {
    // windows: gather -> frequency-domain layer on B TY TX windows (their spectra kept for the weight gradient) -> scatter of the valid regions
    f.win = 1; f.TY = TY; f.TX = TX;
    const int BW = B * TY * TX;
    ConvArgs ax{};
    ax.B = BW; ax.H = kWin; ax.W = kWin; ax.Cin = L.cin; ax.Cout = L.cout; ax.circ = 1;
    f.xs = arena_alloc<char>(c, conv_fft_xs_bytes(ax, L.ks, fft_np(c)));
    const size_t mark = c->arena_off;
    // (round 6: where the forward row pass can cut the windows out of the map itself, the gathered tensor does not exist)
    const bool gw = conv_fft_win_gather_supported(kWin, L.cin);
    float* xw = gw ? nullptr : arena_alloc<float>(c, (size_t)BW * kWin * kWin * L.cin);
    const bool sw = conv_fft_win_scatter_supported(kWin, L.cout);      // ... and where the inverse row pass can store into the map, neither do the valid regions
    float* rw = sw ? static_cast<float*>(f.r) : arena_alloc<float>(c, (size_t)BW * kWinValid * kWinValid * L.cout);
    if (!c->dry && !gw) HIP_TRY(window_gather_f32(static_cast<const float*>(x), xw, B, Hin, Win, L.cin, kWin, TY, TX, 0, c->stream));
    c->fft_win_B = B; c->fft_win_H = Hin; c->fft_win_W = Win; c->fft_win_TY = TY; c->fft_win_TX = TX;
    if (gw) c->fft_win_map = x;
    c->fft_win_scatter = sw;
    c->fft_xs = f.xs;
    JCM_TRY(run_conv_fft(c, &L, f.scope, xw, BW, kWin, kWin, rw, 0, 0, 1));
    if (!c->dry) {
      f.xs_tmax = c->fft_last_tmax;
      if (!sw) HIP_TRY(window_scatter_f32(rw, static_cast<float*>(f.r), B, f.H, f.W, L.cout, kWin, TY, TX, c->stream));
    }
    c->arena_off = mark;      // (later work runs behind the scatter on the stream)
    return JCM_OK;
  }
  if (stride == 1 && !bf(c) && takes_fft(c, &L, B, Hin, Win)) {      // keep the input spectra: the weight gradient is taken in the frequency domain too
    ConvArgs ax{};
    ax.B = B; ax.H = Hin; ax.W = Win; ax.Cin = L.cin; ax.Cout = L.cout;
    f.xs = arena_alloc<char>(c, conv_fft_xs_bytes(ax, L.ks, fft_np(c)));
    c->fft_xs = f.xs;
  }
  JCM_TRY(run_conv_layer(c, &L, f.scope, stride, x, B, Hin, Win, sub, f.r, bf(c), !f.L->has_bn));
  if (f.xs && !c->dry) f.xs_tmax = c->fft_last_tmax;
  return JCM_OK;
}

// pool_out (optional): the 2x2/2 max pool of the layer's output goes there -- taken by the BatchNorm kernel while it writes y where that form exists
int conv_train_fwd(jcm_ctx* c, LayerFwd& f, int stride, const void* x, int B, int Hin, int Win, int sub, void* pool_out = nullptr) {
  TrainState* t = c->train;
  JCM_TRY(conv_train_fwd_conv(c, f, stride, x, B, Hin, Win, sub));
  const size_t N = (size_t)B * f.H * f.W;
  if (!f.L->has_bn) {
    f.y = f.r;
    if (pool_out && !c->dry) HIP_TRY(max_pool_2x2(f.y, pool_out, bf(c), B, f.H, f.W, f.L->cout, c->stream));
    return JCM_OK;
  }
  f.y = act(c, N * f.L->cout);
  if (c->dry) return JCM_OK;
  BnSave& s = t->bn[f.scope];
  Tensor& mm = c->params[f.scope + "/BatchNorm/moving_mean"];
  Tensor& mv = c->params[f.scope + "/BatchNorm/moving_variance"];
  HIP_TRY(bn_batch_stats(f.r, bf(c), N, f.L->cout, kBnEps, 0.9f, s.mean, s.rstd, mm.d, mv.d, t->red, c->stream));   // main.py:129,557
  const float *ga = find(c, f.scope + "/BatchNorm/gamma")->d, *be = find(c, f.scope + "/BatchNorm/beta")->d;
  if (pool_out && bn_apply_pool(f.r, s.mean, s.rstd, ga, be, f.y, pool_out, bf(c), B, f.H, f.W, f.L->cout, c->stream)) return hipGetLastError() == hipSuccess ? JCM_OK : fail(JCM_ERR_HIP, "bn_apply_pool");
  HIP_TRY(bn_apply(f.r, s.mean, s.rstd, ga, be, f.y, bf(c), N, f.L->cout, c->stream));
  if (pool_out) HIP_TRY(max_pool_2x2(f.y, pool_out, bf(c), B, f.H, f.W, f.L->cout, c->stream));
  return JCM_OK;
}

// ---- backward of one BN(relu(conv+b)) layer given dy (scaled by dy_scale): fills the parameter
// gradients, returns dz (arena) for the caller to push through wgrad / dgrad
int conv_train_bwd_pre(jcm_ctx* c, const LayerFwd& f, const void* dy, float dy_scale, int B, float* grads, void** dz_out) {
  TrainState* t = c->train;
  const size_t N = (size_t)B * f.H * f.W;
  const int C = f.L->cout;
  void* dz = act(c, N * C);
  *dz_out = dz;
  if (c->dry) return JCM_OK;
  const BnSave& s = t->bn[f.scope];
  float* sums = t->small;   // [2C] <= 1024 floats
  HIP_TRY(bn_bwd_reduce(dy, dy_scale, f.r, bf(c), s.mean, s.rstd, N, C, sums, grad_of(t, grads, f.scope + "/BatchNorm/gamma"),
                        grad_of(t, grads, f.scope + "/BatchNorm/beta"), t->red, c->stream));
  // (the bias gradient = the column sums of dz: taken while dz is written)
  HIP_TRY(bn_bwd_apply_colsum(dy, dy_scale, f.r, bf(c), s.mean, s.rstd, find(c, f.scope + "/BatchNorm/gamma")->d, sums, N, C, 1, dz, grad_of(t, grads, f.scope + "/biases"), t->red,
                              c->stream));
  return JCM_OK;
}
// ... of a layer whose output y went through the 2x2/2 max pool, given the POOLED gradient dp: the pool's backward pass is formed inside the BatchNorm backward
// kernels where that form exists (bn_bwd_pooled); otherwise max_pool_bwd writes it to dy_tmp ([B, H, W, C]) and the plain kernels run
int conv_train_bwd_pre_pooled(jcm_ctx* c, const LayerFwd& f, const void* dp, void* dy_tmp, int B, float* grads, void** dz_out) {
  TrainState* t = c->train;
  const size_t N = (size_t)B * f.H * f.W;
  const int C = f.L->cout;
  void* dz = act(c, N * C);
  *dz_out = dz;
  if (c->dry) return JCM_OK;
  const BnSave& s = t->bn[f.scope];
  float* sums = t->small;   // [2C] <= 1024 floats
  const float* ga = find(c, f.scope + "/BatchNorm/gamma")->d;
  hipError_t e = hipSuccess;
  if (bn_bwd_pooled(dp, f.y, f.r, bf(c), s.mean, s.rstd, ga, B, f.H, f.W, C, sums, grad_of(t, grads, f.scope + "/BatchNorm/gamma"), grad_of(t, grads, f.scope + "/BatchNorm/beta"), dz,
                    grad_of(t, grads, f.scope + "/biases"), t->red, c->stream, &e)) {
    HIP_TRY(e);
    return JCM_OK;
  }
  HIP_TRY(max_pool_bwd(f.y, dp, dy_tmp, bf(c), B, f.H, f.W, C, c->stream));
  HIP_TRY(bn_bwd_reduce(dy_tmp, 1.0f, f.r, bf(c), s.mean, s.rstd, N, C, sums, grad_of(t, grads, f.scope + "/BatchNorm/gamma"), grad_of(t, grads, f.scope + "/BatchNorm/beta"), t->red,
                        c->stream));
  HIP_TRY(bn_bwd_apply_colsum(dy_tmp, 1.0f, f.r, bf(c), s.mean, s.rstd, ga, sums, N, C, 1, dz, grad_of(t, grads, f.scope + "/biases"), t->red, c->stream));
  return JCM_OK;
}

// dW (+ lmbd*W) of a stride-1 layer: x = layer input [B,H,W,Cin], dz [B,H,W,ldz]
int conv_wgrad(jcm_ctx* c, const LayerFwd& f, const void* dz, int ldz, int B, float lmbd, float* grads) {
  TrainState* t = c->train;
  const ConvLayer* L = f.L;
  t->zs = nullptr;
  if (f.win && f.xs && ldz >= L->cout && ldz % 64 == 0) {
    // windows: the spectra of dz on VALID-ONLY windows (zero halo: every output pixel counts once; in window coordinates the correlation with the forward
    // pass's windows is alias-free for |lag| <= 4), P and the taps on the 32 x 32 transform with B TY TX "images"
    const int BW = B * f.TY * f.TX, np = fft_np(c);
    int NY = 0, NX = 0, MTx = 0, MTz = 0, ny2 = 0, nx2 = 0;
    if (!conv_fft_geometry(kWin, kWin, L->ks, BW, L->cout, np, &NY, &NX, &MTx, 1) || !conv_fft_geometry(kWin, kWin, L->ks, BW, L->cin, np, &ny2, &nx2, &MTz, 1))
      return fail(JCM_ERR_STATE, "window geometry of '" + f.scope + "'");
    const size_t mark = c->arena_off;
    const bool gw = conv_fft_win_gather_supported(kWin, ldz);
    float* zw = gw ? nullptr : arena_alloc<float>(c, (size_t)BW * kWin * kWin * ldz);
    ConvArgs az{};
    az.x = zw; az.B = BW; az.H = kWin; az.W = kWin; az.Cin = ldz; az.Cout = L->cin; az.circ = 1;
    if (gw) { az.win_map = dz; az.win_B = B; az.win_H = f.H; az.win_W = f.W; az.win_TY = f.TY; az.win_TX = f.TX; az.win_valid_only = 1; }
    char* zs = arena_alloc<char>(c, conv_fft_xs_bytes(az, L->ks, np));
    char* work = arena_alloc<char>(c, conv_fft_workspace_bytes(az, L->ks, np));
    char* P = arena_alloc<char>(c, wgrad_fft_scratch_bytes(NY, NX, L->cin, ldz));
    if (!c->dry) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      JCM_TRY(prof_begin(c, &e0, &e1));
      float* ztmax = nullptr;
      if (np == 4) JCM_TRY(fft_new_words(c, BW, &ztmax));
      hipError_t le = gw ? hipSuccess : window_gather_f32(static_cast<const float*>(dz), zw, B, f.H, f.W, ldz, kWin, f.TY, f.TX, 1, c->stream);
      if (le == hipSuccess) le = conv_fft_spectra(az, L->ks, np, work, zs, c->stream, ztmax, 1);
      if (le == hipSuccess)
        le = wgrad_fft(f.xs, zs, P, L->w_raw, lmbd, grad_of(t, grads, f.scope + "/weights"), L->ks, NY, NX, BW, MTx, MTz, L->cin, ldz, L->cout, c->stream,
                       np, f.xs_tmax, ztmax, kWin);
      prof_end(c, "wgrad:" + f.scope, e0, e1, le == hipSuccess);
      if (le != hipSuccess) return fail(JCM_ERR_HIP, "frequency-domain weight gradient (windows) of '" + f.scope + "': " + hipGetErrorString(le));
    }
    c->arena_off = mark;
    notify_ready(c, f.scope + "/");
    return JCM_OK;
  }
  if (!f.win && f.xs && !bf(c) && ldz >= L->cout && ldz % 64 == 0) {
    // frequency domain (wgrad_fft.hip): spectra of dz (kept in t->zs for the data gradient that follows), P[f] = conj(X)^T dZ per frequency, k x k taps
    ConvLayer Lz;      // dz as the input of a frequency-domain layer: the same pseudo-layer conv_dgrad runs
    Lz.ks = L->ks; Lz.cin = ldz; Lz.cout = L->cin; Lz.has_bn = false; Lz.w_raw = t->scratch_flip;
    int NY = 0, NX = 0, MTx = 0, MTz = 0, ny2 = 0, nx2 = 0;
    const int np = fft_np(c);
    if (takes_fft(c, &Lz, B, f.H, f.W) && conv_fft_geometry(f.H, f.W, L->ks, B, L->cout, np, &NY, &NX, &MTx) &&
        conv_fft_geometry(f.H, f.W, L->ks, B, L->cin, np, &ny2, &nx2, &MTz)) {
      ConvArgs az{};
      az.x = dz; az.B = B; az.H = f.H; az.W = f.W; az.Cin = ldz; az.Cout = L->cin;
      char* zs = arena_alloc<char>(c, conv_fft_xs_bytes(az, L->ks, np));      // stays allocated: conv_dgrad of this layer reads it
      const size_t mark = c->arena_off;
      char* work = arena_alloc<char>(c, conv_fft_workspace_bytes(az, L->ks, np));
      char* P = arena_alloc<char>(c, wgrad_fft_scratch_bytes(NY, NX, L->cin, ldz));
      if (!c->dry) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        JCM_TRY(prof_begin(c, &e0, &e1));
        float* ztmax = nullptr;
        if (np == 4) JCM_TRY(fft_new_words(c, B, &ztmax));
        hipError_t le = conv_fft_spectra(az, L->ks, np, work, zs, c->stream, ztmax, 1);
        if (le == hipSuccess)
          le = wgrad_fft(f.xs, zs, P, L->w_raw, lmbd, grad_of(t, grads, f.scope + "/weights"), L->ks, NY, NX, B, MTx, MTz, L->cin, ldz, L->cout, c->stream,
                         np, f.xs_tmax, ztmax, f.H);
        t->zs_tmax = ztmax;
        prof_end(c, "wgrad:" + f.scope, e0, e1, le == hipSuccess);
        if (le != hipSuccess) return fail(JCM_ERR_HIP, "frequency-domain weight gradient of '" + f.scope + "': " + hipGetErrorString(le));
        t->zs = zs;
        t->zs_of = dz;
        t->zs_cin = ldz;
      }
      c->arena_off = mark;
      notify_ready(c, f.scope + "/");
      return JCM_OK;
    }
  }
  const size_t n = (size_t)L->ks * L->ks * L->cin * L->cout;
  const int splits = wgrad_splits(L->ks, L->cin, L->cout, B, f.H);
  const size_t mark = c->arena_off;
  float* partial = arena_alloc<float>(c, n * splits);
  // handles with f32_conv = 2: the fp16x3 split kernel on pre-split operands (wgrad_split.hip); dz is lifted into the fp16 range by its own power-of-two scale
  const bool h16 = !bf(c) && c->f32_conv == 2 && wgrad_split_supported(L->ks, L->cin, ldz);
  if (bf(c) && !wgrad_split_supported(L->ks, L->cin, ldz)) return fail(JCM_ERR_ARG, "no bf16 weight-gradient kernel for layer '" + f.scope + "'");
  const size_t nx = (size_t)B * f.H * f.W * L->cin, nz = (size_t)B * f.H * f.W * ldz;
  char* xparts = h16 ? arena_alloc<char>(c, nx * 4) : nullptr;
  char* zparts = h16 ? arena_alloc<char>(c, nz * 4) : nullptr;
  if (!c->dry) {
    if (h16) {
      // the scale computed here is reused by conv_dgrad of the same layer (every conv_dgrad follows its layer's conv_wgrad)
      HIP_TRY(pow2_scale_of(static_cast<const float*>(dz), nz, t->gscale, t->gscratch, c->stream));
      t->gscale_of = dz;
      HIP_TRY(split_parts16(static_cast<const float*>(f.in), xparts, nx, nullptr, c->stream));
      HIP_TRY(split_parts16(static_cast<const float*>(dz), zparts, nz, t->gscale, c->stream));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    JCM_TRY(prof_begin(c, &e0, &e1));
    hipError_t le;
    if (bf(c)) le = wgrad_bf16(f.in, dz, partial, splits, L->ks, B, f.H, f.W, L->cin, L->cout, ldz, c->stream);
    else if (h16) le = wgrad_split16(xparts, zparts, partial, splits, L->ks, B, f.H, f.W, L->cin, L->cout, ldz, c->stream);
    else le = wgrad_f32(static_cast<const float*>(f.in), static_cast<const float*>(dz), partial, splits, L->ks, B, f.H, f.W, L->cin, L->cout, ldz, c->stream);
    prof_end(c, "wgrad:" + f.scope, e0, e1, le == hipSuccess);      // read with jcm_profile_read("wgrad:<scope>")
    if (le != hipSuccess) return fail(JCM_ERR_HIP, "weight-gradient launch of '" + f.scope + "': " + hipGetErrorString(le));
    HIP_TRY(wgrad_reduce(partial, splits, n, L->w_raw, lmbd, grad_of(t, grads, f.scope + "/weights"), c->stream, h16 ? t->gscale + 1 : nullptr));
  }
  c->arena_off = mark;
  notify_ready(c, f.scope + "/");      // weights were the layer's last gradient (BatchNorm and bias gradients precede them)
  return JCM_OK;
}

// dX = conv_SAME(dZ, flipped weights): [B,H,W,ldz] -> [B,H,W,Cin]
// ldz_fft: dz's channel stride when it differs from the packed data-gradient weights' (the logits gradient widened to 64 channels for the
// frequency-domain route; 0 = d.cinp)
int conv_dgrad(jcm_ctx* c, const LayerFwd& f, const void* dz, int B, void* dx, int ldz_fft = 0) {
  TrainState* t = c->train;
  const DgradW& d = t->dgrad[f.scope];
  if (!bf(c)) {
    const int cin_fft = ldz_fft ? ldz_fft : d.cinp;
    // fp32 handles: the data gradient is a SAME correlation with the flipped, transposed filter -- in the frequency domain like the forward
    // pass (conv_fft.hip); its filter spectra ("dgrad:<scope>") are packed from the flipped weights after every update, on first use.
    ConvLayer Ld;
    Ld.ks = f.L->ks; Ld.cin = cin_fft; Ld.cout = f.L->cin; Ld.has_bn = false;
    Ld.w_raw = t->scratch_flip; Ld.bias = t->zeros; Ld.scale = t->ones; Ld.shift = t->zeros;
    if (f.win) {
      // windows WITH their halo of real gradient pixels -> the flipped, transposed filter's layer on the 32 x 32 transform -> scatter
      const std::string key = "dgrad:" + f.scope;
      const int BW = B * f.TY * f.TX;
      if (!c->dry && !fft_spectra_valid(c, key, kWin, kWin, 1))
        HIP_TRY(flip_transpose_weights(f.L->w_raw, t->scratch_flip, f.L->ks, f.L->cin, f.L->cout, cin_fft, c->stream));
      t->zs = nullptr;
      const size_t mark = c->arena_off;
      const bool gw = conv_fft_win_gather_supported(kWin, cin_fft);
      float* zw = gw ? nullptr : arena_alloc<float>(c, (size_t)BW * kWin * kWin * cin_fft);
      const bool sw = conv_fft_win_scatter_supported(kWin, f.L->cin);
      float* xw = sw ? static_cast<float*>(dx) : arena_alloc<float>(c, (size_t)BW * kWinValid * kWinValid * f.L->cin);
      if (!c->dry && !gw) HIP_TRY(window_gather_f32(static_cast<const float*>(dz), zw, B, f.H, f.W, cin_fft, kWin, f.TY, f.TX, 0, c->stream));
      c->fft_win_B = B; c->fft_win_H = f.H; c->fft_win_W = f.W; c->fft_win_TY = f.TY; c->fft_win_TX = f.TX;
      if (gw) c->fft_win_map = dz;
      c->fft_win_scatter = sw;
      JCM_TRY(run_conv_fft(c, &Ld, key, zw, BW, kWin, kWin, xw, 0, 0, 1));
      if (!c->dry && !sw) HIP_TRY(window_scatter_f32(xw, static_cast<float*>(dx), B, f.H, f.W, f.L->cin, kWin, f.TY, f.TX, c->stream));
      c->arena_off = mark;
      return JCM_OK;
    }
    if (takes_fft(c, &Ld, B, f.H, f.W)) {
      const std::string key = "dgrad:" + f.scope;
      if (!c->dry && !fft_spectra_valid(c, key, f.H, f.W))
        HIP_TRY(flip_transpose_weights(f.L->w_raw, t->scratch_flip, f.L->ks, f.L->cin, f.L->cout, cin_fft, c->stream));
      if (!c->dry && t->zs && t->zs_of == dz && t->zs_cin == cin_fft) { c->fft_xs = t->zs; c->fft_xs_ready = true; c->fft_tmax_in = t->zs_tmax; }      // the spectra of dz are there (conv_wgrad just made them)
      t->zs = nullptr;
      return run_conv_fft(c, &Ld, key, dz, B, f.H, f.W, dx, 0, 0);
    }
    if (ldz_fft) return fail(JCM_ERR_STATE, "data gradient of '" + f.scope + "': widened dz without the frequency-domain route");
  }
  if (c->dry) return JCM_OK;
  JCM_TRY(ensure_dgrad_packed(c, f.scope));
  ConvArgs a;
  if (bf(c)) {      // bf16 gradients through the bf16 forward kernels on flipped weights
    a.x = dz; a.wp = d.wd_bf16; a.bias = t->zeros; a.scale = t->ones; a.shift = t->zeros; a.out = dx;
    a.B = B; a.H = f.H; a.W = f.W; a.Cin = d.cinp_bf16; a.Cout = f.L->cin; a.CoutP = d.coutp_bf16; a.relu_bn = 0;
    HIP_TRY(conv_igemm_bf16(a, f.L->ks, false, c->stream));
    return JCM_OK;
  }
  a.x = dz; a.wp = d.wd; a.bias = t->zeros; a.scale = t->ones; a.shift = t->zeros; a.out = dx;
  a.B = B; a.H = f.H; a.W = f.W; a.Cin = d.cinp; a.Cout = f.L->cin; a.CoutP = d.coutp; a.relu_bn = 0;
  const bool split = d.wd_split && conv_split_supported(f.L->ks, d.cinp, f.L->cin, B, f.H, f.W, c->split_min_wgs);
  const int ns = 2;      // fp16 parts (f32_conv = 2)
  if (split) {
    a.wp = d.wd_split; a.CoutP = f.L->cin;
    {
      if (t->gscale_of != dz) {                   // normally set by this layer's conv_wgrad just before
        HIP_TRY(pow2_scale_of(static_cast<const float*>(dz), (size_t)B * f.H * f.W * d.cinp, t->gscale, t->gscratch, c->stream));
        t->gscale_of = dz;
      }
      a.in_scale = t->gscale;
      a.w_scale = f.L->wscale;
      t->gscale_of = nullptr;                     // consumed: arena addresses are reused by later tensors
    }
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  JCM_TRY(prof_begin(c, &e0, &e1));
  const hipError_t le = split ? conv_split_f32(a, f.L->ks, ns, c->stream) : conv_igemm_f32(a, f.L->ks, c->stream);
  prof_end(c, "dgrad:" + f.scope, e0, e1, le == hipSuccess);
  if (le != hipSuccess) return fail(JCM_ERR_HIP, "data-gradient launch of '" + f.scope + "': " + hipGetErrorString(le));
  return JCM_OK;
}

__global__ void finish_losses_kernel(const float* __restrict__ ce_pd, const float* __restrict__ ce_sm, int n, const double* __restrict__ wsq,
                                     float lmbd, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < n; ++i) { a += ce_pd[i]; b += ce_sm[i]; }
  a /= n; b /= n;
  const double l2 = 0.5 * *wsq;
  out[0] = (float)(a + b + (double)lmbd * l2);   // loss_tower (main.py:540)
  out[1] = (float)a;                             // loss_pd
  out[2] = (float)b;                             // loss_sm
  out[3] = (float)l2;                            // weight_decay('weights')
}

int sm_train_impl(jcm_ctx* c, const float* pd_prob, const float* y, int B, float gscale, float* ce_sm, float* dlogits, float* grads);

// forward + backward of one tower (main.py:522-541,559-560)
int loss_grads_impl(jcm_ctx* c, const float* x, const float* y, int B, int H, int W, int use_sm, float lmbd, float* grads, float* losses) {
  TrainState* t = c->train;
  static const char* const kRes[3] = {"fullres", "halfres", "quarterres"};
  const int K = c->K;
  t->gscale_of = nullptr;
  LayerFwd l1[3], l2[3], l3[3], l4[3], l5, l6;
  void *p1[3], *p2[3];
  const bool b16 = bf(c);
  for (int r = 0; r < 3; ++r) {
    const int sub = 1 << r;
    if (H % sub || W % sub) return fail(JCM_ERR_ARG, "training needs image sizes divisible by 4");
    const std::string res = kRes[r];
    l1[r].scope = "conv1_" + res; l2[r].scope = "conv2_" + res; l3[r].scope = "conv3_" + res; l4[r].scope = "conv4_" + res;
    // (the pools behind conv1 and conv2 are taken by the layers' BatchNorm kernels: conv_train_fwd's pool_out)
    const ConvLayer* L1c = conv_of(c, l1[r].scope);
    const ConvLayer* L2c = conv_of(c, l2[r].scope);
    if (!L1c || !L2c) return fail(JCM_ERR_STATE, "part-detector parameters incomplete (" + res + ")");
    const int C1 = L1c->cout, C2 = L2c->cout;
    const int hc1 = cdiv2(H / sub), wc1 = cdiv2(W / sub);      // conv1's output map (stride 2, SAME)
    const int h2 = cdiv2(hc1), w2 = cdiv2(wc1);
    p1[r] = act(c, (size_t)B * h2 * w2 * C1);
    JCM_TRY(conv_train_fwd(c, l1[r], 2, x, B, H, W, sub, p1[r]));                              // main.py:44-45,52-53,61-62
    if (!c->dry && (l1[r].H != hc1 || l1[r].W != wc1)) return fail(JCM_ERR_STATE, "conv1's map is not the size its pool buffer was made for");
    const int h3 = cdiv2(h2), w3 = cdiv2(w2);
    p2[r] = act(c, (size_t)B * h3 * w3 * C2);
    JCM_TRY(conv_train_fwd(c, l2[r], 1, p1[r], B, h2, w2, 1, p2[r]));                           // :46-47,54-55,63-64
    (void)b16;
    JCM_TRY(conv_train_fwd(c, l3[r], 1, p2[r], B, h3, w3, 1));                                  // :48,56,65
    JCM_TRY(conv_train_fwd(c, l4[r], 1, l3[r].y, B, h3, w3, 1));                                // :49,57,66
  }
  const int hh = l4[0].H, ww = l4[0].W, C4 = l4[0].L->cout;
  const size_t NP = (size_t)B * hh * ww;
  void* merged = act(c, NP * C4);
  if (!c->dry)
    HIP_TRY(upsample_merge3(l4[0].y, l4[1].y, l4[1].H, l4[1].W, l4[2].y, l4[2].H, l4[2].W, merged, b16, B, hh, ww, C4, c->stream));  // :58,67,69-70
  l5.scope = "conv5"; l6.scope = "conv6";
  JCM_TRY(conv_train_fwd(c, l5, 1, merged, B, hh, ww, 1));                                      // :71
  JCM_TRY(conv_train_fwd(c, l6, 1, l5.y, B, hh, ww, 1));                                        // :72 (no ReLU / BN)
  if (l6.L && (l6.L->has_bn || l6.L->cout != K)) return fail(JCM_ERR_STATE, "conv6 must be the K-channel logits layer");
  float* logits = static_cast<float*>(l6.r);

  // ---- losses and the gradient w.r.t. the part-detector logits, kept with a 16-channel stride
  constexpr int LDZ = 16;
  float* dlog = arena_alloc<float>(c, NP * LDZ);
  float* ce_pd = arena_alloc<float>(c, (size_t)B * K);
  float* ce_sm = arena_alloc<float>(c, (size_t)B * K);
  const float gscale = 1.0f / (float)(B * K);                                                  // reduce_mean over (image, joint), main.py:240
  if (!c->dry) {
    HIP_TRY(hipMemsetAsync(dlog, 0, NP * LDZ * sizeof(float), c->stream));
    // use_sm == 0: hm_pred_sm_logit is hm_pred_pd_logit (main.py:535), so the same term counts twice
    HIP_TRY(softmax_ce(logits, y, B, hh * ww, K, K + 1, use_sm ? gscale : 2.0f * gscale, ce_pd, dlog, LDZ, 0, c->stream));   // main.py:538
  }
  if (use_sm) {
    if (hh != kHmH || ww != kHmW || K + 1 != kC) return fail(JCM_ERR_ARG, "the spatial model is defined for 60x90 heat maps and 9 joints");
    float* prob = arena_alloc<float>(c, NP * K);
    if (!c->dry) HIP_TRY(spatial_softmax(logits, prob, B, hh * ww, K, c->stream));             // main.py:523
    JCM_TRY(sm_train_impl(c, prob, y, B, gscale, ce_sm, dlog, grads));                          // main.py:528-531,539 + backward
    notify_ready(c, "bias_");
    notify_ready(c, "bn_sm/");
    notify_ready(c, "energy_");
  } else if (!c->dry) {
    HIP_TRY(hipMemcpyAsync(ce_sm, ce_pd, (size_t)B * K * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  }
  if (!c->dry) {
    // weight_decay('weights') (main.py:195-205): sum of l2_loss over the conv weights
    HIP_TRY(sum_squares_chunks(t->ck_w, t->ck_start, t->ck_len, t->ck_isw, t->n_chunks, t->sumsq + 1, t->red, c->stream));
    hipLaunchKernelGGL(finish_losses_kernel, dim3(1), dim3(64), 0, c->stream, ce_pd, ce_sm, B * K, t->sumsq + 1, lmbd, losses);
    HIP_TRY(hipGetLastError());
  }

  // ---- backward of the part detector
  // conv6: z = conv(y5) + b
  constexpr int LDZB = 32;                      // bf16 kernels read 32-channel chunks
  void* dlogb = b16 ? act(c, NP * LDZB) : nullptr;
  if (!c->dry) {
    HIP_TRY(col_sum(dlog, false, NP, LDZ, t->small, t->red, c->stream));
    HIP_TRY(hipMemcpyAsync(grad_of(t, grads, "conv6/biases"), t->small, K * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (b16) HIP_TRY(cast_pad_bf16(dlog, LDZ, dlogb, LDZB, NP, c->stream));
  }
  const void* dl = b16 ? dlogb : static_cast<const void*>(dlog);
  int ldl = b16 ? LDZB : LDZ, ldl_fft = 0;
  if (l6.xs) {      // frequency-domain route (fp32 handles): its forward transforms take 64-channel blocks
    constexpr int LDZF = 64;
    float* dlogf = arena_alloc<float>(c, NP * LDZF);
    if (!c->dry) HIP_TRY(pad_channels_f32(dlog, LDZ, dlogf, LDZF, NP, c->stream));
    dl = dlogf; ldl = LDZF; ldl_fft = LDZF;
  }
  JCM_TRY(conv_wgrad(c, l6, dl, ldl, B, lmbd, grads));
  void* dy5 = act(c, NP * l5.L->cout);
  JCM_TRY(conv_dgrad(c, l6, dl, B, dy5, ldl_fft));
  void* dz5;
  JCM_TRY(conv_train_bwd_pre(c, l5, dy5, 1.0f, B, grads, &dz5));
  JCM_TRY(conv_wgrad(c, l5, dz5, l5.L->cout, B, lmbd, grads));
  void* dmerged = dy5;                          // dy5 is dead once dz5 exists; same size when C4 == C5
  if (l5.L->cin != l5.L->cout) dmerged = act(c, NP * C4);
  JCM_TRY(conv_dgrad(c, l5, dz5, B, dmerged));
  for (int r = 0; r < 3; ++r) {
    const size_t mark = c->arena_off;
    // merge: x = (x1 + up(x2) + up(x3)) / 3
    const void* dy4 = dmerged;
    float sc = 1.0f / 3.0f;
    if (l4[r].H != hh || l4[r].W != ww) {
      void* d = act(c, (size_t)B * l4[r].H * l4[r].W * C4);
      if (!c->dry) HIP_TRY(resize_bilinear_bwd(dmerged, d, b16, B, l4[r].H, l4[r].W, hh, ww, C4, 1.0f / 3.0f, c->stream));
      dy4 = d;
      sc = 1.0f;
    }
    void *dz4, *dz3, *dz2, *dz1;
    JCM_TRY(conv_train_bwd_pre(c, l4[r], dy4, sc, B, grads, &dz4));
    JCM_TRY(conv_wgrad(c, l4[r], dz4, C4, B, lmbd, grads));
    const size_t n3 = (size_t)B * l3[r].H * l3[r].W;
    void* dy3 = act(c, n3 * l3[r].L->cout);
    JCM_TRY(conv_dgrad(c, l4[r], dz4, B, dy3));
    JCM_TRY(conv_train_bwd_pre(c, l3[r], dy3, 1.0f, B, grads, &dz3));
    JCM_TRY(conv_wgrad(c, l3[r], dz3, l3[r].L->cout, B, lmbd, grads));
    void* dp2 = act(c, n3 * l2[r].L->cout);
    JCM_TRY(conv_dgrad(c, l3[r], dz3, B, dp2));
    const size_t n2 = (size_t)B * l2[r].H * l2[r].W;
    void* dy2 = act(c, n2 * l2[r].L->cout);
    JCM_TRY(conv_train_bwd_pre_pooled(c, l2[r], dp2, dy2, B, grads, &dz2));      // pool2's backward inside the BatchNorm backward kernels
    JCM_TRY(conv_wgrad(c, l2[r], dz2, l2[r].L->cout, B, lmbd, grads));
    void* dp1 = act(c, n2 * l1[r].L->cout);
    JCM_TRY(conv_dgrad(c, l2[r], dz2, B, dp1));
    const size_t n1 = (size_t)B * l1[r].H * l1[r].W;
    const int C1 = l1[r].L->cout;
    void* dy1 = act(c, n1 * C1);
    JCM_TRY(conv_train_bwd_pre_pooled(c, l1[r], dp1, dy1, B, grads, &dz1));
    {
      const size_t n = (size_t)25 * 3 * C1;
      const int nb = wgrad_conv1_blocks();
      float* partial = arena_alloc<float>(c, n * nb);
      if (!c->dry) {
        HIP_TRY(wgrad_conv1(x, dz1, b16, partial, B, H, W, 1 << r, C1, c->stream));
        HIP_TRY(wgrad_reduce_wide(partial, nb, n, l1[r].L->w_raw, lmbd, grad_of(t, grads, l1[r].scope + "/weights"), c->stream));
      }
      notify_ready(c, l1[r].scope + "/");
    }
    c->arena_off = mark;
  }
  return JCM_OK;
}

// spatial model in training mode + its backward (main.py:528-531,539).  pd_prob [B,5400,K]; y [B,5400,K+1];
// adds d loss_sm / d pd_logits into dlog [B,5400,16] and fills the bn_sm / energy / bias gradients.
int sm_train_impl(jcm_ctx* c, const float* pd_prob, const float* y, int B, float gscale, float* ce_sm, float* dlog, float* grads) {
  TrainState* t = c->train;
  const int K = c->K, P = K * (kC - 1);
  const size_t N = (size_t)B * kHmHW;
  float* hm10 = arena_alloc<float>(c, N * kC);
  float* sc = arena_alloc<float>(c, 16);
  float* sh = arena_alloc<float>(c, 16);
  float* frame = arena_alloc<float>(c, (size_t)B * kC * kFrame);      // s_c = sp(bn(h_c)) on zero frames: the backward's d log(s_j + d) term reads it
  float2* lhat = arena_alloc<float2>(c, (size_t)B * kC * kSpec);      // likelihood spectra, transposed [B][C][91][120] (sm_fused.hip), kept for dA
  float* tsave = arena_alloc<float>(c, (size_t)B * P * kHmHW);        // the argument of every pairwise log
  float* sml = arena_alloc<float>(c, N * K);
  float* G = arena_alloc<float>(c, N * K);
  float* dh = arena_alloc<float>(c, N * kC);
  float2* dA_hat = arena_alloc<float2>(c, (size_t)P * kSpec);
  float* dspb = arena_alloc<float>(c, (size_t)P * kHmHW);
  const int Bc = B < c->sm_chunk ? B : c->sm_chunk;
  const size_t mark = c->arena_off;
  BnSave* bs = c->dry ? nullptr : &t->bn["bn_sm"];
  // ---- forward: the fused kernels of the inference path (every transform in LDS), which also leave the log arguments
  if (!c->dry) {
    HIP_TRY(sm_concat_target(pd_prob, y, hm10, N, K, kC, c->stream));                                        // main.py:528
    HIP_TRY(bn_batch_stats(hm10, false, N, kC, kBnEps, 0.9f, bs->mean, bs->rstd, c->params["bn_sm/BatchNorm/moving_mean"].d,
                           c->params["bn_sm/BatchNorm/moving_variance"].d, t->red, c->stream));              // main.py:113
    HIP_TRY(bn_fold_stats(bs->mean, bs->rstd, find(c, "bn_sm/BatchNorm/gamma")->d, find(c, "bn_sm/BatchNorm/beta")->d, sc, sh, kC, c->stream));
    HIP_TRY(sm_pad_frame(hm10, kC, nullptr, sc, sh, frame, B, kC, c->stream));
    void* scr = nullptr;
    unsigned epoch = 0;
    JCM_TRY(sm_scratch_next(c, &scr, &epoch));
    HIP_TRY(sm_fused_forward(hm10, kC, nullptr, 0, sc, sh, c->prior_spec_t, c->cond, c->sp_bias, lhat, sml, B, K, kC, c->stream, tsave, scr, epoch));   // main.py:117-123
    HIP_TRY(softmax_ce(sml, y, B, kHmHW, K, K + 1, gscale, ce_sm, G, K, 0, c->stream));                      // main.py:539
  }
  // ---- backward (sm_train.hip; transforms: sm_lds.hip)
  {
    float2* Dhat = arena_alloc<float2>(c, (size_t)Bc * P * kSpec);
    float2* dLhat = arena_alloc<float2>(c, (size_t)Bc * kC * kSpec);
    float* dLframe = arena_alloc<float>(c, (size_t)Bc * kC * kFrame);      // rows 0..59 are written and read
    float* dAframe = arena_alloc<float>(c, (size_t)P * kFrame);
    float* dhm = arena_alloc<float>(c, N * kC);
    if (!c->dry) {
      for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = B - b0 < Bc ? B - b0 : Bc;
        const float* Gb = G + (size_t)b0 * kHmHW * K;
        const float* Tb = tsave + (size_t)b0 * P * kHmHW;
        HIP_TRY(sm_bwd_dbias(Gb, Tb, dspb, nb, K, P, b0 > 0, c->stream));
        HIP_TRY(sm_lds_fwd_dframes(Gb, Tb, Dhat, nb, K, P, c->stream));      // D_p = R^T (G_j / T_p) on the window, transformed as it is built
        HIP_TRY(sm_bwd_spec_da(Dhat, lhat + (size_t)b0 * kC * kSpec, c->cond, dA_hat, nb, kC, P, b0 > 0, c->stream));
        HIP_TRY(sm_bwd_spec_dl(Dhat, c->prior_spec_t, dLhat, nb, K, kC, c->stream));
        HIP_TRY(sm_lds_inv_frames(dLhat, dLframe, nb * kC, 0, kHmH, 1.0f, c->stream));
        HIP_TRY(sm_bwd_dh(dLframe, Gb, frame + (size_t)b0 * kC * kFrame, hm10 + (size_t)b0 * kHmHW * kC, sc, sh,
                          dh + (size_t)b0 * kHmHW * kC, nb, K, kC, c->stream));
      }
      HIP_TRY(sm_lds_inv_frames(dA_hat, dAframe, P, 0, kPrH, 1.0f, c->stream));
      HIP_TRY(sm_bwd_params(dAframe, dspb, t->e_ptr, t->b_ptr, t->e_off, t->b_off, grads, P, c->stream));
      float* sums = t->small;
      HIP_TRY(bn_bwd_reduce(dh, 1.0f, hm10, false, bs->mean, bs->rstd, N, kC, sums, grad_of(t, grads, "bn_sm/BatchNorm/gamma"),
                            grad_of(t, grads, "bn_sm/BatchNorm/beta"), t->red, c->stream));
      HIP_TRY(bn_bwd_apply(dh, 1.0f, hm10, false, bs->mean, bs->rstd, find(c, "bn_sm/BatchNorm/gamma")->d, sums, N, kC, 0, dhm, c->stream));
      HIP_TRY(softmax_bwd(pd_prob, dhm, B, kHmHW, K, kC, dlog, 16, c->stream));                                // through main.py:523
    }
  }
  c->arena_off = mark;
  return JCM_OK;
}

int need_train(jcm_handle h) {
  JCM_TRY(check(h, true));
  if (!h->train) return fail(JCM_ERR_STATE, "jcm_train_begin has not been called");
  return JCM_OK;
}

}  // namespace

extern "C" {

int jcm_train_begin(jcm_handle h) {
  JCM_TRY(check(h, true));
  if (h->train) return fail(JCM_ERR_STATE, "jcm_train_begin was already called");
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (h->call_depth > 1) return fail(JCM_ERR_STATE, "jcm_train_begin changes the handle's training state or parameters and cannot be called from the gradient-ready callback of the same handle");
  jcm_ctx* c = h;
  TrainState* t = new TrainState();
  c->train = t;
  for (auto& kv : c->params) {       // std::map: sorted by name
    if (!trainable(kv.first)) continue;
    t->index[kv.first] = t->slots.size();
    t->slots.push_back(Slot{kv.first, kv.second.d, kv.second.n, t->total});
    t->total += kv.second.n;
  }
  for (const Slot& sl : t->slots) {      // contiguous name-prefix ranges: "<scope>/" per layer, "bias_", "bn_sm/", "energy_"
    std::string pre;
    if (sl.name.compare(0, 5, "bias_") == 0) pre = "bias_";
    else if (sl.name.compare(0, 7, "energy_") == 0) pre = "energy_";
    else pre = sl.name.substr(0, sl.name.find('/') + 1);
    auto it = t->ranges.find(pre);
    if (it == t->ranges.end()) t->ranges[pre] = {(int64_t)sl.off, (int64_t)sl.n};
    else if (it->second.first + it->second.second == (int64_t)sl.off) it->second.second += (int64_t)sl.n;
    else return fail(JCM_ERR_STATE, "gradient range of '" + pre + "' is not contiguous");
  }
  size_t max_w = 0;
  for (auto& kv : c->convs) {
    const ConvLayer& L = kv.second;
    if (L.cout > t->maxC) t->maxC = L.cout;
    if (L.cin > t->maxC) t->maxC = L.cin;
    if (L.cin == 3) continue;        // conv1: no data gradient (the image is the input)
    if (L.cin % 16 || !(L.ks == 5 || L.ks == 9)) return fail(JCM_ERR_ARG, "no training kernels for layer '" + kv.first + "'");
    DgradW d;
    d.cinp = (L.cout + 15) / 16 * 16;
    const int bn = conv_igemm_bn(L.cin);
    d.coutp = (L.cin + bn - 1) / bn * bn;
    if (c->precision == JCM_PRECISION_BF16) {
      if (L.cin % 32) return fail(JCM_ERR_ARG, "bf16 training needs Cin % 32 == 0 ('" + kv.first + "')");
      d.cinp_bf16 = (L.cout + 31) / 32 * 32;
      const int bnb = conv_igemm_bf16_bn(L.cin, L.ks);
      d.coutp_bf16 = (L.cin + bnb - 1) / bnb * bnb;
      JCM_TRY(dev_alloc(c, &d.wd_bf16, (size_t)L.ks * L.ks * d.cinp_bf16 * d.coutp_bf16 * 2));
    } else {
      JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&d.wd), (size_t)L.ks * L.ks * d.cinp * d.coutp * sizeof(float)));
      if (c->f32_conv == 2 && L.cin % 128 == 0)      // data gradient on the fp16x3 split kernel where its tile fits
        JCM_TRY(dev_alloc(c, &d.wd_split, conv_split_weight_bytes(L.ks, d.cinp, L.cin, 2)));
    }
    const size_t nf = (size_t)L.ks * L.ks * (d.cinp_bf16 > d.cinp ? d.cinp_bf16 : d.cinp) * L.cin;
    if (nf > max_w) max_w = nf;
    t->dgrad[kv.first] = d;
  }
  if (t->maxC < 16) t->maxC = 16;
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->scratch_flip), max_w * sizeof(float)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->opt_m), t->total * sizeof(float)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->opt_v), t->total * sizeof(float)));
  HIP_TRY(hipMemsetAsync(t->opt_m, 0, t->total * sizeof(float), c->stream));
  HIP_TRY(hipMemsetAsync(t->opt_v, 0, t->total * sizeof(float), c->stream));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ones), t->maxC * sizeof(float)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->zeros), t->maxC * sizeof(float)));
  {
    std::vector<float> one(t->maxC, 1.0f);
    HIP_TRY(hipMemcpyAsync(t->ones, one.data(), t->maxC * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(t->zeros, 0, t->maxC * sizeof(float), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  for (auto& kv : c->convs) {
    if (!kv.second.has_bn) continue;
    BnSave s;
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&s.mean), kv.second.cout * sizeof(float)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&s.rstd), kv.second.cout * sizeof(float)));
    t->bn[kv.first] = s;
  }
  if (c->has_sm) {
    BnSave s;
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&s.mean), kC * sizeof(float)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&s.rstd), kC * sizeof(float)));
    t->bn["bn_sm"] = s;
  }
  if (c->has_sm) {
    const int P = c->K * (kC - 1);
    std::vector<const float*> ep(P), bp(P);
    std::vector<int64_t> eo(P), bo(P);
    int p = 0;
    for (int j = 0; j < c->K; ++j)
      for (int cc = 0; cc < kC; ++cc) {
        if (cc == j) continue;
        const std::string key = std::string(kJointNames[j]) + "_" + kJointNames[cc];
        ep[p] = find(c, "energy_" + key)->d;
        bp[p] = find(c, "bias_" + key)->d;
        eo[p] = (int64_t)t->slots[t->index["energy_" + key]].off;
        bo[p] = (int64_t)t->slots[t->index["bias_" + key]].off;
        ++p;
      }
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->e_ptr), P * sizeof(float*)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->b_ptr), P * sizeof(float*)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->e_off), P * sizeof(int64_t)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->b_off), P * sizeof(int64_t)));
    HIP_TRY(hipMemcpyAsync(t->e_ptr, ep.data(), P * sizeof(float*), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->b_ptr, bp.data(), P * sizeof(float*), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->e_off, eo.data(), P * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->b_off, bo.data(), P * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  {
    constexpr int64_t kChunk = 16384;
    std::vector<float*> cw;
    std::vector<int64_t> cs, co;
    std::vector<int> cl, cf;
    for (const Slot& sl : t->slots)
      for (int64_t st0 = 0; st0 < (int64_t)sl.n; st0 += kChunk) {
        cw.push_back(sl.w); cs.push_back(st0); co.push_back((int64_t)sl.off);
        cf.push_back(sl.name.find("weights") != std::string::npos ? 1 : 0);
        cl.push_back((int)((int64_t)sl.n - st0 < kChunk ? (int64_t)sl.n - st0 : kChunk));
      }
    t->n_chunks = (int)cw.size();
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ck_w), cw.size() * sizeof(float*)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ck_start), cs.size() * sizeof(int64_t)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ck_off), co.size() * sizeof(int64_t)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ck_len), cl.size() * sizeof(int)));
    JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->ck_isw), cf.size() * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(t->ck_w, cw.data(), cw.size() * sizeof(float*), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->ck_start, cs.data(), cs.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->ck_off, co.data(), co.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->ck_len, cl.data(), cl.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->ck_isw, cf.data(), cf.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->gscale), 2 * sizeof(float)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->gscratch), 1024 * sizeof(float)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->red), train_reduce_scratch_doubles(t->maxC) * sizeof(double)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->sumsq), 2 * sizeof(double)));
  JCM_TRY(dev_alloc(c, reinterpret_cast<void**>(&t->small), (size_t)(2 * t->maxC + 64) * sizeof(float)));
  JCM_TRY(repack_dgrad(c));
  return JCM_OK;
}

int jcm_train_param_count(jcm_handle h, int64_t* n_tensors, int64_t* n_elements) {
  JCM_TRY(need_train(h));
  if (n_tensors) *n_tensors = (int64_t)h->train->slots.size();
  if (n_elements) *n_elements = (int64_t)h->train->total;
  return JCM_OK;
}

int jcm_train_param_info(jcm_handle h, int64_t index, char* name, int name_cap, int64_t* offset, int64_t* count) {
  JCM_TRY(need_train(h));
  if (index < 0 || index >= (int64_t)h->train->slots.size()) return fail(JCM_ERR_ARG, "parameter index out of range");
  const Slot& s = h->train->slots[(size_t)index];
  if (name) {
    if ((int)s.name.size() + 1 > name_cap) return fail(JCM_ERR_ARG, "name buffer too small");
    std::memcpy(name, s.name.c_str(), s.name.size() + 1);
  }
  if (offset) *offset = (int64_t)s.off;
  if (count) *count = (int64_t)s.n;
  return JCM_OK;
}

int jcm_train_loss_grads(jcm_handle h, const float* x, const float* y, int B, int H, int W, int use_sm, float lmbd, float* grads,
                         float* losses) {
  JCM_TRY(need_train(h));
  if (!x || !y || !grads || !losses || B < 1 || H < 8 || W < 8) return fail(JCM_ERR_ARG, "bad train_loss_grads arguments");
  if (use_sm && !h->has_sm) return fail(JCM_ERR_STATE, "use_sm needs the spatial-model parameters");
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (h->call_depth > 1) return fail(JCM_ERR_STATE, "jcm_train_loss_grads changes the handle's training state or parameters and cannot be called from the gradient-ready callback of the same handle");
  jcm_ctx* c = h;
  HIP_TRY(hipMemsetAsync(grads, 0, c->train->total * sizeof(float), c->stream));   // tensors the loss does not reach keep a zero gradient
  return with_arena(c, [&] { return loss_grads_impl(c, x, y, B, H, W, use_sm, lmbd, grads, losses); });
}

// The two gradient kernels of ONE stride-1 layer on caller-supplied tensors -- the route the training step takes on this handle (frequency
// domain, direct fp32 MFMA chain, split operands), isolated from the rest of the step so that tests can hold the kernels themselves to a tight
// bound (inside a full step a ReLU / max-pool decision that rounds the other way upstream moves a gradient by far more than kernel error).
int jcm_train_layer_grads(jcm_handle h, const char* scope, const float* x, const float* dz, int B, int H, int W, float lmbd, float* grads, float* dx_out) {
  JCM_TRY(need_train(h));
  if (!scope || !x || !dz || !grads || B < 1 || H < 1 || W < 1) return fail(JCM_ERR_ARG, "bad train_layer_grads arguments");
  if (h->precision != JCM_PRECISION_F32) return fail(JCM_ERR_ARG, "train_layer_grads: fp32 handles only (bf16 handles keep bf16 tensors between the layers)");
  DeviceGuard g(h->device);
  CallOrder order(h);
  jcm_ctx* c = h;
  TrainState* t = c->train;
  const ConvLayer* L = conv_of(c, scope);
  if (!L || L->cin == 3 || !grad_of(t, grads, std::string(scope) + "/weights")) return fail(JCM_ERR_ARG, std::string("train_layer_grads: '") + scope + "' is not a stride-1 conv layer");
  t->gscale_of = nullptr;
  return with_arena(c, [&] {
    LayerFwd f;
    f.scope = scope;
    JCM_TRY(conv_train_fwd_conv(c, f, 1, x, B, H, W, 1));      // (the frequency-domain weight gradient reads the input spectra the forward pass keeps)
    const size_t NPX = (size_t)B * H * W;
    const void* dl = dz;
    int ldl = L->cout, ldl_fft = 0;
    if (L->cout % 16) {      // the logits layer: its gradient travels with a 16-channel stride, widened to 64 for the frequency-domain route (loss_grads_impl)
      constexpr int LDZ = 16, LDZF = 64;
      float* d16 = arena_alloc<float>(c, NPX * LDZ);
      if (!c->dry) HIP_TRY(pad_channels_f32(dz, L->cout, d16, LDZ, NPX, c->stream));
      dl = d16; ldl = LDZ;
      if (f.xs) {
        float* d64 = arena_alloc<float>(c, NPX * LDZF);
        if (!c->dry) HIP_TRY(pad_channels_f32(d16, LDZ, d64, LDZF, NPX, c->stream));
        dl = d64; ldl = LDZF; ldl_fft = LDZF;
      }
    }
    JCM_TRY(conv_wgrad(c, f, dl, ldl, B, lmbd, grads));
    if (dx_out) JCM_TRY(conv_dgrad(c, f, dl, B, dx_out, ldl_fft));
    return (int)JCM_OK;
  });
}

int jcm_train_apply(jcm_handle h, const float* grads, int optimizer, float lr, float clip_norm, float* grad_norm_out) {
  JCM_TRY(need_train(h));
  if (!grads || !(lr >= 0.f)) return fail(JCM_ERR_ARG, "bad train_apply arguments");
  if (optimizer != JCM_OPT_ADAM && optimizer != JCM_OPT_MOMENTUM) return fail(JCM_ERR_ARG, "wrong optimizer");   // main.py:506
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (h->call_depth > 1) return fail(JCM_ERR_STATE, "jcm_train_apply changes the handle's training state or parameters and cannot be called from the gradient-ready callback of the same handle");
  jcm_ctx* c = h;
  TrainState* t = c->train;
  const bool clip = clip_norm > 0.f;
  HIP_TRY(sum_squares(grads, t->total, t->sumsq, 0, t->red, c->stream));         // tf.clip_by_global_norm (main.py:302-309)
  const long step = t->step + 1;          // n_iters advances only once the update has been enqueued (a failed launch must not move the LR schedule)
  const double b1 = 0.9, b2 = 0.999;
  const float lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)step)) / (1.0 - std::pow(b1, (double)step)));
  if (optimizer == JCM_OPT_ADAM)
    HIP_TRY(optimizer_chunks(t->ck_w, t->ck_start, t->ck_off, t->ck_len, t->n_chunks, grads, t->opt_m, t->opt_v, clip ? t->sumsq : nullptr,
                             clip_norm, lr_t, 0.9f, 0.999f, 1e-8f, 0, c->stream));
  else
    HIP_TRY(optimizer_chunks(t->ck_w, t->ck_start, t->ck_off, t->ck_len, t->n_chunks, grads, t->opt_m, t->opt_v, clip ? t->sumsq : nullptr,
                             clip_norm, lr, 0.9f, 0.f, 0.f, 1, c->stream));
  t->step = step;
  if (grad_norm_out) {
    double ss = 0.0;
    HIP_TRY(hipMemcpyAsync(&ss, t->sumsq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    order.release();      // other host threads of the device go on while this one waits
    const hipError_t se = hipStreamSynchronize(c->stream);
    order.acquire();
    HIP_TRY(se);
    *grad_norm_out = (float)std::sqrt(ss);
  }
  JCM_TRY(refresh_derived(c, false));   // packed weights, folded moving statistics, softplus'd priors + spectra
  JCM_TRY(repack_dgrad(c));
  return JCM_OK;
}

int jcm_train_set_grad_callback(jcm_handle h, jcm_grad_ready_fn fn, void* user) {
  JCM_TRY(need_train(h));
  h->train->ready_fn = fn;
  h->train->ready_user = user;
  return JCM_OK;
}

// Saver.save / Saver.restore of the optimizer side of the session (main.py:604,612,666 save every global variable: the
// '<var>/Adam', '<var>/Adam_1' -- or '<var>/Momentum' -- slots, beta1_power / beta2_power and n_iters).  slot 0 = first
// moment / momentum accumulator, slot 1 = second moment; same flat layout as the gradient buffer.
int jcm_train_get_state(jcm_handle h, int slot, float* out, int64_t count, int64_t* n_iters) {
  JCM_TRY(need_train(h));
  TrainState* t = h->train;
  if (slot < 0 || slot > 1 || (out && count != (int64_t)t->total)) return fail(JCM_ERR_ARG, "bad train_get_state arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, slot ? t->opt_v : t->opt_m, t->total * sizeof(float), hipMemcpyDefault, h->stream));
    order.release();
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (n_iters) *n_iters = t->step;
  return JCM_OK;
}

int jcm_train_set_state(jcm_handle h, int slot, const float* data, int64_t count, int64_t n_iters) {
  JCM_TRY(need_train(h));
  TrainState* t = h->train;
  if (slot < 0 || slot > 1 || (data && count != (int64_t)t->total) || n_iters < 0) return fail(JCM_ERR_ARG, "bad train_set_state arguments");
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (data) {
    HIP_TRY(hipMemcpyAsync(slot ? t->opt_v : t->opt_m, data, t->total * sizeof(float), hipMemcpyDefault, h->stream));
    order.release();
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller may free `data` on return
  }
  t->step = (long)n_iters;
  return JCM_OK;
}

int jcm_train_steps(jcm_handle h, int64_t* n_iters) {
  JCM_TRY(need_train(h));
  if (n_iters) *n_iters = h->train->step;
  return JCM_OK;
}

int jcm_get_tensor(jcm_handle h, const char* name, float* out, int64_t count) {
  JCM_TRY(check(h, false));
  if (!name || !out) return fail(JCM_ERR_ARG, "bad get_tensor arguments");
  const Tensor* t = find(h, name);
  if (!t) return fail(JCM_ERR_STATE, std::string("no parameter '") + name + "'");
  if ((int64_t)t->n != count) return fail(JCM_ERR_ARG, std::string("'") + name + "' has " + std::to_string(t->n) + " elements");
  DeviceGuard g(h->device);
  CallOrder order(h);
  HIP_TRY(hipMemcpyAsync(out, t->d, t->n * sizeof(float), hipMemcpyDefault, h->stream));
  order.release();
  HIP_TRY(hipStreamSynchronize(h->stream));
  return JCM_OK;
}

int jcm_update_tensor(jcm_handle h, const char* name, const float* data, int64_t count, int refresh) {
  JCM_TRY(check(h, true));
  if (!name || !data) return fail(JCM_ERR_ARG, "bad update_tensor arguments");
  auto it = h->params.find(name);
  if (it == h->params.end()) return fail(JCM_ERR_STATE, std::string("no parameter '") + name + "'");
  if ((int64_t)it->second.n != count) return fail(JCM_ERR_ARG, std::string("'") + name + "' has " + std::to_string(it->second.n) + " elements");
  DeviceGuard g(h->device);
  CallOrder order(h);
  if (h->call_depth > 1) return fail(JCM_ERR_STATE, "jcm_update_tensor changes the handle's training state or parameters and cannot be called from the gradient-ready callback of the same handle");
  HIP_TRY(hipMemcpyAsync(it->second.d, data, it->second.n * sizeof(float), hipMemcpyDefault, h->stream));
  order.release();
  const hipError_t se = hipStreamSynchronize(h->stream);   // the caller may free `data` on return
  if (refresh) order.acquire();
  HIP_TRY(se);
  if (refresh) {
    JCM_TRY(refresh_derived(h, false));
    if (h->train) JCM_TRY(repack_dgrad(h));
  }
  return JCM_OK;
}

}  // extern "C"

namespace jcm {
void train_destroy(jcm_ctx* c) {
  delete c->train;      // device buffers are in c->owned
  c->train = nullptr;
}
}  // namespace jcm
