// Internal context of libjcm shared by the translation units behind include/jcm.h
// (jcm_api.hip: inference graph; jcm_train.hip: training step).  Not part of the ABI.
#pragma once
#include "../../include/jcm.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace jcm {

struct CallOrder;

int fail(int code, const std::string& msg);     // sets the thread-local message of jcm_last_error()

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return ::jcm::fail(JCM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)
#define JCM_TRY(expr)          \
  do {                         \
    int r_ = (expr);           \
    if (r_ != JCM_OK) return r_; \
  } while (0)
// main.py:18 -- channel order of the heat maps and of the pair tables.
extern const char* const kJointNames[10];
constexpr int kC = 10;            // heat-map channels seen by the spatial model (9 joints + torso)
constexpr float kBnEps = 1e-3f;   // tf.contrib.layers.batch_norm default epsilon
constexpr int kHmH = 60, kHmW = 90, kHmHW = kHmH * kHmW;
constexpr int kPrH = 120, kPrW = 180;
constexpr int kCH = 61, kCW = 91;
constexpr size_t kFrame = (size_t)kPrH * kPrW;          // 120*180 real
constexpr size_t kSpec = (size_t)kPrH * (kPrW / 2 + 1); // 120*91 complex

struct Tensor {
  std::vector<int64_t> shape;
  float* d = nullptr;
  size_t n = 0;
};

struct ConvLayer {
  int ks = 0, cin = 0, cout = 0, coutp = 0;
  bool has_bn = false;
  const float* w_raw = nullptr;   // HWIO (conv1 kernel reads it directly)
  float* wp = nullptr;            // packed for conv_igemm_f32
  mutable bool wp_stale = false;  // a refresh after a weight update left `wp` behind: repacked where a direct fp32 kernel next reads it (run_conv_layer)
  void* wp_split = nullptr;       // two fp16 parts per weight for conv_split_f32 (fp32 handles, "f32_conv" = 2)
  int coutp_split = 0;
  float* wscale = nullptr;        // fp16x3: device {Sw, 1/Sw}, the power-of-two scale the packed weights carry
  void* wp_bf16 = nullptr;        // packed for conv_igemm_bf16
  void* wp_kxfold = nullptr;      // logits layer packed for conv_kxfold_bf16 (bf16 handles, Cout == 9)
  void* wq1_bf16 = nullptr;       // packed for conv1_mfma_pool (5x5, Cin=3, Cout=64)
  float* wq1_f32 = nullptr;       // packed for conv1_mfma_pool_f32 (fp32 handles)
  void* wq1_split = nullptr;      // packed for conv1_mfma_pool_split (fp32 handles, default route)
  int coutp_bf16 = 0;
  bool thin = false;              // fp32: conv_thin_f32 instead of conv_igemm_f32
  bool thin_bf16 = false;         // bf16: conv_thin_bf16 (fp32 output) instead of conv_igemm_bf16
  const float* bias = nullptr;
  float* scale = nullptr;
  float* shift = nullptr;
};

struct TrainState;   // jcm_train.hip
}  // namespace jcm
struct jcm_ctx;
namespace jcm {
void train_destroy(jcm_ctx* c);

}  // namespace jcm

struct jcm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int precision = JCM_PRECISION_F32;
  int K = 9;
  int f32_conv = 0;             // fp32 handles: 0 = default (frequency domain, or the exact fp32 MFMA chain with conv9_fft = 0), 2 = the direct fp16x3 split kernels (forward and gradients)
  float* act_scale = nullptr;   // fp16x3: device {S, 1/S} of the current layer input (computed before every launch), + scratch
  float* scale_scratch = nullptr;
  int split_min_wgs = 128;      // grids smaller than this keep the exact kernel (option "split_min_wgs")
  bool finalized = false;
  std::map<std::string, jcm::Tensor> params;
  std::map<std::string, jcm::ConvLayer> convs;
  std::vector<void*> owned;   // device allocations made at finalize
  // spatial model tables
  bool has_sm = false;
  float* sp_energy = nullptr;   // [P][120*180]
  float* sp_bias = nullptr;     // [P][5400]
  float* bn_sm_scale = nullptr; // [10]
  float* bn_sm_shift = nullptr;
  int* cond = nullptr;          // [P] conditioning channel of pair p
  const float** energy_ptrs = nullptr;   // [P] device table of the energy_* / bias_* parameter tensors, graph order
  const float** bias_ptrs = nullptr;
  int* cond0 = nullptr;         // single zero (jcm_conv_mrf)
  int sm_algo = 3;              // 3 = every transform in LDS (sm_fused.hip), 1 = direct sliding-window VALU kernel (the cross-check)
  // transient, set by jcm_pd_forward around two consecutive frequency-domain layers: the first writes the second's row-transformed input
  void* fft_t_next = nullptr;
  const void* fft_t_in = nullptr;
  void* fft_xs = nullptr;            // transient: the next frequency-domain layer keeps its split input spectra here (training step: the weight gradient reads them)
  bool fft_xs_ready = false;         // ... they are there already (data gradient after the weight gradient of the same layer): skip the forward transforms
  void* sm_scratch = nullptr;   // sm_fused.hip: partial sums + flags of sm_inv_finish_kernel's cuts (sm_fused_scratch_bytes(), zeroed once)
  unsigned sm_epoch = 0;        // ... the launch counter its flags carry
  int fft_fuse = 3;             // fp32 handles, jcm_pd_forward: bit 0 = conv2 -> max pool -> conv3, bit 1 = conv4_fullres -> branch merge -> conv5 handed over in row-transformed form (conv_fft_rows_fused.hip)
  int bf16_hpool = 1;           // bf16 handles: the horizontal half of pool2 in conv2's epilogue (ConvArgs::hpool) + vpool_2x1_bf16 instead of the 2x2 pool kernel
  int conv_hpool = 0;           // transient: the next direct bf16 convolution launch takes the half pool
  int fft_rows_mfma = 1;        // bf16 handles with 16-bit row-transformed tensors: conv5's inverse row pass on the matrix cores (ConvArgs::rows_mfma; conv_fft_rows_mfma.hip)
  int fft_next_pool = 0, fft_next_ks = 0;      // transient, with fft_t_next: a 2x2 max pool lies between this layer and the one fft_t_next is for (kernel size fft_next_ks)
  // transient: the next frequency-domain layer's input windows are cut from this map by its forward row pass (ConvArgs::win_map; jcm_train.hip)
  const void* fft_win_map = nullptr;
  int fft_win_B = 0, fft_win_H = 0, fft_win_W = 0, fft_win_TY = 0, fft_win_TX = 0;
  bool fft_win_scatter = false;      // ... and its inverse row pass stores the valid regions into the map `out` (same geometry; ConvArgs::wout_*)
  bool fft_t_in_16 = false;                     // transient, with fft_t_in (bf16 handles): the handed-over T is complex fp16 + its scale words (conv4_fullres -> conv5)
  const void* fft_next_merge = nullptr;        // transient, with fft_t_next: const jcm::FftMerge* -- fft_t_next is the row-transformed MERGED map (this layer = the full-resolution branch)
  const void* fft_merge = nullptr;   // const jcm::FftMerge*: the next frequency-domain layer forms the merged map itself (jcm_pd_forward, conv5)
  int conv9_fft = 1;            // fp32 handles: wide 9x9 layers in the frequency domain (conv_fft.hip) when the shape allows; 0 = fp32 MFMA chain
  int fft_single = 1;           // bf16 handles: the channel GEMM on ONE scaled fp16 part per operand (np = 5; 0 = two bf16 parts, three products)
  int fft_win = 1;              // training step of fp32 handles: frequency-domain layers on 32 x 32 overlap-save windows where that shrinks the filter-sized spectra (jcm_train.hip)
  int fft_t16 = 1;              // bf16 handles on the one-part route (fft_single): the row-transformed tensors T / T' as complex fp16 in block floating point (Fp16Scale::t16)
  // device words of the fp16 scaling (kernels.h: Fp16Scale): zeroed floats, one per image of every row-transformed tensor of a call.  They come from
  // blocks of kFftWords floats; a call that needs more than a block holds (a forward of > 20 000 images in one piece) gets further blocks on demand,
  // and the blocks are re-zeroed and reused from the start BETWEEN calls (CallOrder), in stream order behind every kernel that read the old words.
  static constexpr int kFftWords = 1 << 18, kFftWordsPerCall = 1 << 16;
  struct WordBlock { float* p = nullptr; int cap = 0; };
  std::vector<WordBlock> fft_blocks;
  int fft_block_i = 0, fft_word_i = 0;      // next free word: fft_blocks[fft_block_i].p + fft_word_i
  int call_order = 1;               // 0: this handle's calls are not ordered against other handles' (debugging only)
  int debug_skip = 0;               // bisecting aid (jcm_pd_forward): bit 0 conv1(+pool1), 1 pool2, 2 conv2, 3 conv3, 4 conv4, 5 merge + conv5, 6 conv6 are NOT launched
  float* fft_tmax_in = nullptr;     // transient: the word of the next frequency-domain layer's input (set with fft_t_in / fft_xs_ready by whoever produced that tensor)
  float* fft_last_tmax = nullptr;   // the word the last frequency-domain layer's input used (the training step keeps it with the kept spectra)
  struct FftW { void* p = nullptr; size_t bytes = 0; bool valid = false; float* wscale = nullptr; };      // wscale: two device floats behind the spectra (np = 4)
  std::map<std::string, FftW> fft_w;   // filter spectra per "<scope>@HxW", computed on first use, invalidated by refresh_derived
  int sm_chunk = 32;            // training step: images per slice of the spatial model's backward pass (81 + 10 spectra per image live at once)
  int micro_batch = 0;          // jcm_forward walks a batch in slices of this many images (0 = 256 bf16 / 64 fp32)
  float2* prior_spec_t = nullptr; // [P][91][120]: transposed half spectra of softplus5(energy) (sm_lds.hip)
  // workspace arena (stack allocator, grown on demand between forwards)
  char* arena = nullptr;
  size_t arena_cap = 0, arena_off = 0, arena_peak = 0;
  bool dry = false;             // sizing pass: allocate offsets only, launch nothing
  size_t param_bytes = 0;
  // per-layer HIP-event timing on the launch stream (bench.py roofline object)
  bool profile = false;
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof;
  std::vector<hipEvent_t> event_pool;   // recycled by jcm_profile_read / "profile"=0, destroyed by jcm_destroy
  jcm::TrainState* train = nullptr;   // created by jcm_train_begin
  std::mutex call_mu;                 // held by the thread whose outermost entry point of this handle is running (CallOrder)
  int call_depth = 0;                 // entry points of this handle on that thread's stack (> 1 only inside a gradient-ready callback)
  jcm::CallOrder* order = nullptr;    // the outermost running entry point's chain guard (notify_ready suspends it around the user callback)
};

namespace jcm {

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Calls of DIFFERENT handles on one device are ordered on the GPU: an entry point that enqueues work holds the device's lock for as long
// as it is enqueuing (so two host threads never interleave their launches), first makes its stream wait (hipStreamWaitEvent, the host does
// not block) for the event the previous call of another stream recorded, and records the event again behind its own last kernel.  Each
// call fills the chip on its own, so nothing is lost; what is gained is that no kernel of this library ever runs beside a kernel of another
// handle.  The per-XCD L2s are not coherent with each other: a kernel that touches memory another stream is producing (the look-ahead
// reads past the end of a buffer that cgemm_split.hip used to make were such touches) leaves stale lines that the other stream's next
// kernel then consumes -- found by tests/test_gpu_golden.py::test_two_engines_two_streams_soak.  The known over-reads are fixed at the
// source; this ordering is the guarantee that does not depend on having found them all.  jcm_set_option("call_order", 0) takes a handle
// out of the chain (it neither waits nor records; tools/determinism.py and the soak tests use it to look for what the chain would hide).
// The constructor also clears the transient hand-over fields an aborted call may have left behind and laps the fp16 scale-word ring.
// The lock is NOT held while the host blocks or while user code runs: release() (record the chain event, unlock) precedes every
// host-side wait at the end of an entry point, and the gradient-ready callback of jcm_train_loss_grads runs between release() and acquire(),
// so a callback may call jcm_* entry points (a nested call on the SAME handle keeps the outer call's transient state: `nested`).
struct CallOrder {
  jcm_ctx* c;
  std::unique_lock<std::mutex> lk;       // the device's call chain
  std::unique_lock<std::mutex> hlk;      // the handle (outermost call only)
  bool nested = false;
  explicit CallOrder(jcm_ctx* ctx);
  ~CallOrder();
  void acquire();      // lock the device's chain and make this stream wait for the previous call of another stream
  void release();      // record the chain event behind what has been enqueued so far and unlock (idempotent)
  CallOrder(const CallOrder&) = delete;
  CallOrder& operator=(const CallOrder&) = delete;
};

template <class T>
T* arena_alloc(jcm_ctx* c, size_t count) {
  const size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
  const size_t off = c->arena_off;
  c->arena_off += bytes;
  if (c->arena_off > c->arena_peak) c->arena_peak = c->arena_off;
  return reinterpret_cast<T*>(c->arena + off);   // in a dry pass arena may be null: offsets only
}

int arena_reserve(jcm_ctx* c, size_t bytes);
int dev_alloc(jcm_ctx* c, void** p, size_t bytes);
int sm_scratch_next(jcm_ctx* c, void** scratch, unsigned* epoch);      // sm_fused_forward's scratch (allocated + zeroed at first use) and the next launch epoch
const Tensor* find(jcm_ctx* c, const std::string& name);
int check(jcm_handle h, bool need_final);
const ConvLayer* conv_of(jcm_ctx* c, const std::string& scope);
int fold_bn(jcm_ctx* c, const std::string& scope, int n, float** scale, float** shift);
// Rebuild every derived table (packed weights, folded BN, softplus'd priors and their spectra) from
// the parameter store; called by jcm_finalize and after each optimizer update.
int refresh_derived(jcm_ctx* c, bool first);
// HIP-event pairs for the per-layer timing come from a pool: inside a timed region the only cost is two
// hipEventRecord per launch (events are created on first use and recycled by jcm_profile_read).
int prof_begin(jcm_ctx* c, hipEvent_t* e0, hipEvent_t* e1);
void prof_end(jcm_ctx* c, const std::string& scope, hipEvent_t e0, hipEvent_t e1, bool ok);
void prof_release_all(jcm_ctx* c, bool destroy);
// frequency-domain route (jcm_api.hip): takes_fft() says whether a layer / shape goes there; run_conv_fft() runs it (filter spectra cached in
// c->fft_w under "<scope>@HxW", packed from L->w_raw when missing or invalidated); the training step uses both for its data gradient.
bool takes_fft(jcm_ctx* c, const ConvLayer* L, int B, int H, int W);
bool fft_spectra_valid(jcm_ctx* c, const std::string& scope, int H, int W, int circ = 0);
// operand form of the channel GEMM on this handle (kernels.h): bf16 handles: 5 (one scaled fp16 part, default) or 2 (two bf16 parts); fp32 handles: 4 (two scaled fp16 parts)
inline int fft_np(const jcm_ctx* c) { return c->precision == JCM_PRECISION_BF16 ? (c->fft_single ? 5 : 2) : 4; }
int fft_new_words(jcm_ctx* c, int n, float** w);      // n zeroed device words of the scaling ring (one per image)
// circ: x is a batch of overlap-save windows [B, H, W, Cin] that fill the transform, out their valid regions [B, H - 8, W - 8, Cout] (ConvArgs::circ)
int run_conv_fft(jcm_ctx* c, const ConvLayer* L, const std::string& scope, const void* x, int B, int H, int W, void* out, int in_layout, int out_layout, int circ = 0);
int run_conv_layer(jcm_ctx* c, const ConvLayer* L, const std::string& scope, int stride, const void* x, int B, int H, int W, int sub,
                   void* out, bool act_bf16, bool out_f32, int in_planar = 0, int out_planar = 0);   // bf16 layouts: ConvArgs in kernels.h

inline int cdiv2(int v) { return (v + 1) / 2; }

// Sizing pass then the real pass, so the arena never reallocates mid-graph.
template <class F>
int with_arena(jcm_ctx* c, F&& body) {
  // a call from inside the gradient-ready callback of this handle's running training step would overwrite that step's workspace
  if (c->call_depth > 1) return fail(JCM_ERR_STATE, "this entry point uses the handle's workspace and cannot be called from the gradient-ready callback of the same handle");
  c->dry = true;
  c->arena_off = 0;
  c->arena_peak = 0;
  int r = body();
  c->dry = false;
  if (r != JCM_OK) return r;
  JCM_TRY(arena_reserve(c, c->arena_peak));
  c->arena_off = 0;
  return body();
}

}  // namespace jcm
