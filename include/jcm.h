/* libjcm -- C ABI of the MI355X-native joint-heat-map inference path.
 *
 * The reference (max-andr/joint-cnn-mrf) has no FFI layer: the path is four module-level
 * Python functions in main.py evaluated by TensorFlow at sess.run (main.py:280,406).  Each
 * entry point below cites the reference interface it replaces; the Python host module
 * `joint-cnn-mrf_amd/main.py` binds them through ctypes and keeps the reference's names.
 *
 * Conventions
 *   - every tensor pointer is DEVICE memory owned by the caller (a torch-ROCm allocation),
 *     fp32, dense NHWC exactly as the reference lays it out; coords are int32;
 *   - images are H rows x W cols (480 x 720, data.py:10), heat maps 60 x 90 (data.py:12);
 *   - calls enqueue asynchronously on the stream given to jcm_create and do not synchronise;
 *     one handle per (device, stream); a handle is not thread-safe by contract -- two host threads that call it anyway are SERIALISED
 *     (the handle carries a mutex for the duration of an outermost entry point; re-entry from the same thread is only possible from the
 *     gradient-ready callback, see jcm_train_set_grad_callback);
 *   - every function returns 0 on success, non-zero on error; jcm_last_error() returns a
 *     thread-local message for the last failing call;
 *   - the library owns only its packed weights, precomputed spatial-model tables and its
 *     workspace arena; all are released by jcm_destroy.
 */
#ifndef JCM_H
#define JCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jcm_ctx* jcm_handle;

#define JCM_OK 0
#define JCM_ERR_ARG 1      /* bad shape / name / null pointer                      */
#define JCM_ERR_STATE 2    /* parameter missing, handle not finalised, ...          */
#define JCM_ERR_HIP 3      /* a HIP runtime call or kernel launch failed            */

#define JCM_PRECISION_F32 0   /* fp32 tensors; fp32-class arithmetic: by default the stride-1 layers run in the frequency domain with every fp32
                                 spectrum fed to the 16-bit matrix cores as two scaled fp16 parts (22 significant bits);
                                 "conv9_fft" = 0 selects the exact fp32 MFMA accumulation chain (v_mfma_f32_32x32x2_f32) -- the parity path */
#define JCM_PRECISION_BF16 1  /* bf16 tensors between the layers, fp32 accumulate -- the roofline path */

/* -- lifecycle --------------------------------------------------------------------------
 * Replaces tf.Session(config, graph) / sess.close (main.py:606-608,677-680). `stream` is a
 * hipStream_t (NULL = the device's default stream). */
int jcm_create(int device, void* stream, jcm_handle* out);
int jcm_destroy(jcm_handle h);
const char* jcm_last_error(void);
int jcm_abi_version(void);

/* -- options ------------------------------------------------------------------------------
 * "precision": JCM_PRECISION_*  (the reference is fp32 throughout)
 * "n_joints" : K, default 9      (main.py:458)
 * "f32_conv"  : fp32 handles only; arithmetic of the DIRECT (not frequency-domain) convolution kernels, i.e. of every layer when
 *              "conv9_fft" = 0 and of the shapes the frequency-domain route does not take otherwise: 0 (default) = the exact fp32 MFMA
 *              chain; 2 = the stride-1 layers with Cin % 16 == 0 and Cout % 128 == 0 as two-way fp16 operand splits with three
 *              products on the 16-bit matrix cores (fp32-class error; every operand tensor -- weights, layer inputs, gradients --
 *              is lifted into the fp16 range by its own power-of-two scale first), and the frequency-domain route off: the
 *              A/B arm of that route.  (1, three bf16 parts / six products, was retired in round 5: JCM_ERR_ARG.)
 * "split_min_wgs": any time; grids smaller than this keep the exact kernel (default 128, 0 = always split).
 * All three must be set before jcm_finalize.
 * "profile"  : 0/1, any time: bracket every MFMA conv launch with HIP events on the launch
 *              stream; read the totals back with jcm_profile_read.  Events come from a pool owned by the
 *              handle (created on first use, recycled by jcm_profile_read and by switching the option on,
 *              destroyed by jcm_destroy), so a profiled step only records.
 * "conv9_fft": any time, default 1: stride-1 convolutions in the frequency domain (conv_fft*.hip: in-LDS FFTs around one complex channel
 *              product per frequency on the bf16 matrix cores with split operands, cgemm_split.hip) -- every such layer of an fp32
 *              handle, the wide 9x9 layers of a bf16 handle -- whenever the shape allows (Cin % 64 == 0, map + kernel - 1 <= 192);
 *              0 = the direct MFMA kernels.  The training step of an fp32 handle takes the same route (forward, data and weight
 *              gradients); a bf16 handle trains on the direct bf16 kernels.  Filter spectra are built per (layer, map size) on
 *              first use (11.4 GB for the full-width model on 60x90 maps; cache bound: environment JCM_FFT_CACHE_GB, default 64).
 *              Environment JCM_FFT_REG=0 (read once per process): the LDS kernels instead of the register-resident transforms of
 *              csrc/conv_fft_rows_reg.hip (inverse column / row passes, the bf16 forward row pass, the fused inverse + forward row
 *              pass) -- the A/B arm; results agree to fp32 rounding.
 *              An fp32 handle WITH TRAINING STATE holds more: a second 11.4 GB set of spectra of the flipped, transposed filters for
 *              the data gradient (both sets are repacked after every update) and up to 7.5 GB of weight-gradient scratch (the per-frequency
 *              products P and the column sums R of conv5) in the workspace arena -- about 31 GB beside the 6 GB of activations.
 * "call_order": any time, default 1: calls of different handles on one device are ordered one after the other on the GPU -- an entry
 *              point holds a per-device lock while it enqueues, makes its stream wait for the previous call of another stream and records
 *              an event behind its last kernel -- so they may come from different host threads and streams and results do not depend on
 *              the interleaving.  0 takes the handle out of that chain (debugging: tools/determinism.py).
 * "fft_single": any time, default 1 (bf16 handles): the channel product of the frequency-domain route on ONE fp16 part per operand -- spectra scaled
 *              by one power of two per image (derived from a rigorous bound of the spectrum, so an image's result does not depend on its batch), rounded once to fp16's 11 significant bits, one real product per multiply,
 *              32 channels per GEMM stage.  The layer's input and output tensors are bf16 (8 bits): the spectra are eight times finer.
 *              0 = two bf16 parts per operand, three products (rounds 2-3).  Changing it drops the cached filter spectra.
 *              ACCURACY CLASS of the default bf16 route (fft_single = fft_t16 = 1): 11-bit intermediates inside the wide 9x9 layers --
 *              per layer within one bf16 ulp + 1e-3 of the layer's scale of the bf16-operand oracle (7.6-8.0 % of the entries one ulp off;
 *              0.2 % with both options 0), full tower 4.3e-3 of the logit scale; arg-max agreement with the fp32 engine on 256 images
 *              97.6 % (part detector) / 96.3 % (spatial model), the same as the strict arm's 97.4 / 96.4 % and the direct bf16 MFMA
 *              kernels' 97.4 / 96.4 %, and 100 % / 99.9 % of the joints whose fp32 top-2 margin is clear of the bf16 noise
 *              (tests/test_gpu_argmax_agreement.py).
 * "fft_windows": any time, default 1 (fp32 handles with training state): the training step runs its wide 60x90 layers (conv4_fullres, conv5 -- every
 *              layer with Cin * Cout >= 128 * 256 (round 6; 256 * 512 before) whose map has at least 1.5 x the frequencies of a window: conv3_fullres and the 30 x 45 maps of conv3_halfres / conv4_halfres too) on 32 x 32 overlap-save windows: forward, data
 *              gradient and weight gradient see 3 x 4 windows per image as a batch of 12 B images on a 32 x 32 circular transform, so the filter-sized
 *              spectra (what bounds the step at 16 images per GPU) shrink 5.8x.  0 = the 64 x 96 transform of the whole map (round 3).
 * "fft_t16"  : any time, default 1 (bf16 handles with "fft_single" = 1): the row-transformed tensors between the row and the column passes of
 *              the frequency-domain route (half of the transform passes' HBM traffic) as complex fp16 in block floating point -- one power-of-two
 *              scale per (image, row, 64 channels) tile forward and per (image, kx, 64 channels) tile inverse, 11 significant bits like the
 *              spectra; and the product spectra between the channel GEMM and the inverse column pass as complex fp16 under a CONSTANT
 *              power-of-two shift (2^-(ceil(log2 Cin) + 14): the scaled operands bound every product, so nothing can overflow and typical
 *              entries sit fourteen binades above fp16's smallest normal number; round 5).  0 = complex fp32 for all three (round 3).
 * "fft_fuse" : any time, default 3 (jcm_pd_forward / jcm_forward on the frequency-domain route): hand-overs in row-transformed form, ONE kernel doing the
 *              inverse row transform + bias / ReLU / BatchNorm of the producing layer, the op between the layers and the forward row transform of the
 *              consuming layer.  bit 0 (fp32 handles) = conv2 -> 2x2 max pool -> conv3: a work group owns a row pair, takes the 2x2 maximum in LDS
 *              and transforms the pooled row; neither conv2's output nor the pooled map reaches HBM.  bit 1 = conv4_fullres -> branch merge ->
 *              conv5 for the model's geometry (90-column maps, branches at 1/2 and 1/4): ((x1 + up(x2)) + up(x3)) / 3 formed in registers, x1 never
 *              reaches HBM -- fp32 handles (complex fp32 T) and bf16 handles on the one-part route (16-bit T; x1 and the merged value are rounded to
 *              bf16 exactly where the separate kernels round them: the two arms of a bf16 handle are bit-identical).  Arithmetic: the pool hand-over
 *              evaluates the separate kernels' expressions; the merge hand-over lerps the coarse branches along y first, then along x (TF lerps x
 *              first: the last fp32 bit of the coarse terms, as the register merge of bf16 handles has done since round 5), third = correctly rounded
 *              x / 3 on fp32 handles, one multiplication by RN(1/3) on bf16 handles (against the quotient: the bf16 rounding of two merged values in a
 *              million).  0 = the separate kernels of round 5 (A/B arm; held by the same tests).
 * "fft_rows_mfma" : any time, default 1 (bf16 handles on the one-part route with 16-bit row-transformed tensors): conv5's 96-point inverse row pass as a MATRIX
 *              PRODUCT on the matrix cores (rows_inv_mfma_kernel, conv_fft_rows_mfma.hip): T' (complex fp16) times the 96 x 98 real inverse-transform
 *              matrix held as two fp16 parts (22 significant bits: as exact as the fp32 butterflies), bias / ReLU / BatchNorm on the accumulators, planar
 *              bf16 out.  The register kernel it replaces is bound by vector-ALU issue (a 96-point transform is ~1000 scalar fp32 instructions per row
 *              and channel).  The two arms agree to fp32-level noise in front of the bf16 rounding (rms 1e-5 of the logit scale).  0 = the register kernel.
 * "bf16_hpool" : any time, default 1 (bf16 handles): the horizontal half of the 2x2 max pool behind conv2 is taken in conv2's epilogue (a lane pair of
 *              conv5_strip_bf16_kernel is a pixel pair; even widths) and a two-row kernel finishes the pool: the full-width conv2 map is neither
 *              written nor re-read.  Bit-identical to the 2x2 pool kernel (rounding to bf16 is monotonic).  0 = the 2x2 pool kernel.
 * "sm_algo"  : any time; the pairwise convolutions of the spatial model (main.py:83-87): 3 (default) = every 120x180 transform in LDS,
 *              hand-written (sm_fused.hip; jcm_conv_mrf, the prior spectra and the training step's backward use the whole-frame
 *              kernels of sm_lds.hip); 1 = direct sliding-window kernel, the independent cross-check.  Both pass the same parity
 *              tests.  (Rounds 1-4 also had two rocFFT routes, 0 and 2; the library links no FFT library any more.)
 *              "sm_chunk": images per slice of the training step's spatial-model backward (default 32).
 * "micro_batch": any time; jcm_forward walks its batch in slices of this many images, so the workspace is
 *              sized for one slice (a rank's share of BASELINE configs[3]'s 2048 images fits).  0 (default)
 *              = 256 for bf16 handles, 64 for fp32 handles. */
int jcm_set_option(jcm_handle h, const char* key, int64_t value);

/* -- parameters -----------------------------------------------------------------------------
 * Replaces tf.get_variable + Saver.restore (main.py:147,153,484,487,612).  `name` is the
 * reference's TF variable name:
 *   "<scope>/weights" [k,k,Cin,Cout] HWIO      "<scope>/biases" [Cout]
 *   "<scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}" [Cout]
 *     scopes conv{1..4}_{fullres,halfres,quarterres}, conv5, conv6 (main.py:44-72), bn_sm (:112)
 *   "energy_<j>_<c>" [1,120,180,1]   "bias_<j>_<c>" [1,60,90,1]    (main.py:484,487)
 * `data` may be a host or a device pointer (fp32); the library copies it. */
int jcm_set_tensor(jcm_handle h, const char* name, const float* data, const int64_t* shape, int ndim);
/* Packs weights for the MFMA kernels, folds BatchNorm (inference mode, flag_train=False,
 * main.py:406) into per-channel scale/shift, and precomputes softplus(energy),
 * softplus(bias) (main.py:120,122 are batch-independent).  Call after the last set_tensor. */
int jcm_finalize(jcm_handle h);

/* -- part detector ----------------------------------------------------------------------------
 * conv_layer(x, size, stride, n_in, n_out, name, last_layer) (main.py:156-169):
 * BN(relu(conv_SAME(x,w)+b)), or conv+b for the last layer.  x [B,H,W,Cin] -> out
 * [B,ceil(H/s),ceil(W/s),Cout]; size and channel counts come from the stored "<scope>/weights".
 * Kernels exist for the shapes the model uses: (size 5, stride 2, Cin 3) and (size 5|9,
 * stride 1, Cin % 16 == 0); anything else returns JCM_ERR_ARG.  On a bf16 handle (stride-1 layers, Cin % 32 == 0)
 * x and out are still fp32: the input is rounded to bf16, the layer runs on the bf16 MFMA kernel the tower uses and
 * its bf16 result (fp32 for the last layer) is widened back. */
int jcm_conv_layer(jcm_handle h, const char* scope, int stride, int last_layer, const float* x, int B, int H, int W,
                   float* out);
/* conv_layer(((x1 + up(x2)) + up(x3)) / 3) (main.py:58,67,69-71: the three branches merged, then conv5), run as the tower runs it: where the
 * layer takes the frequency-domain route its forward row pass forms the merge while it loads the rows (the merged map never reaches memory),
 * otherwise the merge kernel runs in front of the layer.  x1 [B,H,W,Cin], x2 [B,H2,W2,Cin], x3 [B,H3,W3,Cin] -> out [B,H,W,Cout]; up() =
 * tf.image.resize_images to H x W (TF-1.x legacy bilinear).  bf16 handles: fp32 at the boundary as for jcm_conv_layer.
 * Arithmetic of the merge: the generic kernels (any geometry; fp32 handles, the strict bf16 arm) lerp along x, then along y, and divide by 3 with
 * correct rounding, as TF does; the register kernel of bf16 handles for the model's geometry (90 / 45 / 23 columns, 16-bit T) lerps along y first
 * and multiplies by RN(1/3) -- the last fp32 bit of a value that is then rounded to bf16 (two merged values in a million round the other way). */
int jcm_conv_layer_merged(jcm_handle h, const char* scope, const float* x1, const float* x2, int H2, int W2, const float* x3, int H3, int W3,
                          int B, int H, int W, float* out);
/* max_pool_layer(x, 2, 2) (main.py:172-174): 2x2/2 SAME. [B,H,W,C] -> [B,ceil(H/2),ceil(W/2),C] */
int jcm_max_pool(jcm_handle h, const float* x, int B, int H, int W, int C, float* out);
/* tf.image.resize_images(x, [OH,OW]) (main.py:51,58,60,67,89): TF-1.x legacy bilinear. */
int jcm_resize_bilinear(jcm_handle h, const float* x, int B, int H, int W, int C, int OH, int OW, float* out);
/* model(x, n_joints) (main.py:29-74): x [B,H,W,3] -> logits [B,H/8,W/8,K]. */
int jcm_pd_forward(jcm_handle h, const float* x, int B, int H, int W, float* logits_out);

/* -- heat-map ops ------------------------------------------------------------------------------
 * spatial_softmax(hm) (main.py:212-217): softmax over the HW pixels of every (b,k) map. */
int jcm_spatial_softmax(jcm_handle h, const float* in, int B, int HW, int K, float* out);
/* conv_mrf(A, B) (main.py:77-91): A [1,120,180,1] prior, Bmaps [B,60,90,1] -> out [B,60,90,1]. */
int jcm_conv_mrf(jcm_handle h, const float* A, const float* Bmaps, int B, float* out);
/* spatial_model(heat_map) (main.py:94-125): hm10 [B,60,90,K+1] -> logits [B,60,90,K]. */
int jcm_sm_forward(jcm_handle h, const float* hm10, int B, float* logits_out);
/* get_joints_coords / argmax_hm (evaluation.py:15-24, main.py:389-397): first-occurrence
 * flat argmax per (b,k); coords[b,0,k] = row, coords[b,1,k] = col.  hm [B,HH,WW,K]. */
int jcm_argmax_coords(jcm_handle h, const float* hm, int B, int HH, int WW, int K, int32_t* coords);

/* spatial_softmax (main.py:212-217) and the arg-max of its result (evaluation.py:15-24) in one pass over the
 * logits -- the tail of the tower as jcm_forward runs it.  logits [B,HH,WW,K]; prob [B,HH,WW,K] and coords
 * [B,2,K] may each be NULL (not both). */
int jcm_softmax_argmax(jcm_handle h, const float* logits, int B, int HH, int WW, int K, float* prob, int32_t* coords);

/* -- the whole tower ----------------------------------------------------------------------------
 * The graph of main.py:522-531: model -> spatial_softmax -> concat torso -> spatial_model ->
 * spatial_softmax -> argmax.  x [B,H,W,3]; torso [B,60,90,1] = y_in[...,K:] (main.py:528),
 * may be NULL when use_sm == 0.  Any output pointer may be NULL:
 *   pd_prob, sm_prob [B,60,90,K] fp32;  pd_coords, sm_coords [B,2,K] int32. */
int jcm_forward(jcm_handle h, const float* x, const float* torso, int B, int H, int W, int use_sm,
                float* pd_prob, float* sm_prob, int32_t* pd_coords, int32_t* sm_coords);

/* The same tower with the two losses of the graph in inference mode -- what eval_error runs per batch
 * (main.py:275-283: sess.run([loss_pd, loss_sm, det_rate_pd, det_rate_sm], flag_train=False)).  y = y_in
 * [B,60,90,K+1] (main.py:488): its first K channels are the targets of softmax_cross_entropy (main.py:220-240,
 * 538-539), channel K the torso map.  losses: device fp32 [2] = loss_pd, loss_sm (loss_sm = loss_pd when use_sm == 0,
 * main.py:535).  The detection rates are a few [B,2,K] operations on the returned coordinates (evaluation.py). */
int jcm_eval_forward(jcm_handle h, const float* x, const float* y, int B, int H, int W, int use_sm,
                     float* pd_prob, float* sm_prob, int32_t* pd_coords, int32_t* sm_coords, float* losses);

/* -- multi-scale test-time evaluation (the caller of the tower, main.py:326-425) ---------------------
 * One pad-or-crop window per output, then skimage.transform.resize(window, [OH,OW]) with the
 * 0.13.x defaults the reference relies on (bilinear, half-pixel centres, zeros outside, clip to
 * the window's [min,max]): get_different_scales (main.py:326-348) and scale_hm_back (:351-379).
 * src [nsrc,H,W,C] device fp32; windows HOST int32 [NW][5] = (source index, y0, x0, h, w), a
 * window may extend beyond the image (= np.lib.pad with zeros); out [NW,OH,OW,C] device fp32. */
int jcm_window_resize(jcm_handle h, const float* src, int nsrc, int H, int W, int C, const int32_t* windows, int NW,
                      int OH, int OW, float* out);
/* np.average over the G scale copies of each image (main.py:413-414): in [n*G, M] -> out [n, M]. */
int jcm_group_mean(jcm_handle h, const float* in, int n, int G, int64_t M, float* out);

/* -- tower concat across processes (main.py:573-574: tf.concat of the per-tower maps; here one process per GPU) --------
 * The only collective of the inference path: every rank contributes its [B_local,2,K] int32 coordinates and receives
 * all ranks' in rank order, moved by RCCL (the ROCm build of the NCCL API) over xGMI.  RCCL is resolved with dlopen on
 * first use (inside a torch process that is the copy torch loaded).
 *   jcm_comm_unique_id : rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId); the host hands it to the other ranks
 *                        by whatever channel it has (the Python host uses the torch.distributed store);
 *   jcm_comm_create    : ncclCommInitRank on `device` (collective over all ranks);
 *   jcm_allgather_coords: ncclAllGather on the handle's stream, then synchronises that stream -- the one entry point of
 *                        the path that does; local [B_local,2,K], all_out [world*B_local,2,K], both device int32. */
#define JCM_COMM_ID_BYTES 128
typedef struct jcm_comm_s* jcm_comm;
int jcm_comm_unique_id(unsigned char* id);
int jcm_comm_create(const unsigned char* id, int world, int rank, int device, jcm_comm* out);
int jcm_comm_destroy(jcm_comm c);
int jcm_allgather_coords(jcm_handle h, jcm_comm c, const int32_t* local, int B_local, int32_t* all_out);

/* CRC-32C (Castagnoli) of host memory, continuing from `crc` (0 to start): the checksum of tf.train.Saver checkpoint
 * files (tf_checkpoint.py reads and writes them; main.py:604,612,666).  Host-only helper, no device work. */
uint32_t jcm_crc32c(const void* data, size_t n, uint32_t crc);

/* -- introspection (used by bench.py for the roofline object) ------------------------------------
 * Sum of the HIP-event durations (ms) and the number of launches recorded for conv layer
 * `scope` since the last read; synchronises the stream and clears the record. */
int jcm_profile_read(jcm_handle h, const char* scope, double* total_ms, int* launches);
/* Name of the HIP kernel a launch of conv layer `scope` on a [B,H,W,Cin] input takes on this handle (the
 * dispatch depends on precision, options and shape); bench.py labels its roofline object with it and the
 * tests assert that the intended kernel is the one that runs. */
int jcm_conv_kernel_name(jcm_handle h, const char* scope, int B, int H, int W, char* name, int cap);
/* Bytes currently held by the workspace arena + packed parameters. */
int64_t jcm_workspace_bytes(jcm_handle h);

/* -- joint training step (main.py:511-577,644) ------------------------------------
 * One `sess.run(train_step, {flag_train: True})` of one tower, split at the point where the
 * reference averages the tower gradients (average_gradients, main.py:243-267) so that the host
 * can all-reduce between the two calls:
 *
 *   jcm_train_loss_grads : forward with batch-statistics BatchNorm (is_training=True, main.py:113,129;
 *       the moving_mean / moving_variance update ops of main.py:557 run here, decay 0.9), loss_tower =
 *       CE(pd) + CE(sm) + lmbd * weight_decay('weights') (main.py:538-540), and opt.compute_gradients
 *       (main.py:560) into one flat caller-owned buffer laid out by jcm_train_param_info.
 *   jcm_train_apply      : grad_renorm(., clip_norm) = tf.clip_by_global_norm (main.py:302-309,576) and
 *       opt.apply_gradients (main.py:577) with tf.train.AdamOptimizer (beta 0.9/0.999, eps 1e-8) or
 *       MomentumOptimizer(0.9) (main.py:501-504); then every derived table (packed weights, folded BN,
 *       prior spectra) is rebuilt, so inference entry points see the new parameters.
 *
 * Trainable tensors are all parameters except the BatchNorm moving statistics, in ascending name
 * order.  fp32 handles take the default route of "conv9_fft" (forward, data and weight gradients in the frequency domain, two scaled
 * fp16 parts per operand) or, with "conv9_fft" = 0 / "f32_conv", the direct kernels; bf16 handles train in mixed precision (bf16 tensors
 * and MFMA operands, fp32 master weights, statistics, losses, spatial model and optimizer). */
#define JCM_OPT_ADAM 0
#define JCM_OPT_MOMENTUM 1
int jcm_train_begin(jcm_handle h);                       /* after jcm_finalize: allocates optimizer slots, n_iters = 0 */
int jcm_train_param_count(jcm_handle h, int64_t* n_tensors, int64_t* n_elements);
int jcm_train_param_info(jcm_handle h, int64_t index, char* name, int name_cap, int64_t* offset, int64_t* count);
/* x [B,H,W,3], y = y_in [B,60,90,K+1] target heat maps (main.py:488); grads: device fp32 [n_elements];
 * losses: device fp32 [4] = loss_tower, loss_pd, loss_sm, weight_decay('weights'). */
int jcm_train_loss_grads(jcm_handle h, const float* x, const float* y, int B, int H, int W, int use_sm, float lmbd,
                         float* grads, float* losses);
/* The gradient kernels of one stride-1 conv layer on caller tensors, on the route the training step takes on this handle (what
 * opt.compute_gradients, main.py:560, evaluates for tf.nn.conv2d, main.py:135): x [B,H,W,Cin] the layer input, dz [B,H,W,Cout] the gradient
 * w.r.t. the convolution output;  grads[<scope>/weights] = sum over (b,y,x) of x (*) dz + lmbd * w  (flat buffer in the layout of
 * jcm_train_param_info, only this slice is written) and dx_out [B,H,W,Cin] (may be NULL) = conv_SAME(dz, flipped transposed w).  Used by the
 * tests to hold the gradient kernels to fp32-class error at full-size layer shapes; fp32 handles, after jcm_train_begin. */
int jcm_train_layer_grads(jcm_handle h, const char* scope, const float* x, const float* dz, int B, int H, int W, float lmbd,
                          float* grads, float* dx_out);
/* grads: the (tower-averaged) gradients, same layout; lr: the value of lr_tf for this update
 * (main.py:492); clip_norm <= 0 disables the clip; grad_norm_out (host, may be NULL) receives the
 * global norm before clipping and makes the call synchronise. */
int jcm_train_apply(jcm_handle h, const float* grads, int optimizer, float lr, float clip_norm, float* grad_norm_out);
/* Overlapping the tower average with the backward pass: `fn(user, offset, count)` is called on the calling thread, from
 * inside jcm_train_loss_grads, as soon as every kernel that writes grads[offset, offset+count) has been enqueued on
 * the handle's stream (one call per layer, last layer first; the spatial-model blocks first of all).  The host
 * records an event on that stream and starts the all-reduce of the range on another stream.  Every trainable
 * element is reported exactly once per call.  NULL disables.
 * The callback runs WITHOUT the library's per-device call lock: it may call read-only entry points (jcm_get_tensor,
 * jcm_profile_read, jcm_last_error, another handle's calls).  It must not start a second training or forward call
 * on the SAME handle (that call would reuse the workspace the running step lives in): every entry point of the same handle
 * that uses the workspace arena or changes the training state or the parameters (jcm_forward, jcm_pd_forward, jcm_conv_layer*,
 * jcm_sm_forward, jcm_conv_mrf, jcm_train_*, jcm_update_tensor) returns JCM_ERR_STATE when called from the callback. */
typedef void (*jcm_grad_ready_fn)(void* user, int64_t offset, int64_t count);
int jcm_train_set_grad_callback(jcm_handle h, jcm_grad_ready_fn fn, void* user);
int jcm_train_steps(jcm_handle h, int64_t* n_iters);     /* n_iters_tf (main.py:491) */
/* The optimizer side of Saver.save / Saver.restore (main.py:604,612,666 cover every global variable): slot 0 = the
 * '<var>/Adam' first moments (or '<var>/Momentum' accumulators), slot 1 = the '<var>/Adam_1' second moments, flat in the
 * layout of jcm_train_param_info (host or device pointer, may be NULL to move n_iters only); n_iters also fixes Adam's
 * beta powers (beta^n_iters) and the position in the learning-rate schedule. */
int jcm_train_get_state(jcm_handle h, int slot, float* out, int64_t count, int64_t* n_iters);
int jcm_train_set_state(jcm_handle h, int slot, const float* data, int64_t count, int64_t n_iters);
/* Saver.save side (main.py:666): copy a stored parameter out (host or device pointer). */
int jcm_get_tensor(jcm_handle h, const char* name, float* out, int64_t count);
/* Saver.restore on a live session (main.py:612): overwrite a stored parameter after jcm_finalize
 * (same element count; host or device pointer).  refresh != 0 rebuilds the derived tables (packed
 * weights, folded BN, prior spectra); pass 0 on all but the last tensor of a batch of updates. */
int jcm_update_tensor(jcm_handle h, const char* name, const float* data, int64_t count, int refresh);

#ifdef __cplusplus
}
#endif
#endif /* JCM_H */
