// Internal launcher prototypes of libjcm (gfx950 only).  Every launcher enqueues on `st`
// and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace jcm {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: launchers that need more than
// 64 KB of LDS call ensure() before every launch; the attribute is set once per (instantiation, current device).
// Two threads racing on the same device both set the same value, which is harmless.
struct LdsAttr {
  std::atomic<uint64_t> done{0};
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// ---- conv_igemm.hip : implicit-GEMM stride-1 SAME convolution on MFMA ----------------------
// x NHWC [B,H,W,Cin] (Cin % 16 == 0), packed weights [k*k][Cin/4][CoutP][4] (fp32) or
// [k*k][Cin/8][CoutP][8] (bf16), epilogue relu_bn ? max(z+bias,0)*scale+shift : z+bias.
struct ConvArgs {
  const void* x;
  const void* wp;
  const float* bias;
  const float* scale;
  const float* shift;
  void* out;
  int B, H, W, Cin, Cout, CoutP;
  int relu_bn;
  // conv_split fp16x3 on gradients: device pair {S, 1/S}; the input is multiplied by S (a power of two that lifts it
  // into the fp16 range) before it is split and the result by 1/S.  Null = no scaling.
  const float* in_scale = nullptr;
  // fp16x3 weights: device pair {Sw, 1/Sw} the weights were multiplied by at pack time (pack_weights_split(ns = 2)); null = 2^12
  const float* w_scale = nullptr;
  // bf16 kernels: activation layout of the input / output tensor.  0 = NHWC (the boundary layout).  1 = "planar"
  // [B][C/8][H*W][8]: the 8 channels of a 16-byte MFMA operand unit stay together and every unit is its own image plane,
  // so a halo row of one unit is contiguous in HBM (a 64-pixel LDS-DMA reads 1 KB instead of 64 separate 128-byte
  // lines).  The 9x9 chain conv3 -> conv4 -> merge -> conv5 -> conv6 of the bf16 path runs planar.
  int in_planar = 0, out_planar = 0;
  // conv_fft only -- OVERLAP-SAVE windows (the training step, jcm_train.hip): the input is a batch of H x W windows that FILL the circular transform
  // (H, W are transform lengths; each window carries a halo of 4 real pixels around its (H - 8) x (W - 8) valid region), the output is the batch of valid
  // regions [B, H - 8, W - 8, Cout]: row j of it is row j + 4 + (k - 1) / 2 of the circular convolution.  The filter spectra shrink with the window
  // (32 x 32: 544 frequencies instead of the 3136 of a 60 x 90 map), which is what pays when the batch is small.
  int circ = 0;
  // ... whose windows need not exist as a tensor (round 6): win_map = the [win_B, win_H, win_W, Cin] fp32 map they are cut from (x is then unused) -- window
  // (b, ty, tx) of the win_TY x win_TX grid covers map rows ty (H - 8) - 4 .. + H, columns tx (W - 8) - 4 .. + W, zeros outside the map (and in the halo of 4
  // when win_valid_only: the weight gradient's dZ).  The forward row pass gathers while it loads (conv_fft_win_gather_supported()).
  const void* win_map = nullptr;
  int win_B = 0, win_H = 0, win_W = 0, win_TY = 0, win_TX = 0, win_valid_only = 0;
  // ... and the valid regions need not either: wout_TY > 0 -> `out` is the [win_B, wout_H, wout_W, Cout] MAP and the inverse row pass stores valid pixel (y, x)
  // of window (b, ty, tx) at map pixel (ty (H - 8) + y, tx (W - 8) + x) where that lies inside the map (conv_fft_win_scatter_supported())
  int wout_H = 0, wout_W = 0, wout_TY = 0, wout_TX = 0;
  // conv_fft, bf16 handles with 16-bit row-transformed tensors: the 96-point inverse row pass with planar bf16 output (conv5 of the model) as a matrix product
  // on the matrix cores (conv_fft_rows_mfma.hip).  1 = on where the kernel exists, 0 = the register kernel.
  int rows_mfma = 0;
  // conv5_strip_bf16 only: the LEFT HALF of the 2x2/2 max pool that follows the layer in its epilogue -- `out` is the [B, H, W / 2, Cout] map of
  // max(pixel 2 i, pixel 2 i + 1) (W even; either activation layout), half the bytes; vpool_2x1() finishes the pool.  The max of two bf16-rounded values is
  // the bf16 rounding of the max: bit-identical to pooling the stored map.
  int hpool = 0;
};
int conv_igemm_bn(int Cout);                     // N-tile the dispatcher will use for this Cout
hipError_t conv_igemm_f32(const ConvArgs& a, int ks, hipStream_t st);
hipError_t pack_weights_f32(const float* w_hwio, float* wp, int ks, int Cin, int Cout, int CoutP, hipStream_t st);

// ---- conv_split.hip : fp32 9x9 convolution on the bf16 matrix cores (three-way operand split, six products) --
// same contract as conv_igemm_f32 (fp32 in / out, same epilogue) for the shapes conv_split_supported() accepts
bool conv_split_supported(int ks, int Cin, int CoutP, int B, int H, int W, int min_wgs);   // 5x5 / 9x9, Cin % 16 == 0, CoutP % 128 == 0; 12x32 patches or whole-row tiles
size_t conv_split_weight_bytes(int ks, int Cin, int CoutP, int ns);     // ns: 3 = bf16x6, 2 = fp16x3
hipError_t pack_weights_split(const float* w_hwio, void* wp, int ks, int Cin, int Cout, int CoutP, int ns, hipStream_t st,
                              const float* w_scale = nullptr);     // ns = 2: weights are stored times w_scale[0] (device; null = 2^12)
hipError_t conv_split_f32(const ConvArgs& a, int ks, int ns, hipStream_t st);

// conv_thin_split.hip: the 9-channel logits layer as fp16x3 (weights from pack_weights_split(ns = 2) with CoutP = 16)
hipError_t conv_thin_split16(const ConvArgs& a, hipStream_t st);

// ---- conv_thin_f32.hip : 9x9 conv with Cout <= 12 (the logits layer) on v_mfma_f32_4x4x1_16b_f32 ----
// weights packed by pack_weights_f32 with CoutP = 16
hipError_t conv_thin_f32(const ConvArgs& a, hipStream_t st);

// ---- conv_thin_bf16.hip : the logits layer on v_mfma_f32_16x16x32_bf16 (bf16 in, fp32 out, Cout <= 16) ----
// weights packed by pack_weights_bf16 with CoutP = 16
hipError_t conv_thin_bf16(const ConvArgs& a, hipStream_t st);

// ---- conv_fft.hip : stride-1 SAME convolution (9x9, 5x5) in the frequency domain: in-LDS FFTs (rows, then columns with the operand split
// fused in) around the channel GEMM of cgemm_split.hip, one complex matrix product per frequency.  a.wp = split filter spectra of this map
// and kernel size (conv_fft_pack_weights).  np = operand form of the channel GEMM: 5 = ONE fp16 part of spectra scaled by powers of two, one product, 32 channels per
// stage (bf16 handles, default: the tensors on either side of the layer are bf16, an 11-bit spectrum is 8x finer); 2 = two bf16 parts (bf16 handles), 3 = three bf16 parts / six
// products, 4 = two FP16 parts / three products of spectra scaled by powers of two (fp32 handles; both fp32-class, 4 is the default).
// Shapes: Cin % 64 == 0, H + k - 1 <= 192, W + k - 1 <= 192.
//
// np = 4 / 5 scaling (fp16 carries 11 bits over 2^-24 .. 2^16): device words, all written and read on the stream --
//   tmax[b]      : max |T| of image b of the layer's row-transformed input (atomic max by the row pass, or by the previous layer's fused kernel:
//                  ZERO them before the producer runs); the column pass scales image b by 2^k with H * tmax[b] * 2^k < 2^15 (components of the scaled spectra then stay below sqrt(2) * 2^15 < 65504: conv_fft_common.h), the inverse row
//                  pass undoes it.  One word per IMAGE: a row of the channel GEMM is one image, so an image's result does not depend on the
//                  batch it is in.  common = 1: one scale for the tensor, from the max over the B words (the training step: the weight gradient
//                  sums over the images and needs one scale);
//   tmax_next[b] : the words of the NEXT layer when t_next is given;
//   winv         : 1 / (scale of the filter spectra), written by conv_fft_pack_weights into wscale[1] (wscale[0] is its scratch).
struct Fp16Scale {
  float* tmax = nullptr;
  float* tmax_next = nullptr;
  const float* winv = nullptr;
  float hf = 0.f;      // H (set by conv_fft_f32)
  int nb = 0;          // B (set by conv_fft_f32)
  int common = 0;
  // np = 5 with 16-bit row-transformed tensors: t16 = 1 asks conv_fft_f32 for them; it carves the tile scale words out of its workspace and fills the
  // pointers below for its kernels (T between the forward row and column pass, T' between the inverse column and row pass: complex fp16 in block
  // floating point, conv_fft_common.h; t16_cb = channels per tile of the inverse column pass).  bf16 in / out layouts only.
  int t16 = 0;
  float* t16_fwd = nullptr;
  float* t16_inv = nullptr;
  int t16_cb = 64;
};
bool conv_fft_supported(const ConvArgs& a, int ks);
size_t conv_fft_weight_bytes(int H, int W, int ks, int Cin, int Cout, int np, int circ = 0);      // circ: H x W is the window = the transform (ConvArgs::circ)
hipError_t conv_fft_pack_weights(const float* w_hwio, void* wf, int H, int W, int ks, int Cin, int Cout, int np, bool round_bf16, hipStream_t st,
                                 float* wscale = nullptr, int circ = 0,
                                 const float* bound_from = nullptr);      // bound_from: device word holding the filter's bound already (wscale[0] of the same filter's other spectra)
size_t conv_fft_workspace_bytes(const ConvArgs& a, int ks, int np);
// in / out layout: 0 = fp32 NHWC, 1 = bf16 NHWC, 2 = bf16 planar (bf16 handles: operands are bf16 values, the arithmetic is fp32-class); g0/g1: optional events around the GEMM
// t_in / t_next: the row-transformed tensor handed from one frequency-domain layer to the next (fp32 NHWC handles; conv_fft_fusable).
// merge: a.x is the full-resolution branch x1 and the layer's input is ((x1 + up(x2)) + up(x3)) / 3, formed while the rows are loaded.
struct FftMerge { const void* x2; int H2, W2; const void* x3; int H3, W3; };      // x1 (a.x), x2, x3: NHWC in the layer's input type
// What lies between this layer and the one t_next is for (fp32 handles, conv_fft_rows_fused.hip): pool = the 2x2/2 SAME max pool (t_next is the
// row-transformed POOLED map, for a next layer of kernel size ks_next); merge = this layer is the full-resolution branch and t_next the row-transformed
// merged map ((x1 + up(x2)) + up(x3)) / 3.  Neither: t_next is this layer's own output (same map).
struct FftNext { int pool = 0; int ks_next = 0; const FftMerge* merge = nullptr; };
// xs / xs_ready: keep the split activation spectra in a caller buffer (conv_fft_xs_bytes) / they are there already (skip the forward transforms)
hipError_t conv_fft_f32(const ConvArgs& a, int ks, int np, int in_layout, int out_layout, void* work, const void* t_in, void* t_next, const FftMerge* merge,
                        hipEvent_t g0, hipEvent_t g1, hipStream_t st, void* xs = nullptr, bool xs_ready = false, const Fp16Scale* sc = nullptr, const FftNext* nx = nullptr);
// the fused hand-overs of FftNext: is there a kernel for this pair of layers, and the size of the row-transformed tensor handed over
bool conv_fft_win_gather_supported(int win, int Cin);
bool conv_fft_win_scatter_supported(int win, int Cout);    // ... and its inverse row pass store the valid regions straight into the map?      // can the forward row pass of `win` x `win` overlap-save windows read them straight from the map?
bool conv_fft_pool_fusable(const ConvArgs& a, int ks, int ks_next);
size_t conv_fft_pool_handover_bytes(const ConvArgs& a, int ks_next);
bool conv_fft_merge_fusable(const ConvArgs& a, int ks, int ks_next, const FftMerge& m, bool h16 = false);      // h16: bf16 handles (16-bit T / T', bf16 branches)
size_t conv_fft_xs_bytes(const ConvArgs& a, int ks, int np);
// NHWC fp32 -> split spectra (the two forward passes); np = 4: tmax = the (zeroed) device word of this tensor
hipError_t conv_fft_spectra(const ConvArgs& a, int ks, int np, void* work, void* xs, hipStream_t st, float* tmax = nullptr, int common = 0);
bool conv_fft_geometry(int H, int W, int ks, int B, int Cout, int np, int* NY, int* NX, int* MT, int circ = 0);

// ---- wgrad_fft.hip : weight gradient of a stride-1 layer in the frequency domain (fp32 handles, training step): per frequency
// P[f][ci][co] = sum_b conj(X[f][b][ci]) dZ[f][b][co] on v_mfma_f32_32x32x16_bf16 from the split spectra of the layer input (kept by the forward pass)
// and of dZ, then the k x k taps are read off the inverse transform:  dw = taps(P) / (NY NX) + lmbd * w.   xs / zs: np = 3 layouts of cgemm_split.hip
size_t wgrad_fft_scratch_bytes(int NY, int NX, int Cin, int Cout);
// ldz: channels of the dZ spectra (P's columns, scratch sized with it) >= Cout, the filter's
// np = 3: three bf16 parts per operand; np = 4: two fp16 parts, scaled (common = 1) -- tmax_x / tmax_z = the B device words of the two spectra, H = map height
hipError_t wgrad_fft(const void* xs, const void* zs, void* scratch, const float* w, float lmbd, float* dw, int ks, int NY, int NX, int B, int MTx, int MTz, int Cin,
                     int ldz, int Cout, hipStream_t st, int np = 3, const float* tmax_x = nullptr, const float* tmax_z = nullptr, int H = 0);
hipError_t pad_channels_f32(const float* in, int ldi, float* out, int ldo, size_t N, hipStream_t st);      // train_kernels.hip
// overlap-save windows of an NHWC fp32 map (train_kernels.hip): win [B * TY * TX][WS][WS][C], window (ty, tx) = rows ty V - 4 .. + WS - 1, V = WS - 8,
// zeros outside the map; valid_only: zeros in the 4-pixel halo ring as well (the gradient of a layer's output, where every pixel must count once).
// scatter: the valid regions [B * TY * TX][V][V][C] back into the map
hipError_t window_gather_f32(const float* map, float* win, int B, int H, int W, int C, int WS, int TY, int TX, int valid_only, hipStream_t st);
hipError_t window_scatter_f32(const float* val, float* map, int B, int H, int W, int C, int WS, int TY, int TX, hipStream_t st);
bool conv_fft_fusable(const ConvArgs& a, int ks, int ks_next);
size_t conv_fft_handover_bytes(const ConvArgs& a, int ks);

// ---- cgemm_split.hip : Y[f][b][co] = sum_ci X[f][b][ci] W[f][ci][co] (complex) for F frequencies on v_mfma_f32_32x32x16_bf16, operands
// as np bf16 parts in tile-major LDS-image layout:
//   xs[f][m-tile][Cin/16][re|im][part][k-half][MT rows][8 bf16]   (MT = cgemm_split_mtile(np, B, Cout); rows >= B are never stored)
//   ws[f][n-tile][Cin/16][re|im][part][k-half][ntl cols][8 x 16 bit];   ntl = cgemm_split_ntile(np, Cout): 128, or 32 for Cout <= 32 on fp32 handles
//   y [f][B][ldy] complex fp32, ldy >= Cout rounded up to whole N tiles; with y16_shift != 0 (np = 5): complex FP16 = product * y16_shift, the
//   constant power of two cgemm_split_y16_shift(Cin) under which no component can overflow (the operands' scales bound the products)
int cgemm_split_mtile(int np, int B, int Cout);
int cgemm_split_ntile(int np, int Cout);
int cgemm_split_parts(int np);      // 16-bit parts per operand: 2 (np = 2: bf16, np = 4: fp16) or 1 (np = 5)
size_t cgemm_split_w_bytes(int np, int F, int Cin, int Cout);
float cgemm_split_y16_shift(int Cin);
hipError_t cgemm_split(const void* xs, const void* ws, void* y, int np, int F, int B, int Cin, int Cout, int ldy, hipStream_t st, float y16_shift = 0.f);

// ---- conv5_strip_bf16.hip : 5x5 SAME convolution in 128-channel output tiles on 768-pixel strips (bf16 in NHWC or planar, bf16 out NHWC or planar;
// weights as packed by pack_weights_bf16); shapes: CoutP % 128 == 0, Cin % 32 == 0, Cout % 8 == 0, 8 <= W <= 191 and a window of at most 64 row parts
bool conv5_strip_bf16_supported(const ConvArgs& a, int ks);
hipError_t conv5_strip_bf16(const ConvArgs& a, hipStream_t st);

// ---- conv_kxfold_bf16.hip : the 9-channel logits layer with the kernel columns folded into the MFMA's N axis (81 -> 96
// columns, one 1x1-conv GEMM per kernel row, shifted sum over kx through LDS); bf16 in (NHWC or planar), fp32 NHWC out.
// Weights packed by pack_weights_kxfold: [Cin/16][ky][unit][96][8].  Shapes: Cout == 9, Cin % 32 == 0, 82 <= W <= 90.
bool conv_kxfold_bf16_supported(const ConvArgs& a, int ks);
size_t conv_kxfold_weight_bytes(int Cin);
hipError_t pack_weights_kxfold(const float* w_hwio, void* wp, int Cin, hipStream_t st);
hipError_t conv_kxfold_bf16(const ConvArgs& a, hipStream_t st);

// ---- conv_igemm_bf16.hip : the same dataflow on v_mfma_f32_32x32x16_bf16 ---------------------
// x bf16 NHWC (Cin % 32 == 0), packed weights bf16 [k*k][Cin/8][CoutP][8], out bf16 (fp32 when
// out_f32: the logits layer).
int conv_igemm_bf16_bn(int Cout, int ks);
hipError_t conv_igemm_bf16(const ConvArgs& a, int ks, bool out_f32, hipStream_t st);
hipError_t pack_weights_bf16(const float* w_hwio, void* wp, int ks, int Cin, int Cout, int CoutP, hipStream_t st);

// ---- conv_strip_bf16.hip : the 9x9 layers with M flattened over the batch (384-pixel strips, no padded slots),
// halo and weights by LDS-DMA rings.  Same operands as conv_igemm_bf16 (bf16 NHWC in / out, packed weights
// [81][Cin/8][CoutP][8]); CoutP % 256 == 0, Cin % 16 == 0, H*W >= 384.  conv_igemm_bf16 dispatches to it.
bool conv_strip_bf16_supported(const ConvArgs& a, int ks);
hipError_t conv_strip_bf16(const ConvArgs& a, hipStream_t st);

// ---- conv1.hip : 5x5 stride-2 SAME convolution of the (sub-sampled) RGB image ---------------
// x [B,H0,W0,3] fp32; the branch input is x[:, ::sub, ::sub] (TF-1.x bilinear with an integer
// scale is pure sub-sampling, main.py:51,60); w HWIO [5,5,3,Cout]; out [B,Ho,Wo,Cout] fp32 or bf16.
hipError_t conv1_5x5s2(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                       void* out, bool out_bf16, int B, int H0, int W0, int sub, int Cout, hipStream_t st);

// ---- conv1_mfma.hip : conv1 + bias/ReLU/BN + 2x2 max-pool fused, bf16 MFMA (bf16 path only) ----
// x [B,H0,W0,3] fp32 (branch input = x[:, ::sub, ::sub]) -> out [B,(H0/sub)/4,(W0/sub)/4,64] bf16
hipError_t pack_conv1_bf16(const float* w_hwio, void* wq, hipStream_t st);
hipError_t conv1_mfma_pool(const float* x, const void* wq, const float* bias, const float* scale, const float* shift,
                           void* out, int B, int H0, int W0, int sub, hipStream_t st);

// the same fusion on the exact fp32 path (v_mfma_f32_32x32x2_f32): wq [5][16][64] fp32 from pack_conv1_f32, out fp32
hipError_t pack_conv1_f32(const float* w_hwio, float* wq, hipStream_t st);
// the same with fp32 operands as three bf16 parts on the bf16 matrix cores (fp32 handles on the default frequency-domain route)
size_t conv1_split_weight_bytes();
hipError_t pack_conv1_split(const float* w_hwio, void* wq, hipStream_t st);
hipError_t conv1_mfma_pool_split(const float* x, const void* wq, const float* bias, const float* scale, const float* shift, float* out,
                                 int B, int H0, int W0, int sub, hipStream_t st);
hipError_t conv1_mfma_pool_f32(const float* x, const float* wq, const float* bias, const float* scale, const float* shift, float* out,
                               int B, int H0, int W0, int sub, hipStream_t st);

// ---- glue.hip ----------------------------------------------------------------------------------
// `bf16`: activations are bf16 instead of fp32 (arithmetic stays fp32).
hipError_t max_pool_2x2(const void* x, void* out, bool bf16, int B, int H, int W, int C, hipStream_t st);
// the lower half of the pool behind a conv5_strip_bf16 launch with hpool: out[b][y][x][c] = max(in[b][2 y][x][c], in[b][2 y + 1][x][c]) (bf16, C % 8 == 0)
hipError_t vpool_2x1_bf16(const void* x, void* out, int B, int H, int W, int C, hipStream_t st);
hipError_t resize_bilinear(const float* x, float* out, int B, int H, int W, int C, int OH, int OW, hipStream_t st);
// out = (x1 + resize(x2) + resize(x3)) / 3   (main.py:58,67,69-70); x1 [B,H,W,C]
hipError_t upsample_merge3(const void* x1, const void* x2, int H2, int W2, const void* x3, int H3, int W3,
                           void* out, bool bf16, int B, int H, int W, int C, hipStream_t st);
// the same on planar bf16 tensors ([B][C/8][H*W][8], all four); C % 8 == 0
hipError_t upsample_merge3_planar(const void* x1, const void* x2, int H2, int W2, const void* x3, int H3, int W3,
                                  void* out, int B, int H, int W, int C, hipStream_t st);
hipError_t spatial_softmax(const float* in, float* out, int B, int HW, int K, hipStream_t st);
hipError_t argmax_coords(const float* hm, int32_t* coords, int B, int HW, int WW, int K, hipStream_t st);
// spatial_softmax followed by the first-occurrence argmax of the probabilities, one pass over the logits, one
// workgroup per image: prob (may be null) [B,HW,K], coords (may be null) [B,2,K].  HW % 4 == 0, K <= 9.
hipError_t softmax_argmax(const float* logits, float* prob, int32_t* coords, int B, int HW, int WW, int K, hipStream_t st);
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale
hipError_t bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale, float* shift,
                   int n, hipStream_t st);

// ---- spatial_model.hip -----------------------------------------------------------------------------
// softplus5 tables of the batch-independent operands (main.py:120,122)
hipError_t sm_softplus5(const float* in, float* out, int64_t n, hipStream_t st);
// out[p][i] = softplus5(in[p][i]) for P separately allocated tensors of n elements each (one launch for the 81 pairs)
hipError_t sm_softplus5_multi(const float* const* in, float* out, int P, int64_t n, hipStream_t st);
// hm [B,5400,C] NHWC -> lik [B][C][5400] planar = softplus5(bn(hm))
// (channels [0,Ca) from hm [B,5400,Ca], the rest from extra [B,5400,C-Ca]: main.py:528's concat read in place;
//  extra_ld: floats per pixel of the tensor `extra` points into -- e.g. K+1 when it is channel K of y_in; 0 = C-Ca)
hipError_t sm_likelihood(const float* hm, int Ca, const float* extra, const float* bn_scale, const float* bn_shift, float* lik,
                         int B, int C, hipStream_t st, int extra_ld = 0);
// cpre[b][p][61*91] = valid true convolution of prior p (120x180, already softplus'd) with
// maps[b][cond[p]] (60x90 planar)           (main.py:83-87)
hipError_t sm_pair_conv(const float* priors, const float* maps, const int* cond, float* cpre,
                        int B, int P, int C, hipStream_t st);
// E[b,pix,j] = log(lik[b][j][pix]+d) + sum_{pairs of j, graph order} log(R(cpre)[pix] + spb[p][pix] + d)
hipError_t sm_finish(const float* lik, const float* cpre, const float* spbias, float* logits,
                     int B, int K, int C, hipStream_t st);
// out[b,pix] = R(cpre[b][0])[pix]   (the resize of main.py:89 on its own, for jcm_conv_mrf)
hipError_t sm_resize_only(const float* cpre, float* out, int B, hipStream_t st);

// ---- sm_frames.hip : plain kernels around the spatial model's 120 x 180 frames (no transforms) ----------
// frame[b][c] = softplus5(bn(hm[b,:,:,c])) in the top-left 60x90 of a zero 120x180 frame (sc null: raw)
// (channels [0,Ca) from hm [B,5400,Ca], the rest from extra [B,5400,C-Ca]: main.py:528's concat read in place)
hipError_t sm_pad_frame(const float* hm, int Ca, const float* extra, const float* sc, const float* sh, float* frame, int B, int C,
                        hipStream_t st, int extra_ld = 0);
// spec[b][p] = lhat[b][cond[p]] * phat[p] / (120*180)   (elementwise: any layout of the 120 * 91 complex numbers of a spectrum)
hipError_t sm_spec_mul(const float2* lhat, const float2* phat, const int* cond, float2* spec, int B, int C, int P, hipStream_t st);
// out[b] = the TF-1.x 61x91 -> 60x90 resize (main.py:89) of the VALID window frame[59.., 89..] of cfull [B][120][180]
hipError_t sm_resize_frame(const float* cfull, float* out, int B, hipStream_t st);
// ---- sm_fused.hip (default): every transform of the pairwise convolutions in LDS -- forward spectra per (image, channel) into
// lhat_t [B][C][91][120], then per (image, joint) product + inverse + window + resize + bias + log + sum.
// tsave (may be null): [B][P][5400] the argument of every pairwise log, kept for the training step's backward pass
hipError_t sm_fused_forward(const float* hm, int Ca, const float* extra, int extra_ld, const float* sc, const float* sh, const float2* phat_t,
                            const int* cond, const float* spbias, float2* lhat_t, float* logits, int B, int K, int C, hipStream_t st,
                            float* tsave, void* scratch, unsigned epoch);
// scratch: sm_fused_scratch_bytes() bytes owned by the handle, ZEROED once (partial sums + flags of the balanced kernel's cuts); epoch: a counter of the
// handle, > 0, incremented for every launch (the flags carry it, so they are never reset)
size_t sm_fused_scratch_bytes();
// the first half alone: lhat_t[b][c] = transposed half spectrum of the frame of channel c of image b (sc null: the raw map)
hipError_t sm_fused_spectra(const float* hm, int Ca, const float* extra, int extra_ld, const float* sc, const float* sh, float2* lhat_t, int B, int C,
                            hipStream_t st);
// ---- sm_lds.hip : whole-frame transforms in LDS (spectra transposed, [n][91][120]); unnormalised
hipError_t sm_lds_fwd_frames(const float* frames, float2* spec_t, int n, hipStream_t st);      // real [n][120][180] -> spectra
// D[b][p] = R^T (G_j / T_p) on the window [59..119] x [89..179] (backward of main.py:89,121) -> its spectrum; dhat_t [nb * P][91][120]
hipError_t sm_lds_fwd_dframes(const float* G, const float* T, float2* dhat_t, int nb, int K, int P, hipStream_t st);
// rows [r0, r0 + nrows) of the real frames [n][120][180] of the spectra, times `scale`
hipError_t sm_lds_inv_frames(const float2* spec_t, float* frames, int n, int r0, int nrows, float scale, hipStream_t st);

// ---- multiscale.hip : crop/pad window + skimage-style bilinear resize, mean over scale copies ------
// windows_dev: int32 [NW][5] = (source image, y0, x0, h, w); mm_scratch: float2 [NW]
hipError_t window_resize(const float* src, int H, int W, int C, const int* windows_dev, int NW, float2* mm_scratch,
                         int OH, int OW, float* out, hipStream_t st);
hipError_t group_mean(const float* in, float* out, int n, int G, size_t M, hipStream_t st);

// ---- train_kernels.hip : training-step kernels other than convolutions (fp32 NHWC) ------------------
size_t train_reduce_scratch_doubles(int C);      // scratch the per-channel reductions below need
// batch mean / 1/sqrt(biased var + eps) of x [N,C]; moving stats (may be null) updated with `decay`
// `bf16`: the activation / gradient tensors are bf16 instead of fp32 (statistics and arithmetic stay fp32)
hipError_t bn_batch_stats(const void* x, bool bf16, size_t N, int C, float eps, float decay, float* mean, float* rstd, float* mov_mean,
                          float* mov_var, double* scratch, hipStream_t st);
hipError_t bn_apply(const void* r, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, bool bf16, size_t N,
                    int C, hipStream_t st);
// sums [2][C] = (sum dy, sum dy*(r-mean)) with dy pre-scaled by dy_scale; dgamma / dbeta may be null
hipError_t bn_bwd_reduce(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, size_t N, int C,
                         float* sums, float* dgamma, float* dbeta, double* scratch, hipStream_t st);
hipError_t bn_bwd_apply(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma,
                        const float* sums, size_t N, int C, int relu, void* dz, hipStream_t st);
hipError_t col_sum(const void* x, bool bf16, size_t N, int C, float* out, double* scratch, hipStream_t st);
// bn_apply + 2x2/2 SAME max pool of its output in one pass (fp32, C % 4 == 0); false: not this case, nothing launched
bool bn_apply_pool(const void* r, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, void* p, bool bf16, int B, int H, int W, int C,
                   hipStream_t st);
// BatchNorm backward (reduce + apply + bias-gradient sums) of a layer whose output goes through the 2x2/2 max pool, from the POOLED gradient dp and the layer's saved
// y: the pool's backward pass is formed in registers.  false: not this case (bf16 / shapes), nothing launched.
bool bn_bwd_pooled(const void* dp, const void* y, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma, int B, int H, int W, int C, float* sums,
                   float* dgamma, float* dbeta, void* dz, float* colsum, double* scratch, hipStream_t st, hipError_t* err);
// bn_bwd_apply followed by col_sum(dz) -- the bias gradient of the convolution in front -- in one pass where the shapes allow (fp32, 256 % (C / 4) == 0)
hipError_t bn_bwd_apply_colsum(const void* dy, float dy_scale, const void* r, bool bf16, const float* mean, const float* rstd, const float* gamma, const float* sums, size_t N,
                               int C, int relu, void* dz, float* colsum, double* scratch, hipStream_t st);
hipError_t max_pool_bwd(const void* x, const void* dy, void* dx, bool bf16, int B, int H, int W, int C, hipStream_t st);
hipError_t cast_pad_bf16(const float* in, int ldi, void* out, int ldo, size_t N, hipStream_t st);
hipError_t cast_bf16_f32(const void* in, float* out, size_t n, hipStream_t st);
// dx [B,h,w,C] = scale * adjoint of the TF-1.x bilinear resize h x w -> H x W applied to dy [B,H,W,C]
hipError_t resize_bilinear_bwd(const void* dy, void* dx, bool bf16, int B, int h, int w, int H, int W, int C, float scale, hipStream_t st);
hipError_t softmax_ce(const float* logits, const float* target, int B, int HW, int K, int Kt, float gscale, float* loss, float* dz,
                      int ldz, int accumulate, hipStream_t st);
hipError_t softmax_bwd(const float* p, const float* g, int B, int HW, int K, int ldg, float* dz, int ldz, hipStream_t st);
hipError_t loss_means_accumulate(const float* ce, int n, float scale, float* out, bool first, hipStream_t st);
hipError_t sum_squares(const float* x, size_t n, double* out, int accumulate, double* scratch, hipStream_t st);
// the same over the flagged chunks of a chunk table (w[k] + start[k], len[k]): one pass over many tensors
hipError_t sum_squares_chunks(float* const* w, const int64_t* start, const int* len, const int* flag, int nchunks, double* out, double* scratch, hipStream_t st);
hipError_t adam_update(float* w, const float* g, float* m, float* v, size_t n, const double* sumsq, float clip, float lr_t, float b1,
                       float b2, float eps, hipStream_t st);
hipError_t momentum_update(float* w, const float* g, float* acc, size_t n, const double* sumsq, float clip, float lr, float mom,
                           hipStream_t st);
hipError_t scale_copy(const float* a, float s, float* out, size_t n, hipStream_t st);
// clip + Adam (or momentum: b1 = momentum, momentum_mode = 1) over a chunk table covering every trainable tensor, one launch
hipError_t optimizer_chunks(float* const* wptr, const int64_t* cstart, const int64_t* coff, const int* clen, int nchunks, const float* g,
                            float* m, float* v, const double* sumsq, float clip, float lr_t, float b1, float b2, float eps, int momentum_mode,
                            hipStream_t st);

// ---- wgrad.hip : weight gradients on MFMA, data-gradient weight transform ------------------------------
int wgrad_splits(int ks, int Cin, int Cout, int B, int H);
hipError_t wgrad_f32(const float* x, const float* dz, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                     hipStream_t st);
// dw = out_scale * sum(partials) + lmbd * w   (out_scale: device scalar, null = 1)
hipError_t wgrad_reduce(const float* partial, int splits, size_t n, const float* w, float lmbd, float* dw, hipStream_t st, const float* out_scale = nullptr);
int wgrad_conv1_blocks(void);
hipError_t wgrad_reduce_wide(const float* partial, int splits, size_t n, const float* w, float lmbd, float* dw, hipStream_t st);      // many tiles of a small tensor (conv1)
hipError_t wgrad_conv1(const float* x, const void* dz, bool dz_bf16, float* partial, int B, int H0, int W0, int sub, int Cout, hipStream_t st);
// wgrad_split.hip: the same on the 16-bit matrix cores (two fp16 parts per operand, three products; or plain bf16 operands), LDS transpose reads
// two fp16 parts of x * S (S = scale[0], a device scalar; null = 1)
hipError_t split_parts16(const float* x, void* out_f16_2n, size_t n, const float* scale, hipStream_t st);
// scale[0] = S = the power of two that brings max|x| just below 2^14, scale[1] = 1/S   (scratch: >= 1024 floats)
hipError_t pow2_scale_of(const float* x, size_t n, float* scale, float* scratch, hipStream_t st);
// fp16x3 weight gradient: x / dz as split_parts16 images
hipError_t wgrad_split16(const void* xp, const void* zp, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                         hipStream_t st);
bool wgrad_split_supported(int ks, int Cin, int ldz);
// bf16 activations / gradients as they are (bf16 training): same kernel with one operand part
hipError_t wgrad_bf16(const void* x, const void* dz, float* partial, int splits, int ks, int B, int H, int W, int Cin, int Cout, int ldz,
                      hipStream_t st);
hipError_t flip_transpose_weights(const float* w_hwio, float* wd, int ks, int Cin, int Cout, int CoP, hipStream_t st);

// ---- sm_train.hip : backward of the spatial model (frequency-domain correlations) -----------------------
hipError_t bn_fold_stats(const float* mean, const float* rstd, const float* gamma, const float* beta, float* sc, float* sh, int n,
                         hipStream_t st);
hipError_t sm_concat_target(const float* prob, const float* y, float* out, size_t N, int K, int C, hipStream_t st);
// G [nb,5400,K] = dL/dE, T [nb][P][5400] saved log arguments
hipError_t sm_bwd_dbias(const float* G, const float* T, float* dspb, int nb, int K, int P, int accumulate, hipStream_t st);
hipError_t sm_bwd_spec_da(const float2* Dhat, const float2* Lhat, const int* cond, float2* dA, int nb, int C, int P, int accumulate,
                          hipStream_t st);
hipError_t sm_bwd_spec_dl(const float2* Dhat, const float2* Ahat, float2* dL, int nb, int K, int C, hipStream_t st);
hipError_t sm_bwd_dh(const float* dLframe, const float* G, const float* frame, const float* hm, const float* sc, const float* sh, float* dh,
                     int nb, int K, int C, hipStream_t st);
hipError_t sm_bwd_params(const float* dAframe, const float* dspb, const float* const* e_ptr, const float* const* b_ptr, const int64_t* e_off,
                         const int64_t* b_off, float* grads, int P, hipStream_t st);

}  // namespace jcm
