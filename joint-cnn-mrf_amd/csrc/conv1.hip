// conv1_{fullres,halfres,quarterres}: 5x5 stride-2 SAME convolution 3 -> Cout of the RGB image
// (main.py:44,52,61), fused with the branch's down-sampling resize (main.py:51,60), bias,
// ReLU and inference BatchNorm (main.py:160-163).
//
// K = 75 is too thin for MFMA, and the layer is 0.27 % of the path's FLOPs, so this is a
// direct VALU kernel: a workgroup owns an 8x8 output patch; wave w owns output channels
// [16w, 16w+16) for all 64 pixels, so the 75x16 filter slice is wave-uniform (scalar loads)
// and each lane streams its 5x5x3 window from an LDS copy of the 19x19x3 input patch.
//
// TF SAME with k=5, s=2 on an even extent pads 1 before / 2 after:
// out = ceil(in/2), total = (out-1)*2+5-in, before = total/2.
#include "kernels.h"

namespace jcm {

constexpr int C1_PT = 8;                      // output patch edge
constexpr int C1_IN = 2 * (C1_PT - 1) + 5;    // 19 input rows/cols per patch

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <class TO>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, TO* __restrict__ out,
                                                    int H0, int W0, int sub, int Hin, int Win, int Ho, int Wo,
                                                    int pad_t, int pad_l, int Cout) {
  __shared__ float patch[C1_IN * C1_IN * 3];
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * C1_PT, ox0 = blockIdx.x * C1_PT;
  const int tid = threadIdx.x;
  const float* xb = x + (size_t)b * H0 * W0 * 3;
  for (int i = tid; i < C1_IN * C1_IN * 3; i += blockDim.x) {
    const int c = i % 3;
    const int p = i / 3;
    const int iy = p / C1_IN, ix = p % C1_IN;
    const int gy = oy0 * 2 - pad_t + iy, gx = ox0 * 2 - pad_l + ix;   // coordinates in the sub-sampled image
    float v = 0.f;
    if ((unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win)
      v = xb[((size_t)(gy * sub) * W0 + gx * sub) * 3 + c];
    patch[i] = v;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id = 16-channel group
  const int co0 = cg * 16;
  if (co0 >= Cout) return;
  const int py = lane >> 3, px = lane & 7;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const float* wg = w + co0;
  for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xv = patch[((py * 2 + ky) * C1_IN + px * 2 + kx) * 3 + c];
        const float* wk = wg + (size_t)((ky * 5 + kx) * 3 + c) * Cout;   // wave-uniform address
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(xv, wk[j], acc[j]);
      }
    }
  }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < Ho && ox < Wo) {
    TO* o = out + (((size_t)b * Ho + oy) * Wo + ox) * Cout + co0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      float vp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int co = co0 + j + t;
        vp[t] = fmaxf(acc[j + t] + bias[co], 0.f) * scale[co] + shift[co];
      }
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(o + j) = make_float4(vp[0], vp[1], vp[2], vp[3]);
      } else {
        bf16x4 v;
        v[0] = (__bf16)vp[0]; v[1] = (__bf16)vp[1]; v[2] = (__bf16)vp[2]; v[3] = (__bf16)vp[3];
        *reinterpret_cast<bf16x4*>(o + j) = v;
      }
    }
  }
}

hipError_t conv1_5x5s2(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                       void* out, bool out_bf16, int B, int H0, int W0, int sub, int Cout, hipStream_t st) {
  if (Cout % 16 != 0 || Cout > 64 || H0 % sub != 0 || W0 % sub != 0) return hipErrorInvalidValue;
  const int Hin = H0 / sub, Win = W0 / sub;
  const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
  const int tot_h = (Ho - 1) * 2 + 5 - Hin, tot_w = (Wo - 1) * 2 + 5 - Win;
  const int pad_t = (tot_h > 0 ? tot_h : 0) / 2, pad_l = (tot_w > 0 ? tot_w : 0) / 2;
  dim3 grid((Wo + C1_PT - 1) / C1_PT, (Ho + C1_PT - 1) / C1_PT, B);
  if (out_bf16)
    hipLaunchKernelGGL(conv1_kernel<__bf16>, grid, dim3(64 * (Cout / 16)), 0, st, x, w, bias, scale, shift,
                       static_cast<__bf16*>(out), H0, W0, sub, Hin, Win, Ho, Wo, pad_t, pad_l, Cout);
  else
    hipLaunchKernelGGL(conv1_kernel<float>, grid, dim3(64 * (Cout / 16)), 0, st, x, w, bias, scale, shift,
                       static_cast<float*>(out), H0, W0, sub, Hin, Win, Ho, Wo, pad_t, pad_l, Cout);
  return hipGetLastError();
}

}  // namespace jcm
