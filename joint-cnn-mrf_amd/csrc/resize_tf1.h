// TF-1.x ResizeBilinear taps (align_corners=False), shared by glue.hip and conv_fft.hip: scale = in / float(out); src = i * scale
// (float32, rounded product as TF computes it); lo = floor(src); hi = min(lo + 1, in - 1); lerp = src - lo.  Lerp along x, then along y.
#pragma once
#include <hip/hip_runtime.h>

namespace jcm {
struct Tap { int lo, hi; float t; };
__device__ __forceinline__ Tap tf1_tap(int i, int in_size, float scale) {
  const float src = __fmul_rn((float)i, scale);
  Tap r;
  r.lo = (int)floorf(src);
  r.hi = min(r.lo + 1, in_size - 1);
  r.t = src - (float)r.lo;
  return r;
}
__device__ __forceinline__ float lerp2(float tl, float tr, float bl, float br, float tx, float ty) {
  const float top = tl + (tr - tl) * tx;
  const float bot = bl + (br - bl) * tx;
  return top + (bot - top) * ty;
}
// x / 3 (the branch merge of main.py:69-70), correctly rounded for every x with a normal quotient: q0 = RN(x c), r = x - 3 q0 (exact in one FMA),
// q = RN(q0 + r c) with c = RN(1/3) -- checked against the exact quotient for all 2^23 mantissas of a binade.  Three instructions instead of the ~10 of
// the IEEE division sequence the compiler emits for `/ 3.0f`, in kernels whose issue slots are the bound (the merge is 8-16 divisions per item).
__device__ __forceinline__ float div3(float x) {
  const float c = 0.333333343267440796f;
  const float q = x * c;
  return fmaf(fmaf(-3.0f, q, x), c, q);
}
}  // namespace jcm
