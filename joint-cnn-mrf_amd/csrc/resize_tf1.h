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
}  // namespace jcm
