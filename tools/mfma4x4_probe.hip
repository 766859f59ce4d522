// Probe of v_mfma_f32_4x4x1_16b_f32 operand / result layout (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  // A[block][i] = 100*block + i + 1 ; B[block][j] = 10*(j+1)  (asymmetric)
  float a = 100.f * (l >> 2) + (l & 3) + 1;
  float b = 10.f * ((l & 3) + 1);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // expectation if D[block][i=reg][j=lane&3] = A[block][i]*B[block][j]
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float e1 = (100.f * (l >> 2) + r + 1) * (10.f * ((l & 3) + 1));          // i = reg, j = lane&3
    float e2 = (100.f * (l >> 2) + (l & 3) + 1) * (10.f * (r + 1));          // i = lane&3, j = reg
    if (h[l * 4 + r] != e1) ok1 = 0;
    if (h[l * 4 + r] != e2) ok2 = 0;
  }
  printf("layout i=reg,j=lane&3: %d   layout i=lane&3,j=reg: %d\n", ok1, ok2);
  printf("lane0: %g %g %g %g  lane5: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23]);
  return 0;
}
