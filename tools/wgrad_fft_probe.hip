// Timing of the frequency-domain weight gradient kernels (joint-cnn-mrf_amd/csrc/wgrad_fft.hip) outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I joint-cnn-mrf_amd/csrc tools/wgrad_fft_probe.hip -o tools/wgrad_fft_probe
//   tools/wgrad_fft_probe <NY> <NX> <Cin> <Cout> <ks> [B MT iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../joint-cnn-mrf_amd/csrc/wgrad_fft.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  const int NY = atoi(argv[1]), NX = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), ks = atoi(argv[5]);
  const int B = argc > 6 ? atoi(argv[6]) : 16, MT = argc > 7 ? atoi(argv[7]) : 64, iters = argc > 8 ? atoi(argv[8]) : 5;
  const size_t F = (size_t)NY * (NX / 2 + 1);
  const size_t xb = F * MT * Cin * 12, zb = F * MT * Cout * 12, sb = jcm::wgrad_fft_scratch_bytes(NY, NX, Cin, Cout), wn = (size_t)ks * ks * Cin * Cout;
  void *xs, *zs, *sc; float *w, *dw;
  CK(hipMalloc(&xs, xb)); CK(hipMalloc(&zs, zb)); CK(hipMalloc(&sc, sb)); CK(hipMalloc(&w, wn * 4)); CK(hipMalloc(&dw, wn * 4));
  CK(hipMemset(xs, 0x3c, xb)); CK(hipMemset(zs, 0x3c, zb)); CK(hipMemset(w, 0, wn * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < iters + 1; ++it) {
    if (it == 1) CK(hipEventRecord(e0, 0));
    CK(jcm::wgrad_fft(xs, zs, sc, w, 0.f, dw, ks, NY, NX, B, MT, MT, Cin, Cout, Cout, 0));
  }
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double pb = (double)F * Cin * Cout * 8;
  printf("NY %d NX %d Cin %d Cout %d ks %d: %.1f us per call, P = %.1f MB -> %.2f TB/s if P were written and read once\n", NY, NX, Cin, Cout, ks, ms * 1e3 / iters,
         pb / 1e6, 2 * pb / (ms * 1e-3 / iters) / 1e12);
  return 0;
}
