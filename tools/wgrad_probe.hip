// Stand-alone check of wgrad_conv1 / wgrad_f32 against a CPU loop (debugging aid).
#include "../joint-cnn-mrf_amd/csrc/wgrad.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
using namespace jcm;
int main() {
  const int B = 2, H0 = 480, W0 = 720, C = 16;
  for (int sub = 1; sub <= 4; sub *= 2) {
    const int Hs = H0 / sub, Ws = W0 / sub, Ho = (Hs + 1) / 2, Wo = (Ws + 1) / 2;
    std::vector<float> x((size_t)B * H0 * W0 * 3), dz((size_t)B * Ho * Wo * C);
    srand(1);
    for (auto& v : x) v = rand() / (float)RAND_MAX;
    for (auto& v : dz) v = rand() / (float)RAND_MAX - 0.5f;
    std::vector<double> ref(75 * C, 0.0);
    for (int b = 0; b < B; ++b)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox)
          for (int ky = 0; ky < 5; ++ky)
            for (int kx = 0; kx < 5; ++kx) {
              const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
              if (iy < 0 || iy >= Hs || ix < 0 || ix >= Ws) continue;
              for (int ch = 0; ch < 3; ++ch) {
                const double xv = x[(((size_t)b * H0 + (size_t)iy * sub) * W0 + (size_t)ix * sub) * 3 + ch];
                for (int co = 0; co < C; ++co) ref[((ky * 5 + kx) * 3 + ch) * C + co] += xv * dz[(((size_t)b * Ho + oy) * Wo + ox) * C + co];
              }
            }
    float *dx, *dd, *dp, *dw, *w;
    const int nb = wgrad_conv1_blocks();
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dd, dz.size() * 4); hipMalloc(&dp, (size_t)nb * 75 * C * 4); hipMalloc(&dw, 75 * C * 4); hipMalloc(&w, 75 * C * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, dz.data(), dz.size() * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0, 75 * C * 4);
    wgrad_conv1(dx, dd, dp, B, H0, W0, sub, C, 0);
    wgrad_reduce(dp, nb, 75 * C, w, 0.f, dw, 0);
    std::vector<float> got(75 * C);
    hipMemcpy(got.data(), dw, 75 * C * 4, hipMemcpyDeviceToHost);
    double me = 0, mx = 0; int worst = 0;
    for (int i = 0; i < 75 * C; ++i) { double e = fabs(got[i] - ref[i]); if (e > me) { me = e; worst = i; } if (fabs(ref[i]) > mx) mx = fabs(ref[i]); }
    printf("sub %d: max err %.3e (max |ref| %.3e) worst idx %d (tap %d ch %d co %d) got %.6f ref %.6f\n", sub, me, mx, worst, worst / C / 3, (worst / C) % 3, worst % C, got[worst], ref[worst]);
  }
  return 0;
}
