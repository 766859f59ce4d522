// Stand-alone check of wgrad_conv1 / wgrad_f32 against a CPU loop (debugging aid).
#include "../joint-cnn-mrf_amd/csrc/wgrad.hip"
#include "../joint-cnn-mrf_amd/csrc/wgrad_split.hip"
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
    wgrad_conv1(dx, dd, false, dp, B, H0, W0, sub, C, 0);
    wgrad_reduce(dp, nb, 75 * C, w, 0.f, dw, 0);
    std::vector<float> got(75 * C);
    hipMemcpy(got.data(), dw, 75 * C * 4, hipMemcpyDeviceToHost);
    double me = 0, mx = 0; int worst = 0;
    for (int i = 0; i < 75 * C; ++i) { double e = fabs(got[i] - ref[i]); if (e > me) { me = e; worst = i; } if (fabs(ref[i]) > mx) mx = fabs(ref[i]); }
    printf("sub %d: max err %.3e (max |ref| %.3e) worst idx %d (tap %d ch %d co %d) got %.6f ref %.6f\n", sub, me, mx, worst, worst / C / 3, (worst / C) % 3, worst % C, got[worst], ref[worst]);
  }
  for (int ks = 5; ks <= 9; ks += 4) {
    const int B2 = 2, H = 7, W = 45, Cin = 16, Cout = 24, ldz = 24, pad = (ks - 1) / 2;
    std::vector<float> x((size_t)B2 * H * W * Cin), dz((size_t)B2 * H * W * ldz);
    for (auto& v : x) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : dz) v = rand() / (float)RAND_MAX - 0.5f;
    const size_t n = (size_t)ks * ks * Cin * Cout;
    std::vector<double> ref(n, 0.0);
    for (int b = 0; b < B2; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx)
      for (int ky = 0; ky < ks; ++ky) for (int kx = 0; kx < ks; ++kx) {
        const int yi = y + ky - pad, xi = xx + kx - pad;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
        for (int ci = 0; ci < Cin; ++ci) for (int co = 0; co < Cout; ++co)
          ref[((size_t)(ky * ks + kx) * Cin + ci) * Cout + co] += (double)x[(((size_t)b * H + yi) * W + xi) * Cin + ci] * dz[(((size_t)b * H + y) * W + xx) * ldz + co];
      }
    const int splits = wgrad_splits(ks, Cin, Cout, B2, H);
    float *dx, *dd, *dp, *dw, *w;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dd, dz.size() * 4); hipMalloc(&dp, n * splits * 4); hipMalloc(&dw, n * 4); hipMalloc(&w, n * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, dz.data(), dz.size() * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0, n * 4);
    hipError_t e = wgrad_f32(dx, dd, dp, splits, ks, B2, H, W, Cin, Cout, ldz, 0);
    wgrad_reduce(dp, splits, n, w, 0.f, dw, 0);
    std::vector<float> got(n);
    hipMemcpy(got.data(), dw, n * 4, hipMemcpyDeviceToHost);
    double me = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { me = fmax(me, fabs(got[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
    printf("wgrad ks %d splits %d (%s): max err %.3e (max |ref| %.3e)\n", ks, splits, hipGetErrorString(e), me, mx);
    // the same through the bf16x6 split kernel
    void *xp, *zp;
    hipMalloc(&xp, x.size() * 6); hipMalloc(&zp, dz.size() * 6);
    split_parts(dx, xp, x.size(), 0); split_parts(dd, zp, dz.size(), 0);
    hipMemset(dp, 0xff, n * splits * 4);
    e = wgrad_split(xp, zp, dp, splits, ks, B2, H, W, Cin, Cout, ldz, 0);
    wgrad_reduce(dp, splits, n, w, 0.f, dw, 0);
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(got.data(), dw, n * 4, hipMemcpyDeviceToHost);
    me = 0;
    size_t worst = 0;
    for (size_t i = 0; i < n; ++i) { double er = fabs(got[i] - ref[i]); if (!(er <= me)) { me = er; worst = i; } }
    printf("wgrad_split ks %d (%s / %s): max err %.3e at tap %zu ci %zu co %zu (got %.5f ref %.5f)\n", ks, hipGetErrorString(e), hipGetErrorString(e2), me,
           worst / (Cin * Cout), (worst / Cout) % Cin, worst % Cout, got[worst], ref[worst]);
  }
  return 0;
}
