// Does `s_waitcnt vmcnt(0)` cover an LDS-DMA (buffer_load ... lds) whose SOURCE is outside the buffer descriptor's range?  (round 4)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_dma_oob_probe.hip -o tools/lds_dma_oob_probe
//   tools/lds_dma_oob_probe [iters] [load: 0 = idle chip, 1 = a device copy runs on another stream]
//
// A bounds-checked buffer load that is out of range "returns zero".  With an LDS destination the zero has to be WRITTEN into LDS, and the
// kernels of this library used such requests in two ways: as the look-ahead of a DMA ring past the last stage (no tail logic), and as the
// SAME padding of a convolution window.  The two-engine soak of round 3/4 showed 128-byte pieces of ANOTHER kernel's LDS being zeroed at
// arbitrary times; bisected to the channel GEMM's out-of-range look-ahead: the request is retired from the wave's vmcnt before its zero
// fill lands, the wave ends, the work group's LDS is handed to the next work group on that CU, and the zeros land in the new owner's data.
//
// This probe measures the window directly.  Each work group (1 wave) fills 1 KB of LDS with a pattern, issues ONE 16-byte-per-lane
// LDS-DMA from an out-of-range (mode 0) or in-range (mode 1) source into it, waits vmcnt(0), and reads the 1 KB back at once:
//   stale = lanes that still see the pattern after vmcnt(0)  (the request was retired before its data / zeros landed)
// and then polls until the expected bytes appear, reporting the largest delay in s_memtime ticks (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe_kernel(const uint4* src, unsigned src_bytes, int oob, unsigned long long* out, int rounds) {
  __shared__ uint4 lds[64 * 8];      // 8 KB: the DMA target moves through it
  const int lane = threadIdx.x;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, src_bytes, 0x00020000);
  unsigned long long stale = 0, maxdelay = 0, never = 0;
  for (int r = 0; r < rounds; ++r) {
    uint4* dst = lds + 64 * (r & 7);
    dst[lane] = make_uint4(0xdeadbeefu, 0xdeadbeefu ^ (unsigned)lane, 0xdeadbeefu, 0xdeadbeefu);
    __builtin_amdgcn_s_waitcnt(0);      // the pattern is in LDS
    __builtin_amdgcn_s_barrier();
    const unsigned voff = (unsigned)lane * 16u + (oob ? src_bytes + 4096u : ((unsigned)(blockIdx.x * 131 + r) % (src_bytes / 1024u - 1)) * 1024u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 v = dst[lane];
    const bool fresh0 = oob ? (v.x | v.y | v.z | v.w) == 0 : v.x != 0xdeadbeefu;
    if (!fresh0) {
      ++stale;
      bool ok = false;
      for (int spin = 0; spin < 200000 && !ok; ++spin) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)&dst[lane]) : "memory");
        ok = oob ? (v.x | v.y | v.z | v.w) == 0 : v.x != 0xdeadbeefu;
      }
      const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
      if (!ok) ++never;
      else if (dt > maxdelay) maxdelay = dt;
    }
  }
  if (stale) { atomicAdd(out, stale); atomicMax(out + 1, maxdelay); atomicAdd(out + 2, never); }
}

__global__ void copy_kernel(const uint4* a, uint4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20, load = argc > 2 ? atoi(argv[2]) : 0;
  const unsigned src_bytes = 64u << 20;
  uint4 *src, *ca, *cb;
  unsigned long long* out;
  CK(hipMalloc(&src, src_bytes)); CK(hipMemset(src, 0x5a, src_bytes));
  CK(hipMalloc(&out, 3 * 8));
  const size_t cpn = (size_t)512 << 20;
  CK(hipMalloc(&ca, cpn)); CK(hipMalloc(&cb, cpn)); CK(hipMemset(ca, 1, cpn));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int oob = 1; oob >= 0; --oob) {
    unsigned long long tot[3] = {0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      CK(hipMemsetAsync(out, 0, 3 * 8, s1));
      if (load) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s2, ca, cb, cpn / 16);
      hipLaunchKernelGGL(probe_kernel, dim3(2048), dim3(64), 0, s1, src, src_bytes, oob, out, 256);
      unsigned long long h[3];
      CK(hipMemcpyAsync(h, out, 3 * 8, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipStreamSynchronize(s2));
      tot[0] += h[0]; tot[2] += h[2];
      if (h[1] > tot[1]) tot[1] = h[1];
    }
    printf("%s source, %s: %llu of %llu lane-reads stale right after vmcnt(0); longest wait until the bytes appeared %llu ticks of 10 ns; never appeared: %llu\n",
           oob ? "OUT-OF-RANGE" : "in-range    ", load ? "copy running on another stream" : "idle chip", tot[0], (unsigned long long)iters * 2048 * 256 * 64, tot[1], tot[2]);
  }
  return 0;
}
