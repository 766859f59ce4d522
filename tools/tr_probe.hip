// Probe of ds_read_b64_tr_b16 lane/element semantics on gfx950 (debugging aid for the bf16 weight-gradient kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned short u16;
__global__ void k(unsigned* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) u16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (u16)i;
  __syncthreads();
  const int l = threadIdx.x;
  // hypothesis: in each 16-lane group, lane t' supplies the 8-byte chunk (row t'/4, cols 4*(t'%4)..+3) of a [4 rows][16 cols] block
  const int g = l >> 4, t = l & 15;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) u16*)lds;
  const unsigned addr = base + (unsigned)((t >> 2) * stride_bytes + (t & 3) * 8 + g * 32);   // group g: columns 16g..16g+15
  unsigned v0, v1;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(unsigned long long*)&v0) : "v"(addr) : "memory");
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[l * 2] = (unsigned)r; out[l * 2 + 1] = (unsigned)(r >> 32);
  (void)v0; (void)v1;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 2 * 4);
  const int stride = 256;   // bytes per row (128 elements)
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
  std::vector<unsigned> h(128); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    const unsigned e0 = h[2 * l] & 0xffff, e1 = h[2 * l] >> 16, e2 = h[2 * l + 1] & 0xffff, e3 = h[2 * l + 1] >> 16;
    printf("lane %2d: %4u %4u %4u %4u   (row,col) = (%u,%u) (%u,%u) (%u,%u) (%u,%u)\n", l, e0, e1, e2, e3, e0 / 128, e0 % 128, e1 / 128, e1 % 128, e2 / 128, e2 % 128, e3 / 128, e3 % 128);
  }
  return 0;
}
