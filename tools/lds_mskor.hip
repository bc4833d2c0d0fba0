// ds_mskor_b32 on gfx950: D = (D & ~mask) | data, atomically per lane -- does it update disjoint nibbles of ONE dword from several
// lanes of one instruction correctly?  (the 4-bit plane of the 12-bit filter-code table, beam_search.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  __shared__ unsigned W[64];
  W[threadIdx.x] = 0xFFFFFFFFu;
  __syncthreads();
  unsigned addr = (unsigned)(uintptr_t)&W[threadIdx.x >> 3];
  unsigned sh = (threadIdx.x & 7) * 4;
  unsigned mask = 0xFu << sh, data = ((threadIdx.x * 7 + 3) & 0xF) << sh;
  asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(addr), "v"(mask), "v"(data) : "memory");
  __syncthreads();
  out[threadIdx.x] = W[threadIdx.x];
}
int main() {
  unsigned* d; unsigned h[64];
  hipMalloc(&d, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 8; w++) {
    unsigned e = 0;
    for (int j = 0; j < 8; j++) e |= (((w * 8 + j) * 7 + 3) & 0xF) << (4 * j);
    if (h[w] != e) { bad++; printf("word %d: got %08x want %08x\n", w, h[w], e); }
  }
  for (int w = 8; w < 64; w++) if (h[w] != 0xFFFFFFFFu) bad++;
  printf("ds_mskor_b32 nibble test: %s\n", bad ? "FAIL" : "ok");
  return bad != 0;
}
