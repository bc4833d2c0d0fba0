#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// does global_load_lds_dwordx4 exist / work on gfx950?  each lane names its own 16-byte global source; the LDS destination is
// a wave-uniform base + lane*16
__global__ void k(const uint4* src, const uint32_t* idx, uint4* out, int n) {
  __shared__ uint4 buf[64];
  const int lane = threadIdx.x;
  const uint4* g = src + idx[blockIdx.x * 64 + lane];
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[blockIdx.x * 64 + lane] = buf[lane];
}
int main() {
  const int n = 1 << 16;
  std::vector<uint4> h(n); std::vector<uint32_t> ix(n);
  for (int i = 0; i < n; i++) { h[i] = make_uint4(i, i * 3, i ^ 5, ~i); ix[i] = (uint32_t)((i * 2654435761u) % n); }
  uint4 *d, *o; uint32_t* di;
  hipMalloc(&d, n * 16); hipMalloc(&o, n * 16); hipMalloc(&di, n * 4);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(di, ix.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, di, o, n);
  std::vector<uint4> r(n); hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) { const uint4 e = h[ix[i]]; if (r[i].x != e.x || r[i].y != e.y || r[i].z != e.z || r[i].w != e.w) bad++; }
  printf("global_load_lds_dwordx4: %d mismatches of %d\n", bad, n);
  return bad != 0;
}
