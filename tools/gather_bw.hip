// gather_bw.hip -- ceiling probe: random whole-row gathers (16 B per lane, LPC lanes per row) from a
// table far larger than L2, in the access shape of beam_search_kernel's gather_tile.
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o gpurun_out/gather_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

template <int LPC, int U>
__global__ void __launch_bounds__(64) gather_kernel(const uint8_t* table, uint32_t stride, const uint32_t* ids,
                                                    uint32_t per_wave, uint32_t* sink) {
  const int lane = threadIdx.x, grp = lane / LPC, sub = lane % LPC;
  constexpr int G = 64 / LPC;
  const uint32_t* my = ids + (size_t)blockIdx.x * per_wave;
  uint32_t acc = 0;
  for (uint32_t s0 = 0; s0 + G * U <= per_wave; s0 += G * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const uint4*>(table + (size_t)my[s0 + u * G + grp] * stride + sub * 16);
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// The same gather through LDS: global_load_lds_dwordx4 (gfx950) lands 64 lanes x 16 B = 1 KB per instruction in a wave-uniform
// LDS window without touching VGPRs, so U can be large (U KB of LDS per wave instead of 4*U VGPRs per lane).
template <int LPC, int U>
__global__ void __launch_bounds__(64) gather_lds_kernel(const uint8_t* table, uint32_t stride, const uint32_t* ids,
                                                        uint32_t per_wave, uint32_t* sink) {
  __shared__ uint4 win[U * 64];
  const int lane = threadIdx.x, grp = lane / LPC, sub = lane % LPC;
  constexpr int G = 64 / LPC;
  const uint32_t* my = ids + (size_t)blockIdx.x * per_wave;
  uint32_t acc = 0;
  for (uint32_t s0 = 0; s0 + G * U <= per_wave; s0 += G * U) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint8_t* g = table + (size_t)my[s0 + u * G + grp] * stride + sub * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(win + u * 64), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < U; u++) { const uint4 v = win[u * 64 + lane]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    __builtin_amdgcn_wave_barrier();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int LPC, int U, bool LDS = false>
void run(const uint8_t* d_table, uint32_t stride, uint64_t nrows, int waves, uint32_t per_wave, const char* tag) {
  std::vector<uint32_t> h((size_t)waves * per_wave);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % nrows); }
  uint32_t *d_ids, *d_sink;
  hipMalloc(&d_ids, h.size() * 4); hipMalloc(&d_sink, 4);
  hipMemcpy(d_ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    if (LDS) hipLaunchKernelGGL((gather_lds_kernel<LPC, U>), dim3(waves), dim3(64), 0, 0, d_table, stride, d_ids, per_wave, d_sink);
    else hipLaunchKernelGGL((gather_kernel<LPC, U>), dim3(waves), dim3(64), 0, 0, d_table, stride, d_ids, per_wave, d_sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep == 2) printf("%s%s rows=%llu stride=%u LPC=%d U=%d waves=%d: %.3f ms, %.2f TB/s\n", LDS ? "[via LDS] " : "", tag, (unsigned long long)nrows, stride, LPC, U, waves, ms,
                         (double)waves * per_wave * (LPC * 16) / (ms * 1e-3) / 1e12);
  }
  hipFree(d_ids); hipFree(d_sink);
}

int main() {
  const uint64_t bytes = 4ull << 30;
  uint8_t* d_table; hipMalloc(&d_table, bytes); hipMemset(d_table, 1, bytes);
  // 256-B rows (fp16 d=128): 1M rows (256 MB, the bench's table) and 16M rows (4 GB)
  run<16, 4>(d_table, 256, 1u << 20, 10000, 2048, "sift1m-like");
  run<16, 4>(d_table, 256, 1u << 20, 40000, 2048, "sift1m-like");
  run<16, 8>(d_table, 256, 1u << 20, 40000, 2048, "sift1m-like");
  run<16, 4>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 8>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 4, true>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 8, true>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 16, true>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 32, true>(d_table, 256, 16u << 20, 40000, 2048, "4GB-table");
  run<16, 16, true>(d_table, 256, 1u << 20, 40000, 2048, "sift1m-like");
  run<8, 4>(d_table, 128, 16u << 20, 40000, 2048, "u8-128B");
  run<32, 4>(d_table, 512, 8u << 20, 40000, 2048, "f32-512B");
  hipFree(d_table);
  return 0;
}
