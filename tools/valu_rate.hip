// valu_rate.hip -- issue rate of v_dot4 (the one-byte distance kernels' work instruction) next to v_fma_f32 / v_pk_fma_f32 on
// gfx950: 8 independent chains per wave, 1..4 waves per SIMD.  Prints wave-instructions per cycle per SIMD and
// chip-wide T lane-ops/s.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(int iters, unsigned* out, unsigned seed) {
  unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  int acc[8]; float facc[8]; typedef float f2 __attribute__((ext_vector_type(2))); f2 pacc[8];
  for (int i = 0; i < 8; i++) { acc[i] = i; facc[i] = (float)i; pacc[i] = f2{(float)i, 1.0f}; }
  const float fa = __uint_as_float((a & 0x007fffff) | 0x3f800000), fb = 1.0f / 1024;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) acc[i] = __builtin_amdgcn_sdot4((int)a, (int)b, acc[i], false);
        else if (OP == 1) facc[i] = __builtin_fmaf(fa, fb, facc[i]);
        else if (OP == 2) pacc[i] = __builtin_elementwise_fma(f2{fa, fa}, f2{fb, fb}, pacc[i]);
        else if (OP == 3) {      // 64-bit compare + add-with-carry-in (the rank counters of the frontier merge)
          const unsigned long long x = ((unsigned long long)a << 32) | (unsigned)(b + i), y = ((unsigned long long)(unsigned)acc[i] << 32) | (unsigned)it;
          acc[i] += (x < y) ? 1 : 0;
        } else if (OP == 4) {    // the same decision from 32-bit compares: hi < hi || (hi == hi && lo < lo)
          const unsigned xh = a, xl = b + i, yh = (unsigned)acc[i], yl = (unsigned)it;
          acc[i] += (xh < yh || (xh == yh && xl < yl)) ? 1 : 0;
        } else if (OP == 5) {    // one 32-bit compare + counter
          acc[i] += (a < (unsigned)acc[i]) ? 1 : 0;
        }
      }
  }
  unsigned r = 0;
  for (int i = 0; i < 8; i++) r += (unsigned)acc[i] + __float_as_uint(facc[i]) + __float_as_uint(pacc[i].x) + __float_as_uint(pacc[i].y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> static double run(int blocks, int iters, unsigned* d_out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_out, 2u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  unsigned* d_out; hipMalloc(&d_out, 256 * 8 * 256 * 4 * 4);
  const int iters = 20000;
  const char* names[6] = {"v_dot4_i32_i8", "v_fma_f32", "v_pk_fma_f32", "cmp_u64+addc", "cmp_u32 x3+addc", "cmp_u32+addc"};
  for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD = blocks per CU (256-thread blocks: one wave per SIMD each)
    const int blocks = 256 * wps;
    for (int op = 0; op < 6; op++) {
      const double ms = op == 0 ? run<0>(blocks, iters, d_out) : op == 1 ? run<1>(blocks, iters, d_out) : op == 2 ? run<2>(blocks, iters, d_out) : op == 3 ? run<3>(blocks, iters, d_out) : op == 4 ? run<4>(blocks, iters, d_out) : run<5>(blocks, iters, d_out);
      const double winstr = (double)blocks * 4 * iters * 32;            // wave-instructions
      const double per_simd_per_s = winstr / 1024 / (ms * 1e-3);
      printf("%-14s waves/SIMD=%d  %.3f ms  %.3g wave-instr/s/SIMD  (= one per %.2f cycles at 2.4 GHz)  chip %.1f T lane-ops/s\n", names[op], wps, ms,
             per_simd_per_s, 2.4e9 / per_simd_per_s, winstr * 64 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
