// filter_codes.hip -- class codes for the lossy filter of filtered_beam_search (beamSearch.h:52-59).
//
// The reference's filter is a direct-mapped table of 1 << bits vertex ids: slot(a) = hash64_2(a) & mask, "seen" iff
// table[slot(a)] == a.  What a slot must remember is WHICH of the ids that hash to it was written last -- and there are only
// about n >> bits of those (2 441 at n = 10M, bits = 12), not n.  Give every id its rank among the ids of its own slot class,
// in increasing id order (its "code"): (slot, code) <-> id is a bijection, so a table of CODES decides exactly what the table of
// ids decides, in 12 bits per slot instead of 32 when every class has fewer than 4 095 members (n below ~16M at bits = 12).
// The whole 4 096-slot table of the beam-91..128 searches (Vamana's L = 128 build searches) then takes 6 KB of LDS: no part of
// it lives in HBM any more (round 2: half of it, 1.22 x the algorithmic bytes moved, one dependent table load per row).
//
// The code of a neighbour must not cost a random access per neighbour, so the codes travel WITH the graph: a side array
// gcode[n][gstride] of uint16, slot-aligned with the adjacency rows, read with the row (128 coalesced bytes per visited vertex).
// It is maintained by the builder's row writers (vamana_build.hip) from rank16[id]; anything else that changes the graph marks it
// stale and the searches fall back to the id table.
#include <rocprim/device/device_radix_sort.hpp>

#include "pann_device.h"

namespace pann {

namespace {

__global__ void code_keys_kernel(uint64_t n, uint32_t mask, uint64_t* keys) {
  const uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (a < n) keys[a] = ((hash64_2(a) & mask) << 32) | a;
}

// sorted by (slot, id): the first position of every slot class
__global__ void code_starts_kernel(const uint64_t* keys, uint64_t n, uint32_t* class_start) {
  const uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t s = (uint32_t)(keys[p] >> 32);
  if (p == 0 || (uint32_t)(keys[p - 1] >> 32) != s) class_start[s] = (uint32_t)p;
}

__global__ void code_ranks_kernel(const uint64_t* keys, uint64_t n, const uint32_t* class_start, uint16_t* rank16, uint32_t* max_rank) {
  const uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t k = keys[p];
  const uint32_t r = (uint32_t)p - class_start[(uint32_t)(k >> 32)];
  rank16[(uint32_t)k] = (uint16_t)min(r, 0xFFFFu);
  if (r >= 0xFFFu) atomicMax(max_rank, r);            // (rare: only classes near the 12-bit limit touch the counter)
}

// gcode row <- codes of the ids of the graph row (whole-graph rebuild: a graph that was uploaded, or built by another builder)
__global__ void code_rows_kernel(const uint32_t* graph, uint32_t gstride, uint64_t n, const uint16_t* rank16, uint16_t* gcode) {
  const uint64_t v = blockIdx.x;
  for (uint32_t j = threadIdx.x; j < gstride; j += blockDim.x) {
    const uint32_t a = graph[v * gstride + j];
    gcode[v * gstride + j] = a == SENTINEL ? (uint16_t)0xFFFF : rank16[a];
  }
}

}  // namespace

// rank16[a] for all a < n at 1 << bits slots; *max_rank_out = the largest code of a class with 4 095 or more members (0: none)
int filter_codes_build_ranks(uint64_t n, uint32_t bits, Workspace& ws, hipStream_t st, uint16_t* rank16, uint32_t* max_rank_out) {
  if (n == 0 || n >= 0xFFFFFFF0ull) { set_error("filter codes: n out of range"); return PANN_ERR_BAD_ARG; }
  size_t sort_tmp = 0;
  (void)rocprim::radix_sort_keys(nullptr, sort_tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0, 64, st);
  const size_t kbytes = ((size_t)n * 8 + 255) / 256 * 256;
  const size_t cbytes = (((size_t)1 << bits) * 4 + 255) / 256 * 256;
  if (int rc = ws.ensure(2 * kbytes + cbytes + 256 + sort_tmp + 256)) return rc;
  uint8_t* w = (uint8_t*)ws.buf;
  uint64_t* ka = (uint64_t*)w; uint64_t* kb = (uint64_t*)(w + kbytes);
  uint32_t* cstart = (uint32_t*)(w + 2 * kbytes); uint32_t* d_max = (uint32_t*)(w + 2 * kbytes + cbytes);
  void* tmp = w + 2 * kbytes + cbytes + 256;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(code_keys_kernel, dim3(nb), dim3(256), 0, st, n, (1u << bits) - 1u, ka);
  PANN_HIP(rocprim::radix_sort_keys(tmp, sort_tmp, ka, kb, (size_t)n, 0, 32 + bits, st));
  PANN_HIP(hipMemsetAsync(d_max, 0, 4, st));
  hipLaunchKernelGGL(code_starts_kernel, dim3(nb), dim3(256), 0, st, kb, n, cstart);
  hipLaunchKernelGGL(code_ranks_kernel, dim3(nb), dim3(256), 0, st, kb, n, cstart, rank16, d_max);
  PANN_HIP(hipGetLastError());
  PANN_HIP(hipMemcpyAsync(max_rank_out, d_max, 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

int filter_codes_rebuild_rows(const DeviceIndex& ix, hipStream_t st) {
  if (ix.n == 0) return PANN_OK;
  hipLaunchKernelGGL(code_rows_kernel, dim3((uint32_t)ix.n), dim3(64), 0, st, ix.graph, ix.gstride, ix.n, ix.rank16, ix.gcode);
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

}  // namespace pann
