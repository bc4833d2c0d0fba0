// merge.hip -- top-k merge of per-shard result lists on the device (sharded index): every query
// was searched on W shards, each returning k (global id, dist) pairs; the answer is the k smallest of the W*k under the
// reference's (dist, id) order (beamSearch.h:46-48).  One wave per query: keys in LDS, rank by counting.
#include "pann_device.h"

namespace pann {

__global__ void __launch_bounds__(PANN_WAVE) merge_topk_kernel(const uint32_t* __restrict__ ids, const float* __restrict__ dists,
                                                               uint32_t nlists, uint64_t nq, uint32_t kin, uint32_t kout,
                                                               uint32_t row_stride, const uint32_t* __restrict__ list_base,
                                                               uint32_t* __restrict__ out_ids, float* __restrict__ out_dists) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint64_t* K = reinterpret_cast<uint64_t*>(smem);
  const int lane = threadIdx.x;
  const uint64_t q = blockIdx.x;
  const uint32_t tot = nlists * kin;
  for (uint32_t j = lane; j < tot; j += PANN_WAVE) {
    const uint32_t w = j / kin, t = j % kin;
    const uint64_t at = ((uint64_t)w * nq + q) * row_stride + t;
    const uint32_t id = ids[at];
    // unused slots of a short list sort last; list w holds ids local to shard w when list_base is given
    K[j] = id == SENTINEL ? KEY_INF : make_key(dists[at], list_base ? id + list_base[w] : id);
  }
  __syncthreads();
  for (uint32_t j = lane; j < kout; j += PANN_WAVE) { out_ids[q * kout + j] = SENTINEL; out_dists[q * kout + j] = __builtin_inff(); }
  __syncthreads();
  for (uint32_t j = lane; j < tot; j += PANN_WAVE) {
    const uint64_t key = K[j];
    if (key == KEY_INF) continue;
    uint32_t r = 0;
    for (uint32_t i = 0; i < tot; i++) { const uint64_t o = K[i]; r += (o < key || (o == key && i < j)) ? 1u : 0u; }
    if (r < kout) { out_ids[q * kout + r] = key_id(key); out_dists[q * kout + r] = key_dist(key); }
  }
}

}  // namespace pann

extern "C" int pann_merge_topk_dev(const uint32_t* d_ids, const float* d_dists, uint32_t nlists, uint64_t nq, uint32_t k_in,
                                   uint32_t row_stride, const uint32_t* d_list_base, uint32_t k_out, uint32_t* d_out_ids,
                                   float* d_out_dists, void* stream) {
  using namespace pann;
  if (nq == 0) return PANN_OK;
  if (!d_ids || !d_dists || !d_out_ids || !d_out_dists || nlists == 0 || k_in == 0 || k_out == 0) { set_error("pann_merge_topk_dev: null / zero argument"); return PANN_ERR_BAD_ARG; }
  if (row_stride < k_in) { set_error("pann_merge_topk_dev: row_stride < k_in"); return PANN_ERR_BAD_ARG; }
  const size_t lds = (size_t)nlists * k_in * 8;
  if (lds > 64 * 1024 || nq > 0x7FFFFFFFull) { set_error("pann_merge_topk_dev: nlists * k_in too large (max 8192 pairs per query)"); return PANN_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(merge_topk_kernel, dim3((uint32_t)nq), dim3(PANN_WAVE), lds, (hipStream_t)stream, d_ids, d_dists, nlists, nq, k_in, k_out,
                     row_stride, d_list_base, d_out_ids, d_out_dists);
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}
