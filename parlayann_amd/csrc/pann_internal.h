// pann_internal.h -- private declarations shared by the HIP translation units of libpann.so.
// Target: gfx950 (MI355X, CDNA4) only.  64-lane wavefronts are assumed everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>

#include "../../include/pann.h"

namespace pann {

constexpr uint32_t SENTINEL = 0xFFFFFFFFu;  // empty adjacency slot on the device / empty filter slot

// Diagnostic A/B switches (environment variables that select kernel variants for same-box comparisons, tools/ab_*.sh) exist
// only in `make alt` builds (-DPANN_AB, lib/libpann_alt.so); the shipped libpann.so reads no environment variable.
#ifdef PANN_AB
inline const char* ab_env(const char* name) { return getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif

// ---- device-side layout of one index (DESIGN.md "Data layout in HBM") ----
//  points: n rows, row stride pstride = nch * lpc * 16 bytes (>= d*esize), zero padded
//  graph : n rows of gstride uint32 (gstride = max_deg rounded up to 16), neighbours packed at the
//          front, unused slots = SENTINEL; the degree is not stored (it is the count of
//          non-sentinel slots), so a row of max_deg 64 is exactly one aligned 256-byte read.
struct DeviceIndex {
  uint8_t* points = nullptr;
  uint32_t* graph = nullptr;
  uint64_t n = 0;
  uint32_t d = 0;
  int dtype = 0, metric = 0;
  uint32_t esize = 0;    // bytes per element
  uint32_t dbytes = 0;   // d * esize
  uint32_t pstride = 0;  // device row stride in bytes
  uint32_t lpc = 0;      // lanes per candidate in the gather-distance loops (4,8,16,32)
  uint32_t nch = 0;      // 16-byte chunks per lane: pstride = nch*lpc*16
  uint32_t exact = 0;    // exact float order (validation mode): lane-per-candidate sequential sums; forces lpc=4
  uint32_t max_deg = 0;
  uint32_t gstride = 0;  // uint32 per graph row on the device
  uint32_t forest_group = 0;  // HCNNG: trees split level by level together (0 = as many as 2^31 positions allow)
  // filter-code table of the beam-91..128 searches (filter_codes.hip): the code of every id and, slot-aligned with the graph rows,
  // of every neighbour.  gcode is only read while codes_valid: every writer of graph rows either maintains it or clears the flag.
  const uint16_t* rank16 = nullptr;   // [n]
  uint16_t* gcode = nullptr;          // [n][gstride]
  uint32_t codes_valid = 0;
  // locality cell of every point (nearest of LOCALITY_PIVOTS pivots), for the order in which a batch's searches are launched
  // (vamana_build.hip: queries that run side by side then read rows of the same few regions); null = not computed
  const uint32_t* cell = nullptr;     // [n]
  uint32_t cell_min_batch = 4096;     // batches below this many searches keep the batch order
};

// The kernels come in two families (PANN_LAYOUT_SWITCH): rows that are ONE 16-byte chunk per lane with 8 / 16 / 32
// lanes per candidate keep the query in registers; every other layout -- including 64-byte rows, lpc 4 / nch 1 --
// runs the generic variants, which keep the query in LDS.  All LDS sizes on the host must use THIS predicate.
inline bool layout_query_in_registers(const DeviceIndex& ix) { return ix.nch == 1 && ix.lpc != 4; }
inline size_t query_lds_bytes(const DeviceIndex& ix) { return layout_query_in_registers(ix) ? 0 : (size_t)ix.nch * ix.lpc * 16; }

struct SearchArgs {  // one batched beam search, everything device resident
  const uint8_t* queries; uint64_t qstride;  // external queries (or null)
  const uint32_t* query_ids;                 // base-point queries (or null)
  uint64_t nq;
  const uint32_t* starts; uint32_t nstarts;
  int starts_per_query = 0;                  // starts is nq x nstarts (beamSearchRandom)
  int64_t k, beam, limit, degree_limit; double cut;
  uint32_t dcap = 256;                       // dropped-list entries per query (pann_index_reserve_dropped)
  const uint32_t* order = nullptr;           // device, nq entries: launch slot -> query index (a permutation); null = identity
  pann_search_out out;
};

// per-handle scratch that the search kernels need (grown on demand, never shrunk)
struct Workspace {
  void* buf = nullptr; size_t bytes = 0;
  int ensure(size_t need);
  void release();
};

void set_error(const std::string& s);
int hip_fail(hipError_t e, const char* what);
#define PANN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return pann::hip_fail(e_, #call); } while (0)

// beam_search.hip
size_t search_workspace_bytes(const DeviceIndex& ix, const SearchArgs& a);
int launch_beam_search(const DeviceIndex& ix, const SearchArgs& a, void* ws, size_t ws_bytes,
                       hipStream_t stream);
void choose_point_layout(uint32_t dbytes, uint32_t* lpc, uint32_t* nch);

// vamana_build.hip
int robust_prune_batch_host(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint32_t* owners,
                            uint64_t m, const uint32_t* cand_ids, const float* cand_dists,
                            const uint64_t* cand_offsets, double alpha, uint32_t R, int add_out_nbrs,
                            uint32_t* out_rows, uint32_t* out_dist_cmps);
int insert_batch_dev(const DeviceIndex& ix, Workspace& ws, Workspace& ws2, Workspace& search_ws, Workspace& rows_ws, hipStream_t st,
                     const uint32_t* d_batch, uint32_t m, uint32_t start, uint32_t R, uint32_t L, double alpha,
                     uint32_t* vcap_io, pann_build_stats* stats);
// the two phases of a batch (vamana_build.hip): A reads the graph and yields rows for the points given, B applies the
// rows of the whole batch; d_rows is m x R uint32, SENTINEL padded
int vamana_search_prune_dev(const DeviceIndex& ix, Workspace& ws, Workspace& search_ws, hipStream_t st, const uint32_t* d_batch,
                            uint32_t m, uint32_t start, uint32_t R, uint32_t L, double alpha, uint32_t* vcap_io,
                            uint32_t* d_rows, pann_build_stats* stats);
int vamana_apply_rows_dev(const DeviceIndex& ix, Workspace& ws, Workspace& ws2, hipStream_t st, const uint32_t* d_batch, uint32_t m,
                          const uint32_t* d_rows, uint32_t R, double alpha, pann_build_stats* stats);
int sort_neighbors_dev(const DeviceIndex& ix, hipStream_t st);

// dense.hip
int dense_topk_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint8_t* d_a_ext, uint64_t a_stride,
                   const uint32_t* d_a_ids, const uint32_t* d_b_ids, const uint64_t* d_a_off, const uint64_t* d_b_off,
                   const uint32_t* d_tile_seg, const uint32_t* d_tile_a0, uint32_t ntiles, uint64_t na, uint64_t nb,
                   uint32_t nsplit, uint32_t m, int exclude_same, uint32_t* d_out_ids, float* d_out_dists);
int query_distances_dev(const DeviceIndex& ix, hipStream_t st, const uint8_t* d_q_ext, uint64_t q_stride,
                        const uint32_t* d_q_ids, uint64_t nq, const uint32_t* d_ids, uint64_t m, int paired,
                        float* d_out);

int pivot_split_dev(const DeviceIndex& ix, hipStream_t st, const uint32_t* d_ids, const uint32_t* d_tile_seg,
                    const uint64_t* d_tile_lo, const uint32_t* d_tile_cnt, uint32_t ntiles, const uint32_t* d_pa,
                    const uint32_t* d_pb, uint8_t* d_side);

int rerank_dev(const DeviceIndex& ix, hipStream_t st, const uint8_t* d_q, uint64_t q_stride, uint64_t nq,
               const uint32_t* d_cand, uint32_t c, const uint32_t* d_cnt, uint32_t k, int resort, uint32_t* d_out_ids,
               float* d_out_dists);

// leaf_knn.hip: lane-owns-row all-pairs top-m for one-byte element types (HCNNG leaves)
bool dense_gt_eligible(const DeviceIndex& ix, uint32_t m, bool b_ids, bool segmented, int exclude_same);
uint32_t dense_gt_slots(const DeviceIndex& ix, uint32_t m);
bool leaf_knn_rows_eligible(const DeviceIndex& ix, uint32_t m);
int leaf_knn_rows_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint32_t* d_ids, const uint64_t* d_off,
                      const uint64_t* h_off, uint64_t nseg, uint32_t m, int exclude_same, uint32_t* d_out_ids, float* d_out_dists);

// filter_codes.hip
constexpr uint32_t FILTER_CODE_BITS = 12;      // the table size the codes are made for: beam 91..128 (beamSearch.h:52)
int filter_codes_build_ranks(uint64_t n, uint32_t bits, Workspace& ws, hipStream_t st, uint16_t* rank16, uint32_t* max_rank_out);
int filter_codes_rebuild_rows(const DeviceIndex& ix, hipStream_t st);

// range_search.hip
int range_search_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint8_t* d_q, uint64_t q_stride,
                     const uint32_t* d_qids, uint64_t nq, const uint32_t* d_starts, uint32_t nstarts,
                     int starts_per_query, float radius_2, uint32_t cap, uint32_t* d_out_ids, uint32_t* d_out_counts,
                     uint32_t* d_out_cmps, uint32_t* d_out_trunc);

// hcnng_build.hip
int hcnng_build_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, uint32_t num_clusters, uint32_t cluster_size,
                    uint32_t mst_deg, uint64_t seed, double* times3, uint32_t first_tree = 0, uint32_t tree_step = 1,
                    uint32_t* slab = nullptr, uint32_t slab_stride = 0);
int hcnng_assemble_dev(const DeviceIndex& ix, hipStream_t st, const uint32_t* d_slabs, uint32_t W, uint32_t slab_stride,
                       uint32_t ntrees, uint32_t mst_deg);

}  // namespace pann
