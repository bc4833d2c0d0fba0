/*
 * pann.h -- C-ABI of the MI355X-native graph-ANN hot path (beam search, robustPrune,
 * batched / all-pairs distances) that replaces, for this path only, ParlayANN's
 *
 *   algorithms/utils/beamSearch.h:22-214   filtered_beam_search           -> pann_batch_search*
 *   algorithms/utils/beamSearch.h:353-387  searchAll   (parallel_for seam) -> pann_batch_search*
 *   algorithms/utils/beamSearch.h:537-565  qsearchAll  (parallel_for seam) -> pann_batch_search*
 *   algorithms/utils/beamSearch.h:499-521  beam_search_rerank__ (build)    -> pann_batch_search* with
 *                                          query_ids != NULL and a visited-list output
 *   algorithms/utils/euclidian_point.h:54-90, mips_point.h:43-65           -> device distance functors
 *   algorithms/vamana/index.h:63-137       knn_index::robustPrune          -> pann_robust_prune_batch
 *   algorithms/vamana/index.h:247-266      batch_insert step 1 (search+prune per inserted point)
 *                                                                          -> pann_insert_batch
 *   algorithms/HCNNG/hcnng_index.h:145-181 MSTk all-pairs + per-row 10-NN  -> pann_leaf_knn
 *   data_tools/compute_groundtruth.cpp:22-59 brute-force kNN               -> pann_bruteforce_knn
 *
 * The reference has no FFI of its own for this path (it is a header-only template library); these
 * entry points are what a cgo/ctypes/pybind binding placed at the parallel_for seams above would
 * bind.  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns an int status (PANN_OK == 0); pann_last_error() gives the text for the
 *     calling thread.  The reference prints and abort()s (beamSearch.h:38-41,368-372); host wrappers
 *     reproduce that on a non-zero status.
 *   - "host" entry points take host pointers and stage through the handle's device buffers;
 *     "_dev" entry points take device pointers (HIP) and a hipStream_t passed as void*, perform no
 *     allocation and no synchronisation, and are what bench.py times.
 *   - graph rows on the HOST side use the reference layout (graph.h:134-141,234-242):
 *     n x (max_deg+1) uint32, slot 0 = degree.  The device mirror is private to the handle.
 */
#ifndef PANN_H_
#define PANN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANN_ABI_VERSION 3

/* status codes */
#define PANN_OK 0
#define PANN_ERR_BAD_ARG 1
#define PANN_ERR_HIP 2
#define PANN_ERR_NO_DEVICE 3
#define PANN_ERR_UNSUPPORTED 4
#define PANN_ERR_OVERFLOW 5 /* a per-query device buffer (visited list) was too small */

/* element type of the stored vectors (reference: Euclidian_Point<T>/Mips_Point<T>, T in
 * {uint8_t,int8_t,float}; PANN_F16 (IEEE binary16) and PANN_BF16 (bfloat16) are this build's two-byte
 * extensions: values are widened exactly to f32, arithmetic is f32) */
typedef enum { PANN_U8 = 0, PANN_I8 = 1, PANN_F32 = 2, PANN_F16 = 3, PANN_BF16 = 4 } pann_dtype;

/* distance functor: euclidian_point.h:54-90 / mips_point.h:43-65 */
typedef enum { PANN_L2 = 0, PANN_MIPS = 1 } pann_metric;

/* field-for-field mirror of QueryParams (algorithms/utils/types.h:218-231) */
typedef struct pann_query_params {
  int64_t k;            /* QueryParams::k            (0 during build: no cut-prune) */
  int64_t beam;         /* QueryParams::beamSize */
  double cut;           /* QueryParams::cut */
  int64_t limit;        /* QueryParams::limit        (max number of visited vertices) */
  int64_t degree_limit; /* QueryParams::degree_limit */
  int32_t rerank_factor;/* QueryParams::rerank_factor (used by the rerank wrapper only) */
  float pad;            /* QueryParams::pad */
} pann_query_params;

/* Outputs of one batched search.  Any pointer may be NULL (that output is skipped).
 * ids/dists rows hold the first min(out_k, frontier size) entries of the final frontier, sorted
 * by (dist, id) (beamSearch.h:46-48,211); unused slots are 0xFFFFFFFF / +inf.
 * visited_ids/visited_dists hold the visited list (beamSearch.h:112-113) in VISIT order, at most
 * visited_cap entries per query (the reference keeps it sorted by (dist,id); callers that need
 * that order sort the row, robustPrune sorts its candidates anyway, vamana/index.h:83). */
typedef struct pann_search_out {
  uint32_t* ids;           /* nq x out_k */
  float* dists;            /* nq x out_k */
  uint32_t out_k;
  uint32_t* frontier_size; /* nq */
  uint32_t* visited_count; /* nq : visitedElts.size()  (stats.h:70-73 increment_visited) */
  uint32_t* dist_cmps;     /* nq : full_dist_cmps      (beamSearch.h:213) */
  uint32_t* degree_sum;    /* nq : sum over visited v of min(deg(v), degree_limit) -- roofline numerator */
  uint32_t* visited_ids;   /* nq x visited_cap */
  float* visited_dists;    /* nq x visited_cap */
  uint32_t visited_cap;
  uint32_t* status;        /* 1 word, optional: PANN_STATUS_* bits of this launch.  pann_batch_search_dev copies the
                              kernel's status word here on the launch stream (device pointer; read it after the stream
                              has been synchronised); the host entry points write it before returning. */
} pann_search_out;

/* bits of pann_search_out::status */
#define PANN_STATUS_VISITED_OVERFLOW 1u /* a visited list was longer than visited_cap: the lists are truncated */
#define PANN_STATUS_DROPPED_OVERFLOW 2u /* the per-query "dropped" scratch (equivalence 2) was too small:
                                           results of this launch are NOT valid.  The host entry points grow the scratch
                                           and run the batch again by themselves; after a _dev launch that reports it,
                                           call pann_index_reserve_dropped() with a larger capacity and launch again. */

typedef struct pann_index pann_index;

/* ---- library ------------------------------------------------------------------------------- */
int pann_abi_version(void);
const char* pann_last_error(void);
int pann_device_count(void);

/* ---- index handle: device mirror of PointRange (point_range.h:42-141) + Graph (graph.h:125-250) */

/* points: host slab, n rows of d elements, row stride row_stride_bytes (PointRange::aligned_bytes,
 * point_range.h:94).  graph: host n x (max_deg+1) uint32 in the reference layout, or NULL for an
 * empty graph (all degrees 0, as Graph(maxDeg,n) gives, graph.h:145-147). */
int pann_index_create(pann_index** out, const void* points, uint64_t n, uint32_t d, int dtype,
                      uint64_t row_stride_bytes, int metric, const uint32_t* graph,
                      uint32_t max_deg, int device);
/* Streaming form of the same (the reference's loaders read a whole file into host memory first --
 * point_range.h:74-117, graph.h:147-232; a shard of a 100M-point file need not exist on the host at once): an index of n zero
 * points and an empty graph, then any number of row ranges.  rows: nrows x d elements with row stride row_stride_bytes; the
 * range [first_row, first_row + nrows) must lie inside the index.  Graph rows stream through pann_index_update_rows. */
int pann_index_create_empty(pann_index** out, uint64_t n, uint32_t d, int dtype, int metric, uint32_t max_deg, int device);
int pann_index_upload_points(pann_index* idx, uint64_t first_row, const void* rows, uint64_t nrows, uint64_t row_stride_bytes);
void pann_index_destroy(pann_index* idx);

uint64_t pann_index_size(const pann_index* idx);
uint32_t pann_index_dims(const pann_index* idx);
uint32_t pann_index_max_degree(const pann_index* idx);
int pann_index_device(const pann_index* idx);

/* Validation mode for float element types (f32, f16): every distance is summed strictly left to right
 * with one rounding per subtract / multiply / add, as the reference's scalar loops do
 * (euclidian_point.h:83-90, mips_point.h:59-65), so results on REAL-valued data are bit-identical to
 * the CPU path (one lane per candidate: several times slower).  No effect on integer types. */
int pann_index_set_exact_float_order(pann_index* idx, int on);

/* Capacity (entries per query) of the scratch list that remembers visited vertices which the cut-prune
 * (beamSearch.h:190-195) dropped from a frontier that is not yet full; default 256.  Only searches with k > 0 on a
 * metric index that visit more than `cap` vertices before the frontier fills can exceed it (e.g. cut = 1.0 on a
 * path-like graph).  Never shrinks. */
int pann_index_reserve_dropped(pann_index* idx, uint32_t cap);
uint32_t pann_index_dropped_capacity(const pann_index* idx);

/* Replace the whole graph from a host n x (max_deg+1) slab.  Neighbour ids >= n are rejected with
 * PANN_ERR_BAD_ARG (the offending rows are left empty): the kernels gather points[id] unchecked. */
int pann_index_set_graph(pann_index* idx, const uint32_t* graph);
/* Replace m rows: rows is m x (max_deg+1) in the reference layout (edgeRange::update_neighbors,
 * graph.h:84-99).  Must not overlap a search on the same handle (vamana/index.h:247-270). */
int pann_index_update_rows(pann_index* idx, const uint32_t* row_ids, const uint32_t* rows,
                           uint64_t m);
/* Empty graph again (all degrees 0, as Graph(maxDeg, n) gives, graph.h:145-147) without a host slab: one fill on the handle's
 * stream.  A rebuild of the same points (bench.py's build modes) starts from it. */
int pann_index_clear_graph(pann_index* idx);
/* Copy the device graph back to a host n x (max_deg+1) slab. */
int pann_index_get_graph(pann_index* idx, uint32_t* graph_out);

/* ---- batched beam search (beamSearch.h:22-214 run for nq queries at once) -------------------- */

/* queries: nq rows of d elements of the index dtype, row stride q_stride_bytes; OR query_ids:
 * nq base-point ids (the query is Points[id] and neighbours equal to id are skipped, the
 * `Points[a].same_as(p)` test of beamSearch.h:133).  Exactly one of the two is non-NULL.
 * starts: nstarts start vertices shared by all queries (all in-scope drivers pass {0}:
 * vamana/index.h:148, check_nn_recall.h:178). */
int pann_batch_search(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                      uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts,
                      const pann_query_params* qp, const pann_search_out* out);

/* Same, all pointers are device pointers; launched on `stream` (hipStream_t); no sync, no alloc
 * except growth of the handle's private workspace on first use (call once to warm up). */
int pann_batch_search_dev(pann_index* idx, const void* d_queries, const uint32_t* d_query_ids,
                          uint64_t nq, uint64_t q_stride_bytes, const uint32_t* d_starts,
                          uint32_t nstarts, const pann_query_params* qp,
                          const pann_search_out* d_out, void* stream);

/* beamSearchRandom (beamSearch.h:309-351): like pann_batch_search, but every query has its OWN start
 * vertices: starts is nq x nstarts.  (The reference draws one random start per query from
 * parlay::random_generator; the caller supplies the draws here.)  Host pointers. */
int pann_batch_search_per_query_starts(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                                       uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts,
                                       const pann_query_params* qp, const pann_search_out* out);

/* ---- distances ------------------------------------------------------------------------------ */

/* out[i] = distance(Points[a_ids[i]], Points[b_ids[i]]), i < m  (Point::distance). host pointers */
int pann_pair_distances(pann_index* idx, const uint32_t* a_ids, const uint32_t* b_ids, uint64_t m,
                        float* out);
/* out[q*m + j] = distance(query q, Points[ids[j]])  for nq external queries. host pointers */
int pann_query_distances(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes,
                         const uint32_t* ids, uint64_t m, float* out);

/* Re-scoring of beam_search_rerank (beamSearch.h:426-452): for query i the first cand_counts[i]
 * (or c when cand_counts is NULL) ids of row i of cand_ids (nq x c) get their exact distance to
 * query i.  resort != 0: sort by (dist,id) and keep k (:437-442); resort == 0: keep the first k in
 * the given order (:447-452).  Unused output slots are 0xFFFFFFFF / +inf.  Host pointers. */
int pann_rerank(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes,
                const uint32_t* cand_ids, uint32_t c, const uint32_t* cand_counts, uint32_t k, int resort,
                uint32_t* out_ids, float* out_dists);

/* ---- robustPrune (vamana/index.h:63-137) ---------------------------------------------------- */

/* For each of m owners p_i: candidates = given list (ids, and dists to p_i; if cand_dists is NULL
 * they are computed as in the id-only overload :124-137) plus, when add_out_nbrs != 0, p_i's current
 * out-neighbours (:72-77); sort by (dist,id), unique by id, greedy alpha-prune to at most R.
 * cand_offsets has m+1 entries (CSR).  out_rows is m x (R+1) in the reference row layout
 * (slot 0 = count).  out_dist_cmps (optional) gets distance_comps per owner.  Host pointers. */
int pann_robust_prune_batch(pann_index* idx, const uint32_t* owners, uint64_t m,
                            const uint32_t* cand_ids, const float* cand_dists,
                            const uint64_t* cand_offsets, double alpha, uint32_t R,
                            int add_out_nbrs, uint32_t* out_rows, uint32_t* out_dist_cmps);

/* ---- Vamana batch_insert (vamana/index.h:188-316) ------------------------------------------- */

typedef struct pann_build_stats {
  double t_search_s, t_prune_s, t_bidirect_s, t_reprune_s; /* the reference's three phase timers, :217-222 */
  uint64_t search_dist_cmps, prune_dist_cmps, visited_total;
  /* optional host arrays of n entries each (NULL: skipped), ACCUMULATED like the reference's BuildStats
   * (stats.h:63-73): per inserted point its |visited| (vamana/index.h:262) and its beam-search + robustPrune
   * comparisons (:261,266); a re-pruned reverse-edge target gets that prune's comparisons (:298) */
  uint32_t* per_point_visited;
  uint32_t* per_point_dist_cmps;
} pann_build_stats;

/* One batch: for every id in batch_ids (m of them) beam-search from `start` with
 * QueryParams(0, L, 0.0, n, max_deg) (:250), robustPrune the visited list (:264), write the rows
 * (:268-270), then add the reverse edges: append-without-repeats when the row stays within R,
 * otherwise re-prune (:278-300).  The graph inside the handle is updated in place. */
int pann_vamana_insert_batch(pann_index* idx, const uint32_t* batch_ids, uint64_t m, uint32_t start,
                             uint32_t R, uint32_t L, double alpha, pann_build_stats* stats);

/* Whole build_index (vamana/index.h:150-186): num_passes passes of prefix-doubling batches over a
 * seeded random permutation (this build's own permutation), alpha = 1.0 on all but
 * the last pass, optional final neighbour sort by distance. */
int pann_vamana_build(pann_index* idx, uint32_t R, uint32_t L, double alpha, int num_passes,
                      uint64_t seed, int sort_neighbors, pann_build_stats* stats);

/* build_index with BP.single_batch = degree != 0 (vamana/index.h:156-170,236-240): every vertex first gets `degree` random
 * out-edges, then each pass inserts ALL points as one batch.  The reference draws the edges from parlay::random_generator over
 * [0, n] (n itself is out of range there); this build draws splitmix64(seed, i * degree + j) mod n. */
int pann_vamana_build_single_batch(pann_index* idx, uint32_t R, uint32_t L, double alpha, int num_passes, uint32_t degree,
                                   uint64_t seed, int sort_neighbors, pann_build_stats* stats);

/* The two phases of one batch on DEVICE pointers -- the seam of the multi-GPU build (SURVEY.md section 8e row 3;
 * parlayann_amd/distributed.py): with the points and the graph replicated, every rank runs phase A on its slice of the
 * batch, the slices' rows are all-gathered (ONE collective per batch, m x R x 4 bytes), and every rank applies the rows of
 * the whole batch with phase B, which is a deterministic function of (graph, batch ids, rows): all replicas stay identical
 * and equal to the single-GPU build.  Both calls run on the handle's stream and return after it has drained.
 *   A: beam search from `start` + robustPrune of the visited list (vamana/index.h:247-266) for the m ids given; reads the
 *      graph only.  d_rows_out: m x R uint32, unused slots 0xFFFFFFFF.
 *   B: write the rows (:268-270), reverse edges grouped by target, append-or-re-prune (:278-300). */
int pann_vamana_search_prune_dev(pann_index* idx, const uint32_t* d_batch_ids, uint64_t m, uint32_t start, uint32_t R, uint32_t L,
                                 double alpha, uint32_t* d_rows_out, pann_build_stats* stats);
int pann_vamana_apply_rows_dev(pann_index* idx, const uint32_t* d_batch_ids, uint64_t m, const uint32_t* d_rows, uint32_t R,
                               double alpha, pann_build_stats* stats);
/* final neighbour sort of build_index (:180-185), ties by id */
int pann_vamana_sort_neighbors(pann_index* idx);
/* host-only helpers (no device needed), so that every rank derives the same schedule: the insertion order of this build
 * (Fisher-Yates driven by splitmix64(seed), DESIGN.md "Build determinism") and the prefix-doubling batch bounds of
 * batch_insert (:206-209, :223-234; base 2, max_fraction .02) for m inserts into a graph of n vertices: bounds gets
 * (floor, ceiling) pairs, at most cap of them; returns the number of batches. */
void pann_build_permutation(uint64_t n, uint64_t seed, uint32_t* out);
uint64_t pann_vamana_batch_schedule(uint64_t n, uint64_t m, uint64_t* bounds, uint64_t cap);

/* ---- dense all-pairs: HCNNG leaf (hcnng_index.h:145-181) and ground truth ------------------- */

/* For one leaf given by N ids: for each i the m smallest (dist,id) neighbours among the other
 * leaf members.  out_ids/out_dists are N x m (row i sorted ascending).  Host pointers. */
int pann_leaf_knn(pann_index* idx, const uint32_t* ids, uint32_t N, uint32_t m, uint32_t* out_ids,
                  float* out_dists);
/* Many leaves in one call: leaf_offsets has nleaves+1 entries into ids; outputs are
 * (total ids) x m. */
int pann_leaf_knn_batch(pann_index* idx, const uint32_t* ids, const uint64_t* leaf_offsets,
                        uint64_t nleaves, uint32_t m, uint32_t* out_ids, float* out_dists);

/* HCNNG cluster-tree split (clusterEdge.h:66-83): ids is the concatenation of nseg clusters
 * (seg_offsets, nseg+1 entries); cluster s has pivots pivot_a[s], pivot_b[s];
 * out_side[i] = 0 when d(ids[i], pivot_a) <= d(ids[i], pivot_b), else 1.  Host pointers. */
int pann_pivot_split(pann_index* idx, const uint32_t* ids, const uint64_t* seg_offsets, uint64_t nseg,
                     const uint32_t* pivot_a, const uint32_t* pivot_b, uint8_t* out_side);

/* BFS range search -- range_search (algorithms/utils/beamSearch.h:245-306; caller vamana/neighbors.h:88-101).
 * Per query: every start that is not the query's own vertex and lies within radius_2 seeds `result`
 * (:271-277); then result[position++] is expanded breadth first: neighbours not yet seen (an EXACT set,
 * :256) and not the query's own vertex are remembered, cost one distance comparison each and join `result`
 * iff dist <= radius_2 (:280-297).  Exactly one of queries / query_ids is given (a base-point query skips
 * its own vertex, Point::same_as).  starts is nstarts ids (shared) or nq x nstarts (starts_per_query != 0);
 * 0xFFFFFFFF entries are padding.  out_ids is nq x max_results in BFS order, out_counts[i] <= max_results (entries of a
 * row past its count are unspecified);
 * a query whose result would exceed max_results stops there and sets out_truncated[i] = 1.
 * out_dist_cmps / out_truncated may be NULL.  (The reference's first radius argument is unused, :250.) */
int pann_range_search(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                      uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts, int starts_per_query,
                      float radius_2, uint32_t max_results, uint32_t* out_ids, uint32_t* out_counts,
                      uint32_t* out_dist_cmps, uint32_t* out_truncated);

/* Whole HCNNG build_index (hcnng_index.h:273-281) on the device: for each of num_clusters trees the
 * random two-pivot cluster tree (clusterEdge.h:99-144; level-synchronous: split kernel, prefix scan,
 * stable scatter), the all-pairs 10-NN of every leaf (hcnng_index.h:145-181), the per-leaf
 * de-duplicated, degree-bounded Kruskal (:183-228) and process_edges (:117-131).  Edges are appended to
 * the handle's graph (max_deg must be >= num_clusters * mst_deg, types.h:210-214).  Seeding rules as in
 * DESIGN.md "Build determinism".  times3 (optional): seconds spent in {tree, leaf kNN, MST}. */
int pann_hcnng_build(pann_index* idx, uint32_t num_clusters, uint32_t cluster_size, uint32_t mst_deg,
                     uint64_t seed, double* times3);
/* The stream the handle's calls run on (default: a private non-blocking stream).  A caller that produces the inputs of the
 * _dev phases on its own stream (torch's current stream: the all-gathered rows of the multi-GPU build) passes that stream here
 * once; work is then ordered by the stream and needs no synchronisation between the caller's kernels / collectives and the
 * library's.  use_private != 0: back to the private stream (stream is ignored); otherwise `stream` is used as given -- NULL is
 * the device's default stream, which is what torch's current stream usually is.  The stream must outlive its use by the handle. */
int pann_index_set_stream(pann_index* idx, void* stream, int use_private);

/* Per-handle tuning knobs; results never depend on them (0 = the library's own choice).  Unknown names: PANN_ERR_BAD_ARG.
 *   "forest_group": HCNNG -- the independent cluster trees (clusterEdge.h:146-153) are split level by level in groups of this
 *                   many trees (scratch: trees x n positions; default: as many as 2^31 positions allow)
 *   "gt_pieces"   : pann_bruteforce_knn -- the base is cut into this many pieces per 64-query tile (default: the count that
 *                   fills whole rounds of the 256 CUs best)
 *   "locality_order": 1 (default) / 0 -- on tables beyond the Infinity Cache (> 1 GB of points) the Vamana builder launches the
 *                   searches of a batch ordered by the locality cell of the inserted point (nearest of 256 pivots, one pass per
 *                   handle); the graph does not depend on the launch order.  2: also on small tables and batches (tests)
 *   "filter_codes": 1 (default) / 0 -- the Vamana builder's L = 91..128 searches keep the lossy filter (beamSearch.h:52-59) as
 *                   12-bit class codes in LDS when every slot class has fewer than 4 095 members (n below ~16M), else as ids */
int pann_index_set_option(pann_index* idx, const char* name, int64_t value);
/* the current value; for "filter_codes": 1 while the class codes are in step with the graph (the next beam-91..128 search uses
 * them), 0 otherwise; -1: unknown name */
int64_t pann_index_get_option(const pann_index* idx, const char* name);

/* Sharded index: nlists result lists per query -- the output of ONE all-gather of every shard's
 * top-k -- laid out [nlists][nq][row_stride]: row (w, q) holds k_in ids at d_ids and k_in distances at d_dists (0xFFFFFFFF =
 * unused slot).  Two separate arrays: row_stride = k_in.  One packed array of [ids | distance bits] rows (what
 * parlayann_amd.distributed gathers): row_stride = 2 * k_in, d_dists = (float*)(d_ids + k_in).  d_list_base (optional, nlists
 * words): list w holds ids LOCAL to shard w and gets d_list_base[w] added; NULL: the ids are global.  out = per query the k_out
 * smallest under (dist, id) (beamSearch.h:46-48).  Device pointers, launched on `stream` of the current device, no
 * synchronisation. */
int pann_merge_topk_dev(const uint32_t* d_ids, const float* d_dists, uint32_t nlists, uint64_t nq, uint32_t k_in,
                        uint32_t row_stride, const uint32_t* d_list_base, uint32_t k_out, uint32_t* d_out_ids, float* d_out_dists,
                        void* stream);

/* HCNNG with the TREES split over GPUs (the cluster trees are independent,
 * clusterEdge.h:146-153): every rank holds all points; rank r builds trees r, r + W, r + 2W, ... of the forest that
 * pann_hcnng_build(seed) builds (same per-tree seeds) into a device slab, the slabs are all-gathered (ONE collective) and
 * every rank interleaves them in tree order: the graph of the single-GPU build, bit for bit.
 *   build_trees: trees first_tree, first_tree + tree_step, ... (ntrees of them); tree j of the call writes vertex v's edges
 *     (at most mst_deg, hcnng_index.h:213) into slots [j*mst_deg, (j+1)*mst_deg) of row v of d_slab (device, n rows of
 *     slab_stride uint32, 0xFFFFFFFF = empty; filled here).  The handle's own graph is not touched.
 *   assemble: d_slabs = nslabs slabs one after the other (the all-gather's output); slab w holds trees w, w + nslabs, ...;
 *     row v of the handle's graph gets, after its current neighbours, the edges of trees 0 .. ntrees-1 in tree order. */
int pann_hcnng_build_trees_dev(pann_index* idx, uint32_t first_tree, uint32_t tree_step, uint32_t ntrees, uint32_t cluster_size,
                               uint32_t mst_deg, uint64_t seed, uint32_t* d_slab, uint32_t slab_stride, double* times3);
int pann_hcnng_assemble_dev(pann_index* idx, const uint32_t* d_slabs, uint32_t nslabs, uint32_t slab_stride, uint32_t ntrees,
                            uint32_t mst_deg);

/* Brute-force k nearest base points for nq external queries (compute_groundtruth.cpp:22-59):
 * out rows sorted by (dist,id). Host pointers. */
int pann_bruteforce_knn(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes,
                        uint32_t k, uint32_t* out_ids, float* out_dists);

#ifdef __cplusplus
}
#endif
#endif /* PANN_H_ */
