// Entry points declared in include/pann.h that are not implemented yet fail loudly.
#include "pann_internal.h"
using namespace pann;
#define PANN_TODO(name) set_error(name ": not implemented yet"); return PANN_ERR_UNSUPPORTED
extern "C" {
int pann_pair_distances(pann_index*, const uint32_t*, const uint32_t*, uint64_t, float*) { PANN_TODO("pann_pair_distances"); }
int pann_query_distances(pann_index*, const void*, uint64_t, uint64_t, const uint32_t*, uint64_t, float*) { PANN_TODO("pann_query_distances"); }
int pann_robust_prune_batch(pann_index*, const uint32_t*, uint64_t, const uint32_t*, const float*, const uint64_t*, double, uint32_t, int, uint32_t*, uint32_t*) { PANN_TODO("pann_robust_prune_batch"); }
int pann_vamana_insert_batch(pann_index*, const uint32_t*, uint64_t, uint32_t, uint32_t, uint32_t, double, pann_build_stats*) { PANN_TODO("pann_vamana_insert_batch"); }
int pann_vamana_build(pann_index*, uint32_t, uint32_t, double, int, uint64_t, int, pann_build_stats*) { PANN_TODO("pann_vamana_build"); }
int pann_leaf_knn(pann_index*, const uint32_t*, uint32_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_leaf_knn"); }
int pann_leaf_knn_batch(pann_index*, const uint32_t*, const uint64_t*, uint64_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_leaf_knn_batch"); }
int pann_bruteforce_knn(pann_index*, const void*, uint64_t, uint64_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_bruteforce_knn"); }
}
