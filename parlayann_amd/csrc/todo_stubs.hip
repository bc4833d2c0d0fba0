// Entry points declared in include/pann.h that are not implemented yet fail loudly.
#include "pann_internal.h"
using namespace pann;
#define PANN_TODO(name) set_error(name ": not implemented yet"); return PANN_ERR_UNSUPPORTED
extern "C" {
int pann_pair_distances(pann_index*, const uint32_t*, const uint32_t*, uint64_t, float*) { PANN_TODO("pann_pair_distances"); }
int pann_query_distances(pann_index*, const void*, uint64_t, uint64_t, const uint32_t*, uint64_t, float*) { PANN_TODO("pann_query_distances"); }
int pann_leaf_knn(pann_index*, const uint32_t*, uint32_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_leaf_knn"); }
int pann_leaf_knn_batch(pann_index*, const uint32_t*, const uint64_t*, uint64_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_leaf_knn_batch"); }
int pann_bruteforce_knn(pann_index*, const void*, uint64_t, uint64_t, uint32_t, uint32_t*, float*) { PANN_TODO("pann_bruteforce_knn"); }
}
