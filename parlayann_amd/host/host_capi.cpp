// host_capi.cpp -- C entry points of the C++ host mirror for callers that are not C++ (the Python
// wrapper): the HCNNG builder of hcnng_index.h (host tree + Kruskal around the device calls).
#include "hcnng_index.h"

using namespace parlayANN;

template <class Point>
static int hcnng_build_t(const void* pts, uint64_t n, uint32_t d, long nc, long cs, long mst, uint64_t seed, int device,
                         uint32_t* graph_out, double* times) {
  using PR = PointRange<Point>;
  PR Points((const typename Point::T*)pts, n, d);
  Graph<unsigned int> G(nc * mst, n);
  hcnng_index<Point, PR, unsigned int> I;
  I.seed = seed;
  I.device = device;
  I.build_index_host_tree(G, Points, nc, cs, mst);
  std::memcpy(graph_out, G.data(), n * (size_t)(nc * mst + 1) * sizeof(uint32_t));
  if (times) { times[0] = I.t_tree_s; times[1] = I.t_leaf_s; times[2] = I.t_mst_s; }
  return 0;
}

extern "C" int pann_host_hcnng_build(const void* pts, uint64_t n, uint32_t d, int dtype, int metric, long num_clusters,
                                     long cluster_size, long mst_deg, uint64_t seed, int device, uint32_t* graph_out,
                                     double* times) {
#define GO(T, M) return hcnng_build_t<Point_<T, M>>(pts, n, d, num_clusters, cluster_size, mst_deg, seed, device, graph_out, times)
  if (dtype == PANN_U8 && metric == PANN_L2) GO(uint8_t, PANN_L2);
  if (dtype == PANN_U8 && metric == PANN_MIPS) GO(uint8_t, PANN_MIPS);
  if (dtype == PANN_I8 && metric == PANN_L2) GO(int8_t, PANN_L2);
  if (dtype == PANN_I8 && metric == PANN_MIPS) GO(int8_t, PANN_MIPS);
  if (dtype == PANN_F32 && metric == PANN_L2) GO(float, PANN_L2);
  if (dtype == PANN_F32 && metric == PANN_MIPS) GO(float, PANN_MIPS);
  if (dtype == PANN_F16 && metric == PANN_L2) GO(half_t, PANN_L2);
  if (dtype == PANN_F16 && metric == PANN_MIPS) GO(half_t, PANN_MIPS);
  if (dtype == PANN_BF16 && metric == PANN_L2) GO(bf16_t, PANN_L2);
  if (dtype == PANN_BF16 && metric == PANN_MIPS) GO(bf16_t, PANN_MIPS);
#undef GO
  return 1;
}
