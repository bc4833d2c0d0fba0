// device_index.h -- RAII owner of a pann_index: an EXPLICIT device mirror of (PointRange, Graph), for callers that
// want to manage the device copy themselves.  The functions with the reference's own argument lists
// (beam_search(p, G, Points, ...), qsearchAll(...), knn_index::build_index(G, Points, ...)) find their mirror
// through device_mirror.h instead; both forms end in the same C-ABI calls.  Errors follow the
// reference's convention: print and abort() (beamSearch.h:38-41,368-372; graph.h:56-58).
#pragma once
#include <cstdlib>
#include <iostream>

#include "device_mirror.h"

namespace parlayANN {

template <class PointRange, typename indexType = unsigned int>
struct DeviceIndex {
  using Point = typename PointRange::Point;
  pann_index* h = nullptr;

  // uploads Points and (if given) G; with G == nullptr the device graph starts empty with max_deg
  DeviceIndex(const PointRange& Points, const Graph<indexType>* G, long max_deg = 0, int device = 0) {
    const long md = G ? G->max_degree() : max_deg;
    pann_check(pann_index_create(&h, Points.data(), Points.size(), (uint32_t)Points.dimension(),
                                 pann_dtype_of<typename Point::T>::value, Points.get_aligned_bytes(), Point::metric,
                                 G ? G->data() : nullptr, (uint32_t)md, device));
  }
  ~DeviceIndex() { if (h) pann_index_destroy(h); }
  DeviceIndex(const DeviceIndex&) = delete;
  DeviceIndex& operator=(const DeviceIndex&) = delete;

  void upload_graph(const Graph<indexType>& G) { pann_check(pann_index_set_graph(h, G.data())); }
  void download_graph(Graph<indexType>& G) { pann_check(pann_index_get_graph(h, G.data())); }
  size_t size() const { return pann_index_size(h); }
  long max_degree() const { return pann_index_max_degree(h); }
};

}  // namespace parlayANN
