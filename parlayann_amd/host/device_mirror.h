// device_mirror.h -- device copies of (PointRange, Graph) pairs, found by the host objects themselves.
//
// The reference's free functions take `(G, Points)` by reference and read them on the CPU
// (beamSearch.h:217-223, vamana/index.h:150-151, ...).  The mirror functions keep those argument lists; what
// they need on the GPU -- a pann_index holding the points and the adjacency rows -- is looked up here by the
// identity of the two host slabs and kept until either slab is freed:
//
//   key      (points slab, graph slab, device)
//   valid    while both slabs are alive (weak handles: a recycled address never matches an old entry)
//   fresh    Graph / PointRange carry a change counter bumped by every edgeRange mutation (graph.h) and by
//            PointRange::touch(); a stale graph is re-uploaded (pann_index_set_graph), stale points rebuild the entry
//
// One mirror serves one call at a time (the C-ABI stages through per-handle buffers): acquire() returns the handle
// together with its lock.  Errors follow the reference: print and abort() (beamSearch.h:38-41, graph.h:56-58).
#pragma once
#include <cstdlib>
#include <functional>
#include <iostream>
#include <list>
#include <memory>
#include <mutex>

#include "../../include/pann.h"
#include "graph.h"
#include "point_range.h"

namespace parlayANN {

inline void pann_check(int rc) {
  if (rc != PANN_OK) {
    std::cout << pann_last_error() << std::endl;
    abort();
  }
}

inline int& default_device() { static int d = 0; return d; }       // `-device` of the CLI
inline void set_default_device(int d) { default_device() = d; }

struct DeviceMirror {
  pann_index* h = nullptr;
  const void* pts = nullptr; const void* graph = nullptr; int device = 0;
  uint64_t pts_version = 0, graph_version = 0;
  std::function<bool()> alive;
  std::mutex busy;
  ~DeviceMirror() { if (h) pann_index_destroy(h); }
};

// handle + the mirror's lock for the duration of one C-ABI call sequence
struct MirrorLease {
  std::shared_ptr<DeviceMirror> m;
  std::unique_lock<std::mutex> lk;
  pann_index* h() const { return m->h; }
};

class MirrorCache {
 public:
  static MirrorCache& get() { static MirrorCache c; return c; }

  // upload = false: the caller is about to overwrite the device graph (a build) -- a stale host graph is not sent
  template <class PointRange, typename indexType>
  MirrorLease acquire(const Graph<indexType>& G, const PointRange& Points, bool upload_graph = true, int device = default_device()) {
    using Point = typename PointRange::Point;
    if (G.size() != Points.size()) {
      std::cout << "ERROR: graph has " << G.size() << " vertices but the point range has " << Points.size() << std::endl;
      abort();
    }
    std::shared_ptr<DeviceMirror> m;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto it = entries_.begin(); it != entries_.end();) {            // slabs that were freed take their mirrors along
        if (!(*it)->alive()) it = entries_.erase(it); else ++it;
      }
      for (auto& e : entries_)
        if (e->pts == (const void*)Points.data() && e->graph == (const void*)G.data() && e->device == device &&
            e->pts_version == Points.version()) { m = e; break; }
      if (!m) {
        for (auto it = entries_.begin(); it != entries_.end(); ++it)       // same slabs, points changed in place: rebuild
          if ((*it)->pts == (const void*)Points.data() && (*it)->graph == (const void*)G.data() && (*it)->device == device) { entries_.erase(it); break; }
        m = std::make_shared<DeviceMirror>();
        m->pts = Points.data(); m->graph = G.data(); m->device = device; m->pts_version = Points.version();
        std::weak_ptr<typename PointRange::byte[]> wp = Points.slab_handle();
        std::weak_ptr<indexType[]> wg = G.slab_handle();
        m->alive = [wp, wg]() { return !wp.expired() && !wg.expired(); };
        pann_check(pann_index_create(&m->h, Points.data(), Points.size(), (uint32_t)Points.dimension(),
                                     pann_dtype_of<typename Point::T>::value, Points.get_aligned_bytes(), Point::metric,
                                     upload_graph ? G.data() : nullptr, (uint32_t)G.max_degree(), device));
        m->graph_version = upload_graph ? G.version() : 0;
        entries_.push_back(m);
      }
    }
    MirrorLease L{m, std::unique_lock<std::mutex>(m->busy)};
    if (upload_graph && m->graph_version != G.version()) {
      pann_check(pann_index_set_graph(m->h, G.data()));
      m->graph_version = G.version();
    }
    return L;
  }

  // device graph -> host G (after a build); the mirror then counts as in sync with G
  template <typename indexType>
  static void download_graph(MirrorLease& L, Graph<indexType>& G) {
    pann_check(pann_index_get_graph(L.h(), G.data()));
    G.touch();
    L.m->graph_version = G.version();
  }

  void clear() { std::lock_guard<std::mutex> g(mu_); entries_.clear(); }
  size_t size() { std::lock_guard<std::mutex> g(mu_); return entries_.size(); }

 private:
  std::mutex mu_;
  std::list<std::shared_ptr<DeviceMirror>> entries_;
};

template <class PointRange, typename indexType>
inline MirrorLease device_mirror(const Graph<indexType>& G, const PointRange& Points, bool upload_graph = true) {
  return MirrorCache::get().acquire(G, Points, upload_graph);
}
inline void release_device_mirrors() { MirrorCache::get().clear(); }

// the query's own vertex when p is a view into Points (Point::same_as is pointer equality, euclidian_point.h:178-180), else -1
template <class PointRange>
inline long own_vertex(const typename PointRange::Point& p, const PointRange& Points) {
  const uint8_t* v = (const uint8_t*)p.values;
  const uint8_t* lo = Points.data();
  if (!lo || v < lo) return -1;
  const size_t off = (size_t)(v - lo);
  if (off % Points.get_aligned_bytes() != 0 || off / Points.get_aligned_bytes() >= Points.size()) return -1;
  return (long)(off / Points.get_aligned_bytes());
}

}  // namespace parlayANN
