// vamana_index.h -- host mirror of algorithms/vamana/index.h: knn_index<PointRange, QPointRange, indexType> (:42-318)
// with the reference's member names and argument lists:
//   robustPrune(p, cand, G, Points, alpha, add)            :63-65    robustPrune(p, candidates, G, Points, alpha, add)  :124-126
//   build_index(G, Points, QPoints, BuildStats, sort)      :150-151  batch_insert(inserts, G, Points, QPoints, BuildStats, alpha, ...) :188-192
// The work runs in libpann.so (pann_robust_prune_batch / pann_vamana_build / pann_vamana_insert_batch) on the device
// mirror of (G, Points); G is read back after every call that changes it, so host and device graphs agree on return.
// QPoints is the reference's second-level filter range: only its num_bytes() is looked at (it must equal Points',
// i.e. no use_filtering, beamSearch.h:515-519).
#pragma once
#include <cmath>
#include <utility>
#include <vector>

#include "device_index.h"
#include "parlay_compat.h"
#include "stats.h"
#include "types.h"

namespace parlayANN {

template <typename PointRange, typename QPointRange, typename indexType>
struct knn_index {
  using Point = typename PointRange::Point;
  using QPoint = typename QPointRange::Point;
  using distanceType = float;
  using pid = std::pair<indexType, distanceType>;
  using PR = PointRange;
  using QPR = QPointRange;
  using GraphI = Graph<indexType>;

  BuildParams BP;
  indexType start_point = 0;
  uint64_t seed = 1;          // insertion-order seed (DESIGN.md "Build determinism": parlay::random_permutation is not reproducible)
  pann_build_stats last{};    // phase timers and totals of the last build_index / batch_insert

  knn_index(BuildParams& BP) : BP(BP) {}
  indexType get_start() { return start_point; }
  void set_start() { start_point = 0; }                                        // :148

  // robustPrune(p, cand (id, dist), G, Points, alpha, add) -> (new out-neighbours of p, distance_comps)    (:63-120)
  std::pair<parlay::sequence<indexType>, long> robustPrune(indexType p, parlay::sequence<pid>& cand, GraphI& G, PR& Points,
                                                           double alpha, bool add = true) {
    std::vector<uint32_t> ids(cand.size()); std::vector<float> ds(cand.size());
    for (size_t i = 0; i < cand.size(); i++) { ids[i] = cand[i].first; ds[i] = cand[i].second; }
    auto L = device_mirror(G, Points);
    return prune_one(L.h(), p, ids, &ds, alpha, add);
  }
  // candidates without distances (:124-137): the distances to p are computed first and counted
  std::pair<parlay::sequence<indexType>, long> robustPrune(indexType p, parlay::sequence<indexType> candidates, GraphI& G, PR& Points,
                                                           double alpha, bool add = true) {
    std::vector<uint32_t> ids(candidates.begin(), candidates.end());
    auto L = device_mirror(G, Points);
    return prune_one(L.h(), p, ids, nullptr, alpha, add);
  }

  // build_index(G, Points, QPoints, BuildStats, sort_neighbors)   (:150-186)
  void build_index(GraphI& G, PR& Points, QPR& QPoints, stats<indexType>& BuildStats, bool sort_neighbors = true) {
    std::cout << "Building graph..." << std::endl;
    set_start();
    check_ranges(Points, QPoints);
    if (BP.single_batch != 0)
      std::cout << "Using single batch per round with " << BP.single_batch << " random start edges" << std::endl;      // :158
    std::cout << "number of passes = " << BP.num_passes << std::endl;
    last = pann_build_stats{};
    attach_stats(BuildStats, Points.size());
    // G holds the starting graph (empty after Graph(maxDeg, n); a loaded graph is extended, as upstream)
    auto L = device_mirror(G, Points);
    if (BP.single_batch != 0)      // :156-170,236-240 (this build's own generator for the start edges: DESIGN.md section 6)
      pann_check(pann_vamana_build_single_batch(L.h(), (uint32_t)BP.R, (uint32_t)BP.L, BP.alpha, BP.num_passes,
                                                (uint32_t)BP.single_batch, seed, sort_neighbors ? 1 : 0, &last));
    else
      pann_check(pann_vamana_build(L.h(), (uint32_t)BP.R, (uint32_t)BP.L, BP.alpha, BP.num_passes, seed, sort_neighbors ? 1 : 0, &last));
    MirrorCache::download_graph(L, G);
    detach_stats(BuildStats);
    std::cout << "beam search time: " << last.t_search_s << std::endl;        // the reference's phase timers (:313-315)
    std::cout << "bidirect time: " << last.t_bidirect_s << std::endl;
    std::cout << "prune time: " << last.t_prune_s + last.t_reprune_s << std::endl;
  }

  // batch_insert(inserts, G, Points, QPoints, BuildStats, alpha, random_order, base, max_fraction, print)   (:188-316):
  // prefix-doubling batches (sizes base^i while <= max_batch, then max_batch) over the (optionally shuffled) inserts
  void batch_insert(parlay::sequence<indexType>& inserts, GraphI& G, PR& Points, QPR& QPoints, stats<indexType>& BuildStats,
                    double alpha, bool random_order = false, double base = 2, double max_fraction = .02, bool print = true) {
    for (indexType p : inserts)
      if ((long)p > (long)G.size()) { std::cout << "ERROR: invalid point " << p << " given to batch_insert" << std::endl; abort(); }   // :193-198
    check_ranges(Points, QPoints);
    const size_t n = G.size(), m = inserts.size();
    size_t max_batch_size = std::min(static_cast<size_t>(max_fraction * static_cast<float>(n)), (size_t)1000000ul);   // :206-207
    if (max_batch_size == 0) max_batch_size = n;                                                                        // :209
    std::vector<indexType> shuffled(inserts.begin(), inserts.end());
    if (random_order) shuffle(shuffled);
    attach_stats(BuildStats, Points.size());
    auto L = device_mirror(G, Points);
    size_t inc = 0, count = 0;
    float frac = 0.0f; const float progress_inc = .1f;
    while (count < m) {                                                                                                 // :223-234
      size_t floor, ceiling;
      if (std::pow(base, (double)inc) <= (double)max_batch_size) {
        floor = static_cast<size_t>(std::pow(base, (double)inc)) - 1;
        ceiling = std::min(static_cast<size_t>(std::pow(base, (double)(inc + 1))) - 1, m);
        count = std::min(static_cast<size_t>(std::pow(base, (double)(inc + 1))) - 1, m);
      } else {
        floor = count;
        ceiling = std::min(count + max_batch_size, m);
        count += max_batch_size;
      }
      if (ceiling > floor)
        pann_check(pann_vamana_insert_batch(L.h(), shuffled.data() + floor, ceiling - floor, start_point, (uint32_t)BP.R, (uint32_t)BP.L,
                                            alpha, &last));
      if (print) {                                                                                                      // :302-308
        const auto ind = frac * n;
        if (floor <= ind && ceiling > ind) { frac += progress_inc; std::cout << "Pass " << 100 * frac << "% complete" << std::endl; }
      }
      inc += 1;
    }
    MirrorCache::download_graph(L, G);
    detach_stats(BuildStats);
    if (print) {
      std::cout << "beam search time: " << last.t_search_s << std::endl;
      std::cout << "bidirect time: " << last.t_bidirect_s << std::endl;
      std::cout << "prune time: " << last.t_prune_s + last.t_reprune_s << std::endl;
    }
  }

  // ---- the same on an explicitly managed mirror (device_index.h) ----
  using DI = DeviceIndex<PointRange, indexType>;
  std::pair<parlay::sequence<indexType>, long> robustPrune(indexType p, parlay::sequence<pid>& cand, DI& D, double alpha, bool add = true) {
    std::vector<uint32_t> ids(cand.size()); std::vector<float> ds(cand.size());
    for (size_t i = 0; i < cand.size(); i++) { ids[i] = cand[i].first; ds[i] = cand[i].second; }
    return prune_one(D.h, p, ids, &ds, alpha, add);
  }
  void batch_insert(const std::vector<indexType>& inserts, DI& D, double alpha) {
    pann_check(pann_vamana_insert_batch(D.h, inserts.data(), inserts.size(), start_point, (uint32_t)BP.R, (uint32_t)BP.L, alpha, &last));
  }

 private:
  void check_ranges(PR& Points, QPR& QPoints) {
    if (Points.params.num_bytes() != QPoints.params.num_bytes()) {
      std::cout << "Error: building with a second-level filter range (use_filtering, beamSearch.h:515-519) is not mirrored on the device" << std::endl;
      abort();
    }
  }
  // BuildStats counters are filled per inserted point by the library (include/pann.h: pann_build_stats::per_point_*)
  void attach_stats(stats<indexType>& BuildStats, size_t n) {
    static_assert(sizeof(indexType) == 4, "per-point counters are 32-bit");
    last.per_point_visited = BuildStats.visited.size() == n ? (uint32_t*)BuildStats.visited.data() : nullptr;
    last.per_point_dist_cmps = BuildStats.distances.size() == n ? (uint32_t*)BuildStats.distances.data() : nullptr;
  }
  void detach_stats(stats<indexType>&) { last.per_point_visited = nullptr; last.per_point_dist_cmps = nullptr; }

  // this build's insertion order: Fisher-Yates driven by splitmix64(seed), the rule of pann_vamana_build
  void shuffle(std::vector<indexType>& v) {
    uint64_t s = seed;
    auto next = [&]() {
      uint64_t z = (s += 0x9e3779b97f4a7c15ull);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      return z ^ (z >> 31);
    };
    for (uint64_t i = v.size(); i > 1; i--) std::swap(v[i - 1], v[next() % i]);
  }

  std::pair<parlay::sequence<indexType>, long> prune_one(pann_index* h, indexType p, const std::vector<uint32_t>& ids,
                                                         const std::vector<float>* ds, double alpha, bool add) {
    const uint64_t off[2] = {0, ids.size()};
    std::vector<uint32_t> row(BP.R + 1);
    uint32_t dc = 0;
    const uint32_t owner = p;
    pann_check(pann_robust_prune_batch(h, &owner, 1, ids.data(), ds ? ds->data() : nullptr, off, alpha, (uint32_t)BP.R, add ? 1 : 0,
                                       row.data(), &dc));
    parlay::sequence<indexType> out(row.begin() + 1, row.begin() + 1 + row[0]);
    return std::make_pair(out, (long)dc);
  }
};

}  // namespace parlayANN
