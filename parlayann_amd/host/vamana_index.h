// vamana_index.h -- host mirror of algorithms/vamana/index.h: knn_index (:42-318).
// build_index / batch_insert / robustPrune keep their names and argument meaning; the work runs in
// libpann.so (pann_vamana_build / pann_vamana_insert_batch / pann_robust_prune_batch).
#pragma once
#include <utility>
#include <vector>

#include "device_index.h"
#include "stats.h"
#include "types.h"

namespace parlayANN {

template <typename PointRange, typename indexType>
struct knn_index {
  using Point = typename PointRange::Point;
  using distanceType = float;
  using pid = std::pair<indexType, distanceType>;
  using GraphI = Graph<indexType>;
  using DI = DeviceIndex<PointRange, indexType>;

  BuildParams BP;
  indexType start_point = 0;
  uint64_t seed = 1;          // insertion-order seed (DESIGN.md "Build determinism")
  pann_build_stats last{};

  explicit knn_index(BuildParams& BP) : BP(BP) {}
  indexType get_start() { return start_point; }
  void set_start() { start_point = 0; }                                        // :148

  // robustPrune(p, cand (id,dist), ...) -> (new neighbours, distance_comps)    (:63-120)
  std::pair<std::vector<indexType>, long> robustPrune(indexType p, std::vector<pid>& cand, DI& D, double alpha, bool add = true) {
    std::vector<uint32_t> ids(cand.size()); std::vector<float> ds(cand.size());
    for (size_t i = 0; i < cand.size(); i++) { ids[i] = cand[i].first; ds[i] = cand[i].second; }
    return prune_one(p, ids, &ds, D, alpha, add);
  }
  // id-only overload (:124-137)
  std::pair<std::vector<indexType>, long> robustPrune(indexType p, std::vector<indexType> candidates, DI& D, double alpha, bool add = true) {
    std::vector<uint32_t> ids(candidates.begin(), candidates.end());
    return prune_one(p, ids, nullptr, D, alpha, add);
  }

  // build_index(G, Points, BuildStats, sort_neighbors)   (:150-186)
  void build_index(GraphI& G, PointRange& Points, stats<indexType>& BuildStats, bool sort_neighbors = true) {
    std::cout << "Building graph..." << std::endl;
    set_start();
    DI D(Points, &G);
    std::cout << "number of passes = " << BP.num_passes << std::endl;
    last = pann_build_stats{};
    pann_check(pann_vamana_build(D.h, (uint32_t)BP.R, (uint32_t)BP.L, BP.alpha, BP.num_passes, seed, sort_neighbors ? 1 : 0, &last));
    D.download_graph(G);
    std::cout << "beam search time: " << last.t_search_s << std::endl;        // the reference's phase timers (:313-315)
    std::cout << "bidirect time: " << last.t_bidirect_s << std::endl;
    std::cout << "prune time: " << last.t_prune_s + last.t_reprune_s << std::endl;
    // per-point counters are aggregated on the device; spread the averages so the stats keep their meaning
    const size_t n = Points.size();
    if (BuildStats.visited.size() == n && n) {
      for (size_t i = 0; i < n; i++) {
        BuildStats.increment_visited((indexType)i, (indexType)(last.visited_total / n));
        BuildStats.increment_dist((indexType)i, (indexType)((last.search_dist_cmps + last.prune_dist_cmps) / n));
      }
    }
  }

  // one batch of inserts against the device graph (:188-316, steps 1-4)
  void batch_insert(const std::vector<indexType>& inserts, DI& D, double alpha) {
    pann_check(pann_vamana_insert_batch(D.h, inserts.data(), inserts.size(), start_point, (uint32_t)BP.R, (uint32_t)BP.L, alpha, &last));
  }

 private:
  std::pair<std::vector<indexType>, long> prune_one(indexType p, const std::vector<uint32_t>& ids, const std::vector<float>* ds,
                                                    DI& D, double alpha, bool add) {
    const uint64_t off[2] = {0, ids.size()};
    std::vector<uint32_t> row(BP.R + 1);
    uint32_t dc = 0;
    pann_check(pann_robust_prune_batch(D.h, &p, 1, ids.data(), ds ? ds->data() : nullptr, off, alpha, (uint32_t)BP.R, add ? 1 : 0,
                                       row.data(), &dc));
    std::vector<indexType> out(row.begin() + 1, row.begin() + 1 + row[0]);
    return std::make_pair(out, (long)dc);
  }
};

}  // namespace parlayANN
