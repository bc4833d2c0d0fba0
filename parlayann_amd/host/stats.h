// stats.h -- host mirror of algorithms/utils/stats.h:57-94 (per-query visited / distance counters)
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace parlayANN {

template <typename indexType>
struct stats {
  stats() {}
  explicit stats(size_t n) : visited(n, 0), distances(n, 0) {}
  void increment_dist(indexType i, indexType j) { distances[i] += j; }
  void increment_visited(indexType i, indexType j) { visited[i] += j; }
  std::vector<indexType> visited_stats() const { return statistics(visited); }
  std::vector<indexType> dist_stats() const { return statistics(distances); }
  void clear() { std::fill(visited.begin(), visited.end(), 0); std::fill(distances.begin(), distances.end(), 0); }
  std::vector<indexType> visited, distances;

 private:
  static std::vector<indexType> statistics(const std::vector<indexType>& s) {   // {average, 99th percentile} :84-92
    if (s.empty()) return {0, 0};
    unsigned long long tot = 0;
    for (auto v : s) tot += v;
    std::vector<indexType> t = s;
    std::sort(t.begin(), t.end());
    return {(indexType)(tot / s.size()), t[(size_t)(.99 * (float)s.size())]};
  }
};

}  // namespace parlayANN
