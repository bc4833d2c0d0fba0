// vamana/neighbors.h -- host mirror of algorithms/vamana/neighbors.h: the operator entry of the Vamana plugin,
//   ANN<Point, PointRange, indexType>(G, k, BP, Query_Points, GT, res_file, graph_built, Points)          :112-116
//   ANN_Quantized<PR, QPR, QQPR, indexType>(G, k, BP, Query_Points, Q_Query_Points, QQ_Query_Points, GT, res_file,
//                                           graph_built, Points, Q_Points, QQ_Points)                       :42-110
// Same argument lists, same printed report; build, search and range search run on the device mirrors.
// BP.quantize: 0 (none) and 1 (one-byte build + first search pass, full-precision rerank) are mirrored; the
// bit / JL sketches of modes 2-5 (:126-183) are out of scope.
#pragma once
#include <algorithm>
#include <chrono>

#include "../beam_search.h"
#include "../check_nn_recall.h"
#include "../parse_results.h"
#include "../quantize.h"
#include "../stats.h"
#include "../types.h"
#include "../vamana_index.h"

namespace parlayANN {

namespace vamana_driver {

inline double seconds_since(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
}

// "-self -range" (:86-104): every base point range-searches from its own vertex -- ONE launch over all points.  same_as()
// skips that start, so upstream reports 0 edges; BP.use_existing (this build's switch for the commented branch
// beamSearch.h:260-262) seeds with the point's out-neighbours instead.  Prints the reference's report lines.
template <typename PointRange, typename indexType>
void self_range_report(Graph<indexType>& G, PointRange& Points, const BuildParams& BP, long build_num_distances) {
  const auto t0 = std::chrono::steady_clock::now();
  std::cout << "radius = " << BP.radius << " radius_2 = " << BP.radius_2 << std::endl;
  const size_t n = Points.size();
  const uint32_t per_point = BP.use_existing ? (uint32_t)G.max_degree() : 1;
  std::vector<uint32_t> starts(n * (size_t)per_point, 0xFFFFFFFFu);
  for (size_t v = 0; v < n; v++) {
    if (!BP.use_existing) { starts[v] = (uint32_t)v; continue; }
    auto nbrs = G[(indexType)v];
    for (size_t j = 0; j < nbrs.size(); j++) starts[v * per_point + j] = nbrs[(indexType)j];
  }
  std::vector<long> found, cmps;
  {
    auto mirror = device_mirror(G, Points);
    std::tie(found, cmps) = self_range_search(mirror.h(), n, starts, per_point, (float)BP.radius_2);
  }
  std::cout << "range search time: " << seconds_since(t0) << std::endl;
  long edges = 0, range_num_distances = 0;
  for (size_t v = 0; v < n; v++) { edges += found[v]; range_num_distances += cmps[v]; }
  std::cout << "edges within range: " << edges << std::endl;
  std::cout << "distance comparisons during build = " << build_num_distances << std::endl;
  std::cout << "distance comparisons during range = " << range_num_distances << std::endl;
}

}  // namespace vamana_driver

// The plugin's operator (:42-110).  Three steps: (1) build the graph on the build-precision points unless one was loaded;
// (2) print the reference's graph report; (3) the query sweep, or the self range search.
template <typename PointRange, typename QPointRange, typename QQPointRange, typename indexType>
void ANN_Quantized(Graph<indexType>& G, long k, BuildParams& BP, PointRange& Query_Points, QPointRange& Q_Query_Points,
                   QQPointRange& QQ_Query_Points, groundTruth<indexType> GT, char* res_file, bool graph_built, PointRange& Points,
                   QPointRange& Q_Points, QQPointRange& QQ_Points) {
  stats<unsigned int> BuildStats(G.size());
  indexType start_point = 0;
  double idx_time = 0;
  if (!graph_built) {
    const auto t_build = std::chrono::steady_clock::now();
    knn_index<QPointRange, QQPointRange, indexType> I(BP);
    I.seed = BP.seed;
    I.build_index(G, Q_Points, QQ_Points, BuildStats);
    start_point = I.get_start();
    idx_time = vamana_driver::seconds_since(t_build);
  }
  std::cout << "start index = " << start_point << std::endl;

  const auto visited = BuildStats.visited_stats();
  std::cout << "Average visited: " << visited[0] << ", Tail visited: " << visited[1] << std::endl;
  const auto [avg_deg, max_deg] = graph_stats_(G);
  Graph_ G_("Vamana", "R = " + std::to_string(BP.R) + ", L = " + std::to_string(BP.L), G.size(), avg_deg, max_deg, idx_time);
  G_.print();

  if (Query_Points.size() != 0) {
    search_and_parse(G_, G, Points, Query_Points, Q_Points, Q_Query_Points, QQ_Points, QQ_Query_Points, GT, res_file, k, false,
                     start_point, BP.verbose, BP.Q, BP.rerank_factor);
  } else if (BP.self && BP.range) {
    long build_num_distances = 0;
    for (auto x : BuildStats.distances) build_num_distances += (long)x;
    vamana_driver::self_range_report(G, Points, BP, build_num_distances);
  }
}

template <typename Point, typename PointRange_, typename indexType>
void ANN(Graph<indexType>& G, long k, BuildParams& BP, PointRange_& Query_Points, groundTruth<indexType> GT, char* res_file,
         bool graph_built, PointRange_& Points) {
  if (BP.quantize != 0) {
    std::cout << "quantizing build and first pass of search to 1 byte" << std::endl;
    if (BP.quantize != 1) { std::cout << "Error: -quantize_mode " << BP.quantize << " (bit / JL sketches) is not mirrored; modes 0 and 1 are" << std::endl; abort(); }
    if constexpr (!std::is_same<typename Point::T, float>::value) {
      std::cout << "Error: -quantize_mode needs float points" << std::endl; abort();
    } else if constexpr (Point::metric == PANN_L2) {
      using QPR = PointRange<Euclidian_Point<uint8_t>>;                      // :119-123
      const euclid_u8_parameters pm = generate_parameters_u8(Points);
      QPR Q_Points = quantize_u8(Points, pm);
      QPR Q_Query_Points = quantize_u8(Query_Points, pm);
      ANN_Quantized(G, k, BP, Query_Points, Q_Query_Points, Q_Query_Points, GT, res_file, graph_built, Points, Q_Points, Q_Points);
    } else {
      using QPR = PointRange<Mips_Point<int8_t>>;                            // Quantized_Mips_Point<8,true,255> (:146-149)
      const float mv = generate_max_val_mips_i8(Points, true);
      QPR Q_Points = quantize_mips_i8(Points, mv);
      QPR Q_Query_Points = quantize_mips_i8(Query_Points, mv);
      ANN_Quantized(G, k, BP, Query_Points, Q_Query_Points, Q_Query_Points, GT, res_file, graph_built, Points, Q_Points, Q_Points);
    }
  } else {
    ANN_Quantized(G, k, BP, Query_Points, Query_Points, Query_Points, GT, res_file, graph_built, Points, Points, Points);
  }
}

}  // namespace parlayANN
