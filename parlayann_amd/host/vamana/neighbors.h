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

template <typename PointRange, typename QPointRange, typename QQPointRange, typename indexType>
void ANN_Quantized(Graph<indexType>& G, long k, BuildParams& BP, PointRange& Query_Points, QPointRange& Q_Query_Points,
                   QQPointRange& QQ_Query_Points, groundTruth<indexType> GT, char* res_file, bool graph_built, PointRange& Points,
                   QPointRange& Q_Points, QQPointRange& QQ_Points) {
  const auto t0 = std::chrono::steady_clock::now();
  bool verbose = BP.verbose;
  using findex = knn_index<QPointRange, QQPointRange, indexType>;
  findex I(BP);
  I.seed = BP.seed;
  indexType start_point;
  double idx_time;
  stats<unsigned int> BuildStats(G.size());
  if (graph_built) {
    idx_time = 0;
    start_point = 0;
  } else {
    I.build_index(G, Q_Points, QQ_Points, BuildStats);
    start_point = I.get_start();
    idx_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  std::cout << "start index = " << start_point << std::endl;

  std::string name = "Vamana";
  std::string params = "R = " + std::to_string(BP.R) + ", L = " + std::to_string(BP.L);
  auto [avg_deg, max_deg] = graph_stats_(G);
  auto vv = BuildStats.visited_stats();
  std::cout << "Average visited: " << vv[0] << ", Tail visited: " << vv[1] << std::endl;
  Graph_ G_(name, params, G.size(), avg_deg, max_deg, idx_time);
  G_.print();

  long build_num_distances = 0;
  for (auto x : BuildStats.distances) build_num_distances += (long)x;

  if (Query_Points.size() != 0) {
    search_and_parse(G_, G, Points, Query_Points, Q_Points, Q_Query_Points, QQ_Points, QQ_Query_Points, GT, res_file, k, false,
                     start_point, verbose, BP.Q, BP.rerank_factor);
  } else if (BP.self) {
    if (BP.range) {
      // :86-104: every base point range-searches from its own vertex -- ONE launch over all points.  same_as() skips
      // that start, so upstream reports 0 edges; BP.use_existing (this build's switch for the commented branch
      // beamSearch.h:260-262) seeds with the point's out-neighbours instead.
      const auto tr = std::chrono::steady_clock::now();
      double radius = BP.radius;
      double radius_2 = BP.radius_2;
      std::cout << "radius = " << radius << " radius_2 = " << radius_2 << std::endl;
      const size_t n = Points.size();
      const uint32_t ns = BP.use_existing ? (uint32_t)G.max_degree() : 1;
      std::vector<uint32_t> starts(n * (size_t)ns, 0xFFFFFFFFu);
      for (size_t i = 0; i < n; i++) {
        if (!BP.use_existing) { starts[i] = (uint32_t)i; continue; }
        auto row = G[(indexType)i];
        for (size_t j = 0; j < row.size(); j++) starts[i * ns + j] = row[(indexType)j];
      }
      std::vector<long> counts, distance_comps;
      {
        auto L = device_mirror(G, Points);
        std::tie(counts, distance_comps) = self_range_search(L.h(), n, starts, ns, (float)radius_2);
      }
      std::cout << "range search time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tr).count() << std::endl;
      long range_num_distances = 0, edges = 0;
      for (size_t i = 0; i < n; i++) { edges += counts[i]; range_num_distances += distance_comps[i]; }
      std::cout << "edges within range: " << edges << std::endl;
      std::cout << "distance comparisons during build = " << build_num_distances << std::endl;
      std::cout << "distance comparisons during range = " << range_num_distances << std::endl;
    }
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
