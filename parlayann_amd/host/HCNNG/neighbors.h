// HCNNG/neighbors.h -- host mirror of algorithms/HCNNG/neighbors.h:39-63: the operator entry of the HCNNG plugin,
//   ANN<Point, PointRange, indexType>(G, k, BP, Query_Points, GT, res_file, graph_built, Points)
#pragma once
#include <chrono>

#include "../beam_search.h"
#include "../check_nn_recall.h"
#include "../hcnng_index.h"
#include "../parse_results.h"
#include "../stats.h"
#include "../types.h"

namespace parlayANN {

template <typename Point, typename PointRange, typename indexType>
void ANN(Graph<indexType>& G, long k, BuildParams& BP, PointRange& Query_Points, groundTruth<indexType> GT, char* res_file,
         bool graph_built, PointRange& Points) {
  double idx_time = 0;
  if (!graph_built) {
    const auto t_build = std::chrono::steady_clock::now();
    hcnng_index<Point, PointRange, indexType> I;
    I.seed = BP.seed;
    if (BP.host_tree) I.build_index_host_tree(G, Points, BP.num_clusters, BP.cluster_size, BP.MST_deg);
    else I.build_index(G, Points, BP.num_clusters, BP.cluster_size, BP.MST_deg);
    idx_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
    std::cout << "tree time: " << I.t_tree_s << " leaf knn time: " << I.t_leaf_s << " mst time: " << I.t_mst_s << std::endl;
  }
  const auto [avg_deg, max_deg] = graph_stats_(G);
  Graph_ G_("HCNNG", "Trees = " + std::to_string(BP.num_clusters), G.size(), avg_deg, max_deg, idx_time);
  G_.print();
  if (Query_Points.size() != 0) search_and_parse(G_, G, Points, Query_Points, GT, res_file, k, BP.verbose, BP.Q);
}

}  // namespace parlayANN
