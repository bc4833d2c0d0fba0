// neighborsTime.cpp -- the `neighbors` driver of algorithms/bench/neighborsTime.C, built once per algorithm directory
// like upstream (vamana/neighbors, HCNNG/neighbors: the Makefile selects the plugin's neighbors.h):
//   main (:73-252) parses the reference's flags, loads base / query / ground truth / graph, applies -normalize and
//   -quantize_bits 8, then timeNeighbors<Point, PointRange, indexType> (:50-70) -> ANN<...>(G, k, BP, Query_Points, GT,
//   res_file, graph_built, Points) -> G.save(-graph_outfile).
// Flags as upstream: -base_path -query_path -gt_path -graph_path -graph_outfile -res_path -data_type {uint8,int8,float}
//   -dist_func {Euclidian,mips} -k -Q -R -L -alpha -num_passes -two_pass -mst_deg -num_clusters -cluster_size -delta
//   -quantize_bits {0,8} -quantize_mode {0,1} -verbose -normalize -self -range -radius -radius_2 -rerank_factor
//   (-quantize_bits 16 and -quantize_mode 2..5 are rejected: out of scope).
// Added here: -device <ordinal>, -seed <s>, -use_existing (self range search seeded with out-neighbours),
//   -host_tree (HCNNG cross-check path).
#include <chrono>
#include <cstring>
#include <string>

#ifdef PANN_ALG_HCNNG
#include "../HCNNG/neighbors.h"
#else
#include "../vamana/neighbors.h"
#endif
#include "../quantize.h"
#include "parse_command_line.h"

using namespace parlayANN;
using uint = unsigned int;

template <typename Point, typename PointRange, typename indexType>
void timeNeighbors(Graph<indexType>& G, PointRange& Query_Points, long k, BuildParams& BP, char* outFile, groundTruth<indexType> GT,
                   char* res_file, bool graph_built, PointRange& Points) {
  const auto t0 = std::chrono::steady_clock::now();                  // time_loop(1, 0, ...) of bench/time_loop.h: one timed round
  ANN<Point, PointRange, indexType>(G, k, BP, Query_Points, GT, res_file, graph_built, Points);
  std::cout << "ANN: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
  if (outFile != NULL) G.save(outFile);
}

template <typename Point>
void run_plain(char* iFile, char* qFile, char* gFile, char* oFile, char* rFile, long maxDeg, long k, BuildParams& BP,
               groundTruth<uint>& GT, bool graph_built, bool normalize, int quantize) {
  using PR = PointRange<Point>;
  PR Points(iFile);
  PR Query_Points(qFile);
  Graph<unsigned int> G;
  if (gFile == NULL) G = Graph<unsigned int>(maxDeg, Points.size());
  else G = Graph<unsigned int>(gFile);
  if constexpr (std::is_same<typename Point::T, float>::value) {
    if (normalize) {                                                   // :147-153
      std::cout << "normalizing data" << std::endl;
      normalize_range(Points);
      normalize_range(Query_Points);
    }
    if (quantize == 8) {                                               // :157-164, :190-197
      std::cout << "quantizing data to 1 byte" << std::endl;
      if constexpr (Point::metric == PANN_L2) {
        using QPoint = Euclidian_Point<uint8_t>;
        using QPR = PointRange<QPoint>;
        const euclid_u8_parameters pm = generate_parameters_u8(Points);
        QPR Points_ = quantize_u8(Points, pm);
        QPR Query_Points_ = quantize_u8(Query_Points, pm);
        timeNeighbors<QPoint, QPR, uint>(G, Query_Points_, k, BP, oFile, GT, rFile, graph_built, Points_);
      } else {
        using QPoint = Mips_Point<int8_t>;                             // Quantized_Mips_Point<8>: trim = false (:193)
        using QPR = PointRange<QPoint>;
        const float mv = generate_max_val_mips_i8(Points, false);
        QPR Points_ = quantize_mips_i8(Points, mv);
        QPR Query_Points_ = quantize_mips_i8(Query_Points, mv);
        timeNeighbors<QPoint, QPR, uint>(G, Query_Points_, k, BP, oFile, GT, rFile, graph_built, Points_);
      }
      return;
    }
  }
  timeNeighbors<Point, PR, uint>(G, Query_Points, k, BP, oFile, GT, rFile, graph_built, Points);
}

int main(int argc, char* argv[]) {
  commandLine P(argc, argv,
                "[-a <alpha>] [-d <delta>] [-R <deg>]"
                "[-L <bm>] [-k <k> ]  [-gt_path <g>] [-query_path <qF>]"
                "[-graph_path <gF>] [-graph_outfile <oF>] [-res_path <rF>]" "[-num_passes <np>]"
                "[-memory_flag <algoOpt>] [-mst_deg <q>] [-num_clusters <nc>] [-cluster_size <cs>]"
                "[-data_type <tp>] [-dist_func <df>] [-base_path <b>] [-device <d>] [-seed <s>] <inFile>");

  char* iFile = P.getOptionValue("-base_path");
  char* oFile = P.getOptionValue("-graph_outfile");
  char* gFile = P.getOptionValue("-graph_path");
  char* qFile = P.getOptionValue("-query_path");
  char* cFile = P.getOptionValue("-gt_path");
  char* rFile = P.getOptionValue("-res_path");
  char* vectype = P.getOptionValue("-data_type");
  long Q = P.getOptionIntValue("-Q", 0);
  long R = P.getOptionIntValue("-R", 0);
  if (R < 0) P.badArgument();
  long L = P.getOptionIntValue("-L", 0);
  if (L < 0) P.badArgument();
  long MST_deg = P.getOptionIntValue("-mst_deg", 0);
  if (MST_deg < 0) P.badArgument();
  long num_clusters = P.getOptionIntValue("-num_clusters", 0);
  if (num_clusters < 0) P.badArgument();
  long cluster_size = P.getOptionIntValue("-cluster_size", 0);
  if (cluster_size < 0) P.badArgument();
  double radius = P.getOptionDoubleValue("-radius", 0.0);
  double radius_2 = P.getOptionDoubleValue("-radius_2", radius);
  long k = P.getOptionIntValue("-k", 0);
  if (k > 1000 || k < 0) P.badArgument();
  double alpha = P.getOptionDoubleValue("-alpha", 1.0);
  int num_passes = P.getOptionIntValue("-num_passes", 1);
  int two_pass = P.getOptionIntValue("-two_pass", 0);
  if (two_pass > 1 || two_pass < 0) P.badArgument();
  if (two_pass == 1) num_passes = 2;
  double delta = P.getOptionDoubleValue("-delta", 0);
  if (delta < 0) P.badArgument();
  char* dfc = P.getOptionValue("-dist_func");
  int quantize = P.getOptionIntValue("-quantize_bits", 0);
  int quantize_build = P.getOptionIntValue("-quantize_mode", 0);
  bool verbose = P.getOption("-verbose");
  bool normalize = P.getOption("-normalize");
  double trim = P.getOptionDoubleValue("-trim", 0.0);  // not used
  bool self = P.getOption("-self");
  int rerank_factor = P.getOptionIntValue("-rerank_factor", 100);
  bool range = P.getOption("-range");
  int single_batch = P.getOptionIntValue("-single_batch", 0);

  if (!iFile || !dfc || !vectype) P.badArgument();
  std::string df = std::string(dfc);
  std::string tp = std::string(vectype);

  BuildParams BP = BuildParams(R, L, alpha, num_passes, num_clusters, cluster_size, MST_deg, delta, verbose, quantize_build, radius,
                               radius_2, self, range, single_batch, Q, trim, rerank_factor);
  BP.seed = (uint64_t)P.getOptionLongValue("-seed", 1);
  BP.use_existing = P.getOption("-use_existing");
  BP.host_tree = P.getOption("-host_tree");
  set_default_device(P.getOptionIntValue("-device", 0));
#ifdef PANN_ALG_HCNNG
  if (BP.alg_type != "HCNNG") { std::cout << "Error: HCNNG needs -num_clusters, -cluster_size and -mst_deg" << std::endl; abort(); }
#else
  if (BP.alg_type != "Vamana") { std::cout << "Error: Vamana needs -R, -L and -alpha" << std::endl; abort(); }
#endif
  long maxDeg = BP.max_degree();

  if ((tp != "uint8") && (tp != "int8") && (tp != "float")) {
    std::cout << "Error: vector type not specified correctly, specify int8, uint8, or float" << std::endl;
    abort();
  }
  if (df != "Euclidian" && df != "mips") {
    std::cout << "Error: specify distance type Euclidian or mips" << std::endl;
    abort();
  }
  if (quantize != 0 && !(quantize == 8 && tp == "float")) {
    std::cout << "Error: -quantize_bits supports 8 with -data_type float (16 is not mirrored)" << std::endl;
    abort();
  }

  bool graph_built = (gFile != NULL);
  groundTruth<uint> GT = groundTruth<uint>(cFile);

  if (tp == "float") {
    if (df == "Euclidian") run_plain<Euclidian_Point<float>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, normalize, quantize);
    else run_plain<Mips_Point<float>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, normalize, quantize);
  } else if (tp == "uint8") {
    if (df == "Euclidian") run_plain<Euclidian_Point<uint8_t>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, false, 0);
    else run_plain<Mips_Point<uint8_t>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, false, 0);
  } else if (tp == "int8") {
    if (df == "Euclidian") run_plain<Euclidian_Point<int8_t>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, false, 0);
    else run_plain<Mips_Point<int8_t>>(iFile, qFile, gFile, oFile, rFile, maxDeg, k, BP, GT, graph_built, false, 0);
  }
  return 0;
}
