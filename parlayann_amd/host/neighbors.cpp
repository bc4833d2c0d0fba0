// neighbors.cpp -- example driver with the reference CLI's flag names (algorithms/bench/neighborsTime.C
// :80-121): builds (Vamana or HCNNG) or loads a graph, runs the fixed-beam query of the `-Q` path
// (check_nn_recall.h:221-226) and prints recall / QPS.  Flags: -base_path -query_path -gt_path
// -graph_path -graph_outfile -data_type {uint8,int8,float} -dist_func {Euclidian,mips} -k -Q -R -L
// -alpha -num_passes -cluster_size -mst_deg -num_clusters -alg {vamana,hcnng} -device -seed -device_build {1,0}
// -self 1 -range 1 -radius -radius_2 [-use_existing 1]  (vamana/neighbors.h:86-104)
// -quantize_bits 8 with -data_type float  (neighborsTime.C:157-164,190-197)
// -quantize_mode 1 [-rerank_factor 100] with -data_type float  (vamana/neighbors.h:117-147)
// -res_path <csv> (sweep mode: one row per recall bucket, check_nn_recall.h:127-158)
// -two_pass 1 (= -num_passes 2), -normalize 1 (float data)  (neighborsTime.C:104-106,113,147-153)
#include <cstring>
#include <map>
#include <string>
#include <type_traits>

#include "check_nn_recall.h"
#include "hcnng_index.h"
#include "quantize.h"
#include "vamana_index.h"

using namespace parlayANN;
using indexType = unsigned int;

struct Args {
  std::map<std::string, std::string> kv;
  Args(int argc, char** argv) { for (int i = 1; i + 1 < argc; i += 2) kv[argv[i]] = argv[i + 1]; }
  const char* str(const char* k) const { auto it = kv.find(k); return it == kv.end() ? nullptr : it->second.c_str(); }
  long num(const char* k, long d) const { auto s = str(k); return s ? atol(s) : d; }
  double flt(const char* k, double d) const { auto s = str(k); return s ? atof(s) : d; }
};

template <class Point>
int run_on(const Args& a, PointRange<Point>& Points, PointRange<Point>* QueriesIn);

template <class Point>
int run(const Args& a) {
  using PR = PointRange<Point>;
  const char* base = a.str("-base_path");
  if (!base) { std::cout << "usage: neighbors -base_path <b> [-graph_path <g>] [-query_path <q> -gt_path <gt>] ..." << std::endl; return 1; }
  PR Points(base);
  const bool norm = a.num("-normalize", 0) != 0;                                   // neighborsTime.C:113,147-153 (float only)
  if constexpr (std::is_same<typename Point::T, float>::value) { if (norm) { std::cout << "normalizing data" << std::endl; normalize_range(Points); } }
  if (a.str("-query_path")) {
    PR Queries(a.str("-query_path"));
    if constexpr (std::is_same<typename Point::T, float>::value) { if (norm) normalize_range(Queries); }
    return run_on<Point>(a, Points, &Queries);
  }
  return run_on<Point>(a, Points, nullptr);
}

// -data_type float -quantize_bits 8 (neighborsTime.C:157-164, 190-197): base and queries are translated with the
// BASE range's parameters, everything downstream (build, search) sees only the one-byte points
template <bool MIPS>
int run_quantized(const Args& a) {
  using FP = typename std::conditional<MIPS, Mips_Point<float>, Euclidian_Point<float>>::type;
  const char* base = a.str("-base_path");
  if (!base) { std::cout << "usage: neighbors -base_path <b> ..." << std::endl; return 1; }
  PointRange<FP> Points(base);
  std::cout << "quantizing data to 1 byte" << std::endl;
  if constexpr (MIPS) {
    const float mv = generate_max_val_mips_i8(Points, false);              // Quantized_Mips_Point<8>: trim = false (:193)
    auto QP = quantize_mips_i8(Points, mv);
    if (a.str("-query_path")) { PointRange<FP> Qf(a.str("-query_path")); auto QQ = quantize_mips_i8(Qf, mv); return run_on<Mips_Point<int8_t>>(a, QP, &QQ); }
    return run_on<Mips_Point<int8_t>>(a, QP, nullptr);
  } else {
    const euclid_u8_parameters pm = generate_parameters_u8(Points);
    auto QP = quantize_u8(Points, pm);
    if (a.str("-query_path")) { PointRange<FP> Qf(a.str("-query_path")); auto QQ = quantize_u8(Qf, pm); return run_on<Euclidian_Point<uint8_t>>(a, QP, &QQ); }
    return run_on<Euclidian_Point<uint8_t>>(a, QP, nullptr);
  }
}

// -quantize_mode 1 (BuildParams::quantize == 1, vamana/neighbors.h:117-147): the index is built on the one-byte
// points; queries search the one-byte mirror first and the best k * rerank_factor are re-scored on the float mirror
// (ANN_Quantized -> search_and_parse -> qsearchAll -> beam_search_rerank).  MIPS uses Quantized_Mips_Point<8,true,255>.
template <bool MIPS>
int run_quantize_mode(const Args& a) {
  using FP = typename std::conditional<MIPS, Mips_Point<float>, Euclidian_Point<float>>::type;
  using QPt = typename std::conditional<MIPS, Mips_Point<int8_t>, Euclidian_Point<uint8_t>>::type;
  using PR = PointRange<FP>; using QPR = PointRange<QPt>;
  const char* base = a.str("-base_path");
  if (!base) { std::cout << "usage: neighbors -base_path <b> ..." << std::endl; return 1; }
  PR Points(base);
  std::cout << "quantizing build and first pass of search to 1 byte" << std::endl;
  float mv = 0; euclid_u8_parameters pm;
  auto quant = [&](PR& src) -> QPR {
    if constexpr (MIPS) return quantize_mips_i8(src, mv); else return quantize_u8(src, pm);
  };
  if constexpr (MIPS) mv = generate_max_val_mips_i8(Points, true); else pm = generate_parameters_u8(Points);
  QPR Q_Points = quant(Points);
  const long k = a.num("-k", 10), Q = a.num("-Q", 0);
  BuildParams BP(a.num("-R", 64), a.num("-L", 128), a.flt("-alpha", 1.2), (int)(a.num("-two_pass", 0) == 1 ? 2 : a.num("-num_passes", 1)));
  Graph<indexType> G;
  if (a.str("-graph_path")) {
    G = Graph<indexType>(a.str("-graph_path"));
  } else {
    G = Graph<indexType>(BP.max_degree(), Points.size());
    knn_index<QPR, indexType> I(BP); I.seed = (uint64_t)a.num("-seed", 1);
    stats<indexType> BuildStats(Points.size());
    I.build_index(G, Q_Points, BuildStats);
    if (a.str("-graph_outfile")) G.save(a.str("-graph_outfile"));
  }
  if (a.str("-query_path")) {
    PR Queries(a.str("-query_path"));
    QPR Q_Queries = quant(Queries);
    groundTruth<indexType> GT(a.str("-gt_path"));
    DeviceIndex<PR, indexType> DI(Points, &G, 0, (int)a.num("-device", 0));
    DeviceIndex<QPR, indexType> QDI(Q_Points, &G, 0, (int)a.num("-device", 0));
    search_and_parse([&](const QueryParams& QP) { return checkRecall<PR, QPR, indexType>(DI, QDI, Queries, Q_Queries, GT, 0, k == 0 ? 10 : k, QP, Q != 0 || a.num("-verbose", 0)); },
                     G.size(), (long)G.max_degree(), k, Q, (int)a.num("-rerank_factor", 100));
  }
  return 0;
}

template <class Point>
int run_on(const Args& a, PointRange<Point>& Points, PointRange<Point>* QueriesIn) {
  using PR = PointRange<Point>;
  const std::string alg = a.str("-alg") ? a.str("-alg") : "vamana";
  const long k = a.num("-k", 10), Q = a.num("-Q", 0);            // 0: sweep (neighborsTime.C: -Q default 0)
  BuildParams BP;
  if (alg == "hcnng") BP = BuildParams(a.num("-num_clusters", 30), a.num("-cluster_size", 1000), a.num("-mst_deg", 3));
  else BP = BuildParams(a.num("-R", 64), a.num("-L", 128), a.flt("-alpha", 1.2), (int)(a.num("-two_pass", 0) == 1 ? 2 : a.num("-num_passes", 1)));
  Graph<indexType> G;
  double build_time = 0;
  if (a.str("-graph_path")) {
    G = Graph<indexType>(a.str("-graph_path"));
  } else {
    G = Graph<indexType>(BP.max_degree(), Points.size());
    const auto t0 = std::chrono::steady_clock::now();
    if (alg == "hcnng") {
      hcnng_index<Point, PR, indexType> I; I.seed = (uint64_t)a.num("-seed", 1);
      if (a.num("-device_build", 1)) I.build_index_on_device(G, Points, BP.num_clusters, BP.cluster_size, BP.MST_deg);
      else I.build_index(G, Points, BP.num_clusters, BP.cluster_size, BP.MST_deg);   // host tree + Kruskal around the device calls
      std::cout << "tree time: " << I.t_tree_s << " leaf knn time: " << I.t_leaf_s << " mst time: " << I.t_mst_s << std::endl;
    } else {
      knn_index<PR, indexType> I(BP); I.seed = (uint64_t)a.num("-seed", 1);
      stats<indexType> BuildStats(Points.size());
      I.build_index(G, Points, BuildStats);
    }
    build_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "ANN: " << build_time << std::endl;
    if (a.str("-graph_outfile")) G.save(a.str("-graph_outfile"));
  }
  size_t tot = 0, mx = 0;
  for (size_t i = 0; i < G.size(); i++) { tot += G[(indexType)i].size(); mx = std::max(mx, G[(indexType)i].size()); }
  std::cout << "Graph has average degree " << (double)tot / G.size() << " and maximum degree " << mx << std::endl;
  if (QueriesIn) {
    PR& Queries = *QueriesIn;
    groundTruth<indexType> GT(a.str("-gt_path"));
    DeviceIndex<PR, indexType> DI(Points, &G, 0, (int)a.num("-device", 0));
    // -Q given: five repetitions at that beam (:224-229); otherwise the reference's sweep + recall-bucket table
    Graph_ G_;                                                    // vamana/neighbors.h:65-73, HCNNG/neighbors.h
    G_.name = alg == "hcnng" ? "HCNNG" : "Vamana";
    G_.params = alg == "hcnng" ? "Trees = " + std::to_string(BP.num_clusters) : "R = " + std::to_string(BP.R) + ", L = " + std::to_string(BP.L);
    G_.size = (long)G.size(); G_.avg_deg = (double)tot / G.size(); G_.max_deg = (int)mx; G_.time = build_time;
    G_.print();
    search_and_parse([&](const QueryParams& QP) { return checkRecall<PR, indexType>(DI, Queries, GT, 0, k == 0 ? 10 : k, QP, Q != 0 || a.num("-verbose", 0)); },
                     G.size(), (long)G.max_degree(), k, Q, 100, a.str("-res_path"), &G_);
  } else if (a.num("-self", 0) && a.num("-range", 0)) {
    // vamana/neighbors.h:86-104: every base point range-searches from its own vertex.  same_as() skips that
    // start, so upstream reports 0 edges here; `-use_existing 1` seeds with the point's out-neighbours instead
    // (the commented branch beamSearch.h:260-262)
    const float radius = (float)a.flt("-radius", 0.0), radius_2 = (float)a.flt("-radius_2", 0.0);
    std::cout << "radius = " << radius << " radius_2 = " << radius_2 << std::endl;
    DeviceIndex<PR, indexType> DI(Points, &G, 0, (int)a.num("-device", 0));
    const size_t n = Points.size();
    const bool existing = a.num("-use_existing", 0) != 0;
    const uint32_t ns = existing ? (uint32_t)G.max_degree() : 1;
    std::vector<uint32_t> starts(n * (size_t)ns, 0xFFFFFFFFu);
    for (size_t i = 0; i < n; i++) {
      if (!existing) { starts[i] = (uint32_t)i; continue; }
      auto row = G[(indexType)i];
      for (size_t j = 0; j < row.size(); j++) starts[i * ns + j] = row[(indexType)j];
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto [counts, cmps] = self_range_search<PR, indexType>(DI, n, starts, ns, radius_2);
    std::cout << "range search time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
    long edges = 0, dc = 0;
    for (size_t i = 0; i < n; i++) { edges += counts[i]; dc += cmps[i]; }
    std::cout << "edges within range: " << edges << std::endl;
    std::cout << "distance comparisons during range = " << dc << std::endl;
  }
  return 0;
}

int main(int argc, char** argv) {
  Args a(argc, argv);
  const std::string dt = a.str("-data_type") ? a.str("-data_type") : "uint8";
  const std::string df = a.str("-dist_func") ? a.str("-dist_func") : "Euclidian";
  if (df != "Euclidian" && df != "mips") { std::cout << "Error: specify distance type Euclidian or mips" << std::endl; abort(); }
  const bool mips = df == "mips";
  if (dt == "uint8") return mips ? run<Mips_Point<uint8_t>>(a) : run<Euclidian_Point<uint8_t>>(a);
  if (dt == "int8") return mips ? run<Mips_Point<int8_t>>(a) : run<Euclidian_Point<int8_t>>(a);
  if (a.num("-quantize_mode", 0) == 1 && dt == "float") return mips ? run_quantize_mode<true>(a) : run_quantize_mode<false>(a);
  if (a.num("-quantize_mode", 0) != 0) { std::cout << "Error: -quantize_mode 1 (one-level) with -data_type float is the mode mirrored here" << std::endl; abort(); }
  const long qbits = a.num("-quantize_bits", 0);
  if (dt == "float" && qbits == 8) return mips ? run_quantized<true>(a) : run_quantized<false>(a);
  if (qbits != 0) { std::cout << "Error: -quantize_bits supports 8 with -data_type float (16 is not mirrored)" << std::endl; abort(); }
  if (dt == "float") return mips ? run<Mips_Point<float>>(a) : run<Euclidian_Point<float>>(a);
  std::cout << "Error: data type not specified correctly, specify int8, uint8, or float" << std::endl;   // neighborsTime.C:143-150
  abort();
}
