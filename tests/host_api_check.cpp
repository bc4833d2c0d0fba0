// host_api_check.cpp -- a test TU that calls the host mirror with the REFERENCE's argument lists and orders (file:line
// per call below) and dumps what comes back as little-endian arrays in <outdir>/<name>.bin; tests/test_host_api_gpu.py
// compares every array with the oracle.  usage: host_api_check <base.bin> <query.bin> <graph> <gt.ibin> <outdir>
#include <cstdio>
#include <string>
#include <vector>

#include "../parlayann_amd/host/HCNNG/neighbors.h"          // ANN of the HCNNG plugin is checked through its driver; here: hcnng_index
#include "../parlayann_amd/host/beam_search.h"
#include "../parlayann_amd/host/check_nn_recall.h"
#include "../parlayann_amd/host/vamana_index.h"

using namespace parlayANN;
using indexType = unsigned int;
using Point = Euclidian_Point<uint8_t>;
using PR = PointRange<Point>;

static std::string g_out;
template <typename T>
static void dump(const std::string& name, const std::vector<T>& v) {
  FILE* f = std::fopen((g_out + "/" + name + ".bin").c_str(), "wb");
  if (!f) { std::printf("cannot write %s\n", name.c_str()); std::abort(); }
  if (!v.empty()) std::fwrite(v.data(), sizeof(T), v.size(), f);
  std::fclose(f);
}
template <typename R>
static void dump_beam(const std::string& name, const R& r) {     // ((frontier, visited), dist_cmps)
  std::vector<uint32_t> fi, vi; std::vector<float> fd, vd;
  for (auto& p : r.first.first) { fi.push_back(p.first); fd.push_back(p.second); }
  for (auto& p : r.first.second) { vi.push_back(p.first); vd.push_back(p.second); }
  dump(name + "_frontier_ids", fi); dump(name + "_frontier_dists", fd);
  dump(name + "_visited_ids", vi); dump(name + "_visited_dists", vd);
  dump(name + "_cmps", std::vector<uint64_t>{(uint64_t)r.second});
}
static std::vector<uint32_t> flat(const parlay::sequence<parlay::sequence<indexType>>& a) {
  std::vector<uint32_t> o;
  for (auto& r : a) o.insert(o.end(), r.begin(), r.end());
  return o;
}

int main(int argc, char** argv) {
  if (argc < 6) { std::printf("usage: host_api_check base query graph gt outdir\n"); return 2; }
  g_out = argv[5];
  PR Points(argv[1]);
  PR Query_Points(argv[2]);
  Graph<indexType> G(argv[3]);
  groundTruth<indexType> GT(argv[4]);
  const long n = (long)Points.size();

  // ---- beam_search(p, G, Points, starting_points, QP)   beamSearch.h:217-223 ----
  QueryParams QP(10, 64, 1.35, n, G.max_degree());
  parlay::sequence<indexType> starts = {0, 5, 9};
  dump_beam("bs_ext", beam_search(Query_Points[3], G, Points, starts, QP));
  // a base point as the query: Point::same_as skips its own vertex (:133)
  dump_beam("bs_base", beam_search(Points[77], G, Points, starts, QP));
  // single start (:234-241), beam_search_impl (:226-231), filtered_beam_search (:22-33)
  dump_beam("bs_single", beam_search(Query_Points[4], G, Points, (indexType)0, QP));
  { parlay::sequence<indexType> s0 = {0}; QueryParams q2 = QP; dump_beam("bs_impl", beam_search_impl(Query_Points[4], G, Points, s0, q2)); }
  { parlay::sequence<indexType> s0 = {0};
    dump_beam("bs_filtered", filtered_beam_search(G, Query_Points[5], Points, Query_Points[5], Points, s0, QP, false)); }
  // build-time form: k = 0, beam = L, returns the visited list (:499-521, vamana/index.h:250-259)
  {
    QueryParams BQ((long)0, 48, (double)0.0, n, G.max_degree());
    auto r = beam_search_rerank__(Points[123], Points[123], G, Points, Points, (indexType)0, BQ);
    std::vector<uint32_t> vi; std::vector<float> vd;
    for (auto& p : r.first) { vi.push_back(p.first); vd.push_back(p.second); }
    dump("build_visited_ids", vi); dump("build_visited_dists", vd); dump("build_visited_cmps", std::vector<uint32_t>{r.second});

    // ---- knn_index::robustPrune(p, cand, G, Points, alpha, add)   vamana/index.h:63-65 and :124-126 ----
    BuildParams BP(32, 48, 1.2, 1);
    knn_index<PR, PR, indexType> I(BP);
    auto cand = r.first;
    auto rp = I.robustPrune((indexType)123, cand, G, Points, 1.2, true);
    dump("prune_pairs_row", std::vector<uint32_t>(rp.first.begin(), rp.first.end()));
    dump("prune_pairs_cmps", std::vector<uint64_t>{(uint64_t)rp.second});
    parlay::sequence<indexType> ids;
    for (auto& p : r.first) ids.push_back(p.first);
    auto rp2 = I.robustPrune((indexType)123, ids, G, Points, 1.0, false);
    dump("prune_ids_row", std::vector<uint32_t>(rp2.first.begin(), rp2.first.end()));
    dump("prune_ids_cmps", std::vector<uint64_t>{(uint64_t)rp2.second});
  }

  // ---- searchAll(Query_Points, G, Base_Points, QueryStats, starting_point(s), QP)   :353-387 ----
  {
    stats<indexType> QS(Query_Points.size());
    QueryParams q2 = QP;
    auto all = searchAll<PR, indexType>(Query_Points, G, Points, QS, (indexType)0, q2);
    dump("searchAll_ids", flat(all)); dump("searchAll_visited", QS.visited); dump("searchAll_dists", QS.distances);
    stats<indexType> QS2(Query_Points.size());
    auto all2 = searchAll<PR, indexType>(Query_Points, G, Points, QS2, starts, q2);
    dump("searchAll3_ids", flat(all2));
  }
  // ---- qsearchAll<PR,QPR,QQPR,indexType>(Query_Points, Q_Query_Points, QQ_Query_Points, G, Base_Points, Q_Base_Points,
  //                                        QQ_Base_Points, QueryStats, starting_point, QP)   :537-548 ----
  {
    stats<indexType> QS(Query_Points.size());
    auto all = qsearchAll<PR, PR, PR, indexType>(Query_Points, Query_Points, Query_Points, G, Points, Points, Points, QS, (indexType)0, QP);
    dump("qsearchAll_ids", flat(all)); dump("qsearchAll_visited", QS.visited);
  }
  // ---- beamSearchRandom(Query_Points, G, Base_Points, QueryStats, QP)   :309-351 (starts: this build's draws) ----
  {
    stats<indexType> QS(Query_Points.size());
    auto all = beamSearchRandom(Query_Points, G, Points, QS, QP);
    dump("random_ids", flat(all));
    std::vector<uint32_t> st(Query_Points.size());
    for (size_t i = 0; i < st.size(); i++) st[i] = (uint32_t)detail::random_start(i, G.size());
    dump("random_starts", st);
  }
  // ---- beam_search_rerank(p, qp, qqp, G, Base_Points, Q_Base_Points, QQ_Base_Points, QueryStats, starting_points, QP)   :390-454
  {
    stats<indexType> QS(Query_Points.size());
    parlay::sequence<indexType> s0 = {0};
    auto r = beam_search_rerank(Query_Points[6], Query_Points[6], Query_Points[6], G, Points, Points, Points, QS, s0, QP);
    std::vector<uint32_t> ri; std::vector<float> rd;
    for (auto& p : r) { ri.push_back(p.first); rd.push_back(p.second); }
    dump("rerank_ids", ri); dump("rerank_dists", rd);
  }
  // ---- range_search(p, G, Points, starting_points, radius, radius_2, QP, use_existing)   :245-252 ----
  {
    // starts: two vertices inside the radius (only those seed the BFS, :271-277) and one outside
    parlay::sequence<indexType> s0 = {(indexType)std::atol(argc > 7 ? argv[7] : "0"), (indexType)std::atol(argc > 8 ? argv[8] : "17"), 0};
    QueryParams q2 = QP;
    const float r2 = (float)std::atof(argc > 6 ? argv[6] : "50000");
    auto r = range_search(Query_Points[2], G, Points, s0, 0.0f, r2, q2, false);
    dump("range_ids", std::vector<uint32_t>(r.first.begin(), r.first.end()));
    dump("range_cmps", std::vector<uint64_t>{(uint64_t)r.second});
  }
  // ---- checkRecall(G, Base, Query, Q_Base, Q_Query, QQ_Base, QQ_Query, GT, random, start_point, k, QP, verbose)   check_nn_recall.h:17-30
  {
    nn_result N = checkRecall<PR, PR, PR, indexType>(G, Points, Query_Points, Points, Query_Points, Points, Query_Points, GT, false, 0, 10, QP, true);
    dump("recall", std::vector<double>{N.recall, (double)N.avg_visited, (double)N.avg_cmps, (double)N.tail_visited, (double)N.tail_cmps});
  }

  // ---- knn_index<PR,QPR,indexType>::build_index(G, Points, QPoints, BuildStats, sort_neighbors)   vamana/index.h:150-151 ----
  {
    BuildParams BP(32, 48, 1.2, 2);
    knn_index<PR, PR, indexType> I(BP);
    I.seed = 7;
    Graph<indexType> G2(BP.max_degree(), Points.size());
    stats<indexType> BuildStats(G2.size());
    I.build_index(G2, Points, Points, BuildStats, true);
    G2.save((g_out + "/built.graph").c_str());
    dump("build_visited", BuildStats.visited); dump("build_dists", BuildStats.distances);
    // the host graph and its mirror agree: a search through the verbatim API needs no re-upload and sees the new graph
    stats<indexType> QS(Query_Points.size());
    auto all = qsearchAll<PR, PR, PR, indexType>(Query_Points, Query_Points, Query_Points, G2, Points, Points, Points, QS, I.get_start(), QP);
    dump("built_search_ids", flat(all));
    // a host-side edit of the graph is seen by the next call (version counter of graph.h)
    parlay::sequence<indexType> none;
    G2[0].update_neighbors(none);
    stats<indexType> QS3(Query_Points.size());
    QueryParams q3 = QP;      // searchAll: qsearchAll aborts when fewer than k vertices are reachable (beamSearch.h:416-419)
    auto cut = searchAll<PR, indexType>(Query_Points, G2, Points, QS3, (indexType)0, q3);
    dump("isolated_start_ids", flat(cut));

    // ---- batch_insert(inserts, G, Points, QPoints, BuildStats, alpha, random_order, base, max_fraction, print)   :188-192 ----
    Graph<indexType> G3(BP.max_degree(), Points.size());
    stats<indexType> BS3(G3.size());
    knn_index<PR, PR, indexType> I3(BP);
    I3.seed = 7; I3.set_start();
    parlay::sequence<indexType> inserts(Points.size());
    for (size_t i = 0; i < inserts.size(); i++) inserts[i] = (indexType)i;
    I3.batch_insert(inserts, G3, Points, Points, BS3, 1.0, true, 2, .02, false);
    I3.batch_insert(inserts, G3, Points, Points, BS3, 1.2, true, 2, .02, false);
    G3.save((g_out + "/batch_insert.graph").c_str());     // == build_index with sort_neighbors = false
  }
  // ---- hcnng_index<Point,PointRange,indexType>::build_index(G, Points, cluster_rounds, cluster_size, MSTDeg)   hcnng_index.h:273-274
  {
    Graph<indexType> GH(8 * 3, Points.size());
    hcnng_index<Point, PR, indexType> I;
    I.seed = 3;
    I.build_index(GH, Points, 8, 200, 3);
    GH.save((g_out + "/hcnng.graph").c_str());
  }
  std::printf("mirrors alive: %zu\n", MirrorCache::get().size());
  release_device_mirrors();
  std::printf("host_api_check done\n");
  return 0;
}
