// check_nn_recall.h -- host mirror of algorithms/utils/check_nn_recall.h with the reference's argument lists:
//   checkRecall(G, Base_Points, Query_Points, Q_Base_Points, Q_Query_Points, QQ_Base_Points, QQ_Query_Points, GT,
//               random, start_point, k, QP, verbose)                                                  :17-125
//   write_to_csv(csv_filename, buckets, results, G)                                                   :127-158
//   search_and_parse(G_, G, Base_Points, Query_Points, GT, res_file, k, verbose, fixed_beam_width)     :170-179
//   search_and_parse(G_, G, Base_.., Query_.., Q_Base_.., Q_Query_.., QQ_Base_.., QQ_Query_.., GT, res_file, k,
//                    random, start_point, verbose, fixed_beam_width, rerank_factor)                    :181-268
// Only the batched search is timed (:47-56); recall is tie-aware with the tie distances RECOMPUTED from the points
// (:91-96), not read from the ground-truth file; QPS = queries / search time (:110).
#pragma once
#include <chrono>
#include <fstream>
#include <set>
#include <string>

#include "beam_search.h"
#include "parse_results.h"

namespace parlayANN {

namespace detail {

// exact distances from query i to the ground-truth ids GT(i, from .. dim): ONE pann_rerank launch (resort == 0 keeps
// the given order, beamSearch.h:447-452), the device form of qp.distance(Base_Points[GT.coordinates(i, l)]) (:91,95)
template <class PointRange, typename indexType>
std::vector<float> gt_distances(pann_index* h, const PointRange& Query_Points, const groundTruth<indexType>& GT) {
  const size_t nq = Query_Points.size(), dim = (size_t)GT.dimension();
  std::vector<uint32_t> cand(nq * dim), oi(nq * dim);
  std::vector<float> od(nq * dim);
  for (size_t i = 0; i < nq; i++) for (size_t l = 0; l < dim; l++) cand[i * dim + l] = GT.coordinates((long)i, (long)l);
  pann_check(pann_rerank(h, Query_Points.data(), nq, Query_Points.get_aligned_bytes(), cand.data(), (uint32_t)dim, nullptr, (uint32_t)dim, 0,
                         oi.data(), od.data()));
  return od;
}

// score + report (:57-125)
template <class PointRange, typename indexType>
nn_result report_recall(const parlay::sequence<parlay::sequence<indexType>>& all_ngh, stats<indexType>& QueryStats,
                        const PointRange& Query_Points, const groundTruth<indexType>& GT, const std::vector<float>& gt_dist, long k,
                        const QueryParams& QP, double query_time, bool verbose) {
  float recall = 0.0;
  if (GT.size() > 0) {
    const size_t n = Query_Points.size(), dim = (size_t)GT.dimension();
    long numCorrect = 0;
    for (size_t i = 0; i < n; i++) {
      if ((long)all_ngh[i].size() < k) { std::cout << "bad number of neighbors reported: " << all_ngh[i].size() << std::endl; abort(); }   // :66-69
      std::vector<indexType> results_with_ties;
      for (long l = 0; l < k; l++) results_with_ties.push_back(GT.coordinates((long)i, l));
      const float last_dist = gt_dist[i * dim + (k - 1)];
      for (size_t l = (size_t)k; l < dim; l++) if (gt_dist[i * dim + l] == last_dist) results_with_ties.push_back(GT.coordinates((long)i, (long)l));
      std::set<indexType> reported_nbhs(all_ngh[i].begin(), all_ngh[i].begin() + k);
      for (auto a : results_with_ties) if (reported_nbhs.count(a)) numCorrect++;
    }
    recall = static_cast<float>(numCorrect) / static_cast<float>(k * n);
  }
  const float QPS = Query_Points.size() / query_time;
  auto ds = QueryStats.dist_stats(); auto vs = QueryStats.visited_stats();
  if (verbose)
    std::cout << "search: Q=" << QP.beamSize << ", k=" << QP.k << ", limit=" << QP.limit << ", recall=" << recall
              << ", visited=" << vs[0] << ", comparisons=" << ds[0] << ", QPS=" << QPS
              << ", ctime=" << 1 / (QPS * ds[0]) * 1e9 << std::endl;
  parlay::sequence<unsigned> st = {ds[0], ds[1], vs[0], vs[1]};
  return nn_result(recall, st, QPS, (int)k, (int)QP.beamSize, (float)QP.cut, (long)Query_Points.size(), (int)QP.limit, (int)QP.degree_limit, (int)k);
}

inline void check_gt_size(long k, long gt_dim, size_t gt_n) {                                     // :32-36
  if (gt_n > 0 && k > gt_dim) { std::cout << k << "@" << k << " too large for ground truth data of size " << gt_dim << std::endl; abort(); }
}

}  // namespace detail

template <typename PointRange, typename QPointRange, typename QQPointRange, typename indexType>
nn_result checkRecall(const Graph<indexType>& G, const PointRange& Base_Points, const PointRange& Query_Points,
                      const QPointRange& Q_Base_Points, const QPointRange& Q_Query_Points, const QQPointRange& QQ_Base_Points,
                      const QQPointRange& QQ_Query_Points, const groundTruth<indexType>& GT, const bool random, const long start_point,
                      const long k, const QueryParams& QP, const bool verbose) {
  detail::check_gt_size(k, GT.dimension(), GT.size());
  parlay::sequence<parlay::sequence<indexType>> all_ngh;
  stats<indexType> QueryStats(Query_Points.size());
  QueryStats.clear();
  { auto warm = device_mirror(G, Base_Points); (void)warm; }          // uploads happen outside the timed region, like the file loads upstream
  if (!random && (const void*)Base_Points.data() != (const void*)Q_Base_Points.data()) { auto warm = device_mirror(G, Q_Base_Points); (void)warm; }
  const auto t0 = std::chrono::steady_clock::now();
  if (random) all_ngh = beamSearchRandom(Query_Points, G, Base_Points, QueryStats, QP);
  else all_ngh = qsearchAll<PointRange, QPointRange, QQPointRange, indexType>(Query_Points, Q_Query_Points, QQ_Query_Points, G, Base_Points,
                                                                              Q_Base_Points, QQ_Base_Points, QueryStats, (indexType)start_point, QP);
  const double query_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<float> gd;
  if (GT.size() > 0) { auto L = device_mirror(G, Base_Points); gd = detail::gt_distances<PointRange, indexType>(L.h(), Query_Points, GT); }
  return detail::report_recall<PointRange, indexType>(all_ngh, QueryStats, Query_Points, GT, gd, k, QP, query_time, verbose);
}

// write_to_csv (:127-158; format of utils/csvfile.h: comma separated, strings quoted, the file is appended to):
// graph block, blank row, one row per recall bucket, two blank rows
inline void write_to_csv(std::string csv_filename, parlay::sequence<float> buckets, parlay::sequence<nn_result> results, Graph_ G) {
  std::ofstream f(csv_filename, std::ios::app);
  if (!f.is_open()) { std::cout << "ERROR: cannot open " << csv_filename << std::endl; abort(); }
  auto q = [](const std::string& v) {                      // strings are quoted, embedded quotes doubled
    std::string o = "\"";
    for (char c : v) { if (c == '"') o += '"'; o += c; }
    return o + "\"";
  };
  f << q("GRAPH") << ',' << q("Parameters") << ',' << q("Size") << ',' << q("Build time") << ',' << q("Avg degree") << ','
    << q("Max degree") << '\n';
  f << q(G.name) << ',' << q(G.params) << ',' << G.size << ',' << G.time << ',' << G.avg_deg << ',' << G.max_deg << '\n' << '\n';
  const char* cols[] = {"Num queries", "Target recall", "Actual recall", "QPS", "Average Cmps", "Tail Cmps", "Average Visited",
                        "Tail Visited", "k", "Q", "cut"};
  for (size_t i = 0; i < 11; i++) f << (i ? "," : "") << q(cols[i]);
  f << '\n';
  for (size_t i = 0; i < results.size(); i++) {
    const nn_result& N = results[i];
    f << N.num_queries << ',' << buckets[i] << ',' << N.recall << ',' << N.QPS << ',' << N.avg_cmps << ',' << N.tail_cmps << ','
      << N.avg_visited << ',' << N.tail_visited << ',' << N.k << ',' << N.beamQ << ',' << N.cut << '\n';
  }
  f << '\n' << '\n';
}

template <typename PointRange, typename QPointRange, typename QQPointRange, typename indexType>
void search_and_parse(Graph_ G_, Graph<indexType>& G, PointRange& Base_Points, PointRange& Query_Points, QPointRange& Q_Base_Points,
                      QPointRange& Q_Query_Points, QQPointRange& QQ_Base_Points, QQPointRange& QQ_Query_Points, groundTruth<indexType> GT,
                      char* res_file, long k, bool random = true, indexType start_point = 0, bool verbose = false,
                      long fixed_beam_width = 0, int rerank_factor = 100) {
  parlay::sequence<nn_result> results;
  auto check = [&](const long k, const QueryParams QP) {
    return checkRecall(G, Base_Points, Query_Points, Q_Base_Points, Q_Query_Points, QQ_Base_Points, QQ_Query_Points, GT, random,
                       (long)start_point, k, QP, verbose);
  };
  QueryParams QP;
  QP.limit = (long)G.size();
  QP.rerank_factor = rerank_factor;
  QP.degree_limit = (long)G.max_degree();
  static const long beams[] = {10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 45, 50, 55, 60,
                               65, 70, 80, 90, 100, 120, 140, 160, 180, 200, 225, 250, 275, 300, 375, 500, 750, 1000};
  const long r = (k == 0) ? 10 : k;                                                          // :220-221
  const double cut = 1.35;
  if (fixed_beam_width != 0) {                                                               // -Q: five timed repetitions (:224-229)
    QP.k = r; QP.cut = cut; QP.beamSize = fixed_beam_width;
    for (int i = 0; i < 5; i++) check(QP.k, QP);
    return;
  }
  QP.k = r; QP.cut = cut;
  for (long Q : beams) if (Q >= r) { QP.beamSize = Q; results.push_back(check(r, QP)); }    // :231-241
  static const long limits[] = {10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 30, 35};
  QP = QueryParams(r, r, 1.35, (long)G.size(), (long)G.max_degree());                        // "limited accuracy" :245-253
  for (long l : limits) {
    QP.limit = l;
    QP.beamSize = std::max<long>(l, r);
    QP.degree_limit = std::min<int>((int)G.max_degree(), (int)(5 * l));
    results.push_back(check(r, QP));
  }
  QP = QueryParams((long)100, (long)1000, (double)10.0, (long)G.size(), (long)G.max_degree());   // "best accuracy" :255-256
  results.push_back(check(r, QP));
  parlay::sequence<float> buckets = {.1f, .2f, .3f, .4f, .5f, .6f, .7f, .75f, .8f, .85f, .9f, .93f, .95f, .97f, .98f, .99f, .995f,
                                     .999f, .9995f, .9999f, .99995f, .99999f};
  auto [res, ret_buckets] = parse_result(results, buckets);
  std::cout << std::endl;
  if (res_file != NULL) write_to_csv(std::string(res_file), ret_buckets, res, G_);           // :265-266
}

template <typename PointRange, typename indexType>
void search_and_parse(Graph_ G_, Graph<indexType>& G, PointRange& Base_Points, PointRange& Query_Points, groundTruth<indexType> GT,
                      char* res_file, long k, bool verbose = false, long fixed_beam_width = 0) {
  search_and_parse(G_, G, Base_Points, Query_Points, Base_Points, Query_Points, Base_Points, Query_Points, GT, res_file, k, false, 0u,
                   verbose, fixed_beam_width);
}

}  // namespace parlayANN
