// check_nn_recall.h -- host mirror of the part of algorithms/utils/check_nn_recall.h needed to REPORT
// QPS at recall the reference's way: checkRecall (:17-125: time only the batched search, tie-aware
// recall :83-109, QPS :110), the sweep of search_and_parse (:181-268: 43 beam widths, 20 visit limits, one
// "best accuracy" point) and the best-QPS-per-recall-bucket table of parse_result (parse_results.h:192-218).
// write_to_csv (:127-158, format of utils/csvfile.h: comma separated, strings quoted, appended to the file).
#pragma once
#include <chrono>
#include <fstream>
#include <string>
#include <set>

#include "beam_search.h"

namespace parlayANN {

struct nn_result {
  double recall; double QPS; long k; long beamQ; double cut; size_t num_queries; long limit;
  unsigned avg_cmps, tail_cmps, avg_visited, tail_visited;
  long degree_limit = 0;
  void print() const {                                                      // parse_results.h:139-144
    std::cout << "For " << k << "@" << k << " recall = " << recall << ", QPS = " << QPS << ", Q = " << beamQ << ", cut = " << cut
              << ", visited limit = " << limit << ", degree limit: " << degree_limit << ", average visited = " << avg_visited
              << ", average cmps = " << avg_cmps << std::endl;
  }
};

// score + report (check_nn_recall.h:83-125) shared by the plain and the quantised-with-rerank searches
template <class PointRange, typename indexType>
nn_result report_recall(const std::vector<std::vector<indexType>>& all_ngh, stats<indexType>& QueryStats, PointRange& Query_Points,
                        const groundTruth<indexType>& GT, long k, const QueryParams& QP, double query_time, bool verbose);

template <class PointRange, typename indexType>
nn_result checkRecall(DeviceIndex<PointRange, indexType>& DI, PointRange& Query_Points, const groundTruth<indexType>& GT,
                      long start_point, long k, const QueryParams& QP, bool verbose) {
  if (GT.size() > 0 && k > GT.dimension()) {
    std::cout << k << "@" << k << " too large for ground truth data of size " << GT.dimension() << std::endl;
    abort();
  }
  stats<indexType> QueryStats(Query_Points.size());
  const auto t0 = std::chrono::steady_clock::now();
  auto all_ngh = qsearchAll<PointRange, indexType>(Query_Points, DI, QueryStats, (indexType)start_point, QP);
  const double query_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return report_recall<PointRange, indexType>(all_ngh, QueryStats, Query_Points, GT, k, QP, query_time, verbose);
}

// checkRecall with a quantised first pass (check_nn_recall.h:51-54: qsearchAll over the three range pairs)
template <class PointRange, class QPointRange, typename indexType>
nn_result checkRecall(DeviceIndex<PointRange, indexType>& DI, DeviceIndex<QPointRange, indexType>& QDI, PointRange& Query_Points,
                      QPointRange& Q_Query_Points, const groundTruth<indexType>& GT, long start_point, long k,
                      const QueryParams& QP, bool verbose) {
  if (GT.size() > 0 && k > GT.dimension()) {
    std::cout << k << "@" << k << " too large for ground truth data of size " << GT.dimension() << std::endl;
    abort();
  }
  stats<indexType> QueryStats(Query_Points.size());
  const auto t0 = std::chrono::steady_clock::now();
  auto all_ngh = qsearchAll<PointRange, QPointRange, indexType>(Query_Points, Q_Query_Points, DI, QDI, QueryStats,
                                                                (indexType)start_point, QP);
  const double query_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return report_recall<PointRange, indexType>(all_ngh, QueryStats, Query_Points, GT, k, QP, query_time, verbose);
}

template <class PointRange, typename indexType>
nn_result report_recall(const std::vector<std::vector<indexType>>& all_ngh, stats<indexType>& QueryStats, PointRange& Query_Points,
                        const groundTruth<indexType>& GT, long k, const QueryParams& QP, double query_time, bool verbose) {
  double recall = 0.0;
  if (GT.size() > 0) {
    const size_t n = Query_Points.size();
    long numCorrect = 0;
    for (size_t i = 0; i < n; i++) {
      std::vector<indexType> accepted;
      for (long l = 0; l < k; l++) accepted.push_back(GT.coordinates(i, l));
      const float last_dist = GT.distances(i, k - 1);
      for (long l = k; l < GT.dimension(); l++) if (GT.distances(i, l) == last_dist) accepted.push_back(GT.coordinates(i, l));
      std::set<indexType> reported(all_ngh[i].begin(), all_ngh[i].begin() + k);
      for (auto a : accepted) if (reported.count(a)) numCorrect++;
    }
    recall = (double)numCorrect / (double)(k * n);
  }
  const double QPS = Query_Points.size() / query_time;
  auto ds = QueryStats.dist_stats(); auto vs = QueryStats.visited_stats();
  if (verbose)
    std::cout << "search: Q=" << QP.beamSize << ", k=" << QP.k << ", limit=" << QP.limit << ", recall=" << recall
              << ", visited=" << vs[0] << ", comparisons=" << ds[0] << ", QPS=" << QPS
              << ", ctime=" << 1 / (QPS * ds[0]) * 1e9 << std::endl;
  return nn_result{recall, QPS, k, QP.beamSize, QP.cut, Query_Points.size(), QP.limit, ds[0], ds[1], vs[0], vs[1], QP.degree_limit};
}

// parse_result (parse_results.h:192-218): for bucket b_i the fastest result with b_i <= recall <= b_{i+1}
// (the last bucket is open above); prints one line per non-empty bucket
inline std::pair<std::vector<nn_result>, std::vector<float>> parse_result(const std::vector<nn_result>& results,
                                                                          const std::vector<float>& buckets) {
  std::vector<nn_result> best; std::vector<float> kept;
  for (size_t i = 0; i < buckets.size(); i++) {
    const nn_result* top = nullptr;
    bool any_above = false;
    for (const nn_result& r : results) any_above = any_above || r.recall >= buckets[i];
    for (const nn_result& r : results) {
      if (r.recall < buckets[i]) continue;
      if (i + 1 < buckets.size() && any_above && r.recall > buckets[i + 1]) continue;
      if (!top || top->QPS < r.QPS) top = &r;
    }
    if (top) { top->print(); best.push_back(*top); kept.push_back(buckets[i]); }
  }
  return {best, kept};
}

// what the report says about the graph (parse_results.h:35-58)
struct Graph_ {
  std::string name, params; long size = 0; double avg_deg = 0; int max_deg = 0; double time = 0;
  void print() const {
    std::cout << name << " graph built with " << size << " points and parameters " << params << std::endl;
    std::cout << "Graph has average degree " << avg_deg << " and maximum degree " << max_deg << std::endl;
    std::cout << "Graph built in " << time << " seconds" << std::endl;
  }
};

// write_to_csv (:127-158): graph block, blank row, one row per recall bucket, two blank rows; the file is appended to
inline void write_to_csv(const std::string& csv_filename, const std::vector<float>& buckets, const std::vector<nn_result>& results,
                         const Graph_& G) {
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
    f << N.num_queries << ',' << buckets[i] << ',' << N.recall << ',' << (float)N.QPS << ',' << N.avg_cmps << ',' << N.tail_cmps << ','
      << N.avg_visited << ',' << N.tail_visited << ',' << N.k << ',' << N.beamQ << ',' << (float)N.cut << '\n';
  }
  f << '\n' << '\n';
}

// search_and_parse (check_nn_recall.h:181-268).  `check(QP)` runs one checkRecall (plain or quantised + rerank).
template <class Check>
std::vector<nn_result> search_and_parse(Check&& check, size_t n, long max_degree, long k, long fixed_beam_width, int rerank_factor = 100,
                                        const char* res_file = nullptr, const Graph_* G_ = nullptr) {
  std::vector<nn_result> results;
  const long r = k == 0 ? 10 : k;                                                          // :220-221
  QueryParams QP(r, r, 1.35, (long)n, max_degree);
  QP.rerank_factor = rerank_factor;
  if (fixed_beam_width != 0) {                                                             // -Q: five timed repetitions (:224-229)
    QP.beamSize = fixed_beam_width;
    for (int i = 0; i < 5; i++) results.push_back(check(QP));
    return results;
  }
  static const long beams[] = {10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 45, 50, 55, 60,
                               65, 70, 80, 90, 100, 120, 140, 160, 180, 200, 225, 250, 275, 300, 375, 500, 750, 1000};
  static const long limits[] = {10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 30, 35};
  for (long Q : beams) if (Q >= r) { QP.beamSize = Q; results.push_back(check(QP)); }      // :231-241
  for (long l : limits) {                                                                  // "limited accuracy" :245-253
    QueryParams L(r, std::max<long>(l, r), 1.35, l, std::min<long>(max_degree, 5 * l));
    L.rerank_factor = rerank_factor;
    results.push_back(check(L));
  }
  {                                                                                        // "best accuracy" :255-256
    QueryParams B(100, 1000, 10.0, (long)n, max_degree);
    results.push_back(check(B));
  }
  const std::vector<float> buckets = {.1f, .2f, .3f, .4f, .5f, .6f, .7f, .75f, .8f, .85f, .9f, .93f, .95f, .97f, .98f, .99f, .995f,
                                      .999f, .9995f, .9999f, .99995f, .99999f};
  auto [best, kept] = parse_result(results, buckets);
  std::cout << std::endl;
  if (res_file != nullptr && G_ != nullptr) write_to_csv(std::string(res_file), kept, best, *G_);       // :265-266
  return results;
}

}  // namespace parlayANN
