// check_nn_recall.h -- host mirror of the part of algorithms/utils/check_nn_recall.h needed to REPORT
// QPS at recall the reference's way: checkRecall (:17-125: time only the batched search, tie-aware
// recall :83-109, QPS :110).  The beam/limit sweep and CSV bucketing (:170-268) are out of scope.
#pragma once
#include <chrono>
#include <set>

#include "beam_search.h"

namespace parlayANN {

struct nn_result { double recall; double QPS; long k; long beamQ; double cut; size_t num_queries; long limit;
                   unsigned avg_cmps, tail_cmps, avg_visited, tail_visited; };

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
  return nn_result{recall, QPS, k, QP.beamSize, QP.cut, Query_Points.size(), QP.limit, ds[0], ds[1], vs[0], vs[1]};
}

}  // namespace parlayANN
