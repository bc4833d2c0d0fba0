// beam_search.h -- host mirror of the batched / single-query search surface of
// algorithms/utils/beamSearch.h: beam_search (:217-241), range_search (:245-306), searchAll (:353-387),
// qsearchAll (:537-565).
// Each call is ONE pann_batch_search over the whole batch (the parallel_for seam :374/:556).
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "device_index.h"
#include "stats.h"
#include "types.h"

namespace parlayANN {

inline pann_query_params to_pann(const QueryParams& QP) {
  pann_query_params q;
  q.k = QP.k; q.beam = QP.beamSize; q.cut = QP.cut; q.limit = QP.limit; q.degree_limit = QP.degree_limit;
  q.rerank_factor = QP.rerank_factor; q.pad = QP.pad;
  return q;
}

// beam_search(p, G, Points, starting_points, QP) -> ((frontier, visited), dist_cmps)   (:217-223)
template <class PointRange, typename indexType>
std::pair<std::pair<std::vector<std::pair<indexType, float>>, std::vector<std::pair<indexType, float>>>, size_t>
beam_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI,
            const std::vector<indexType>& starting_points, const QueryParams& QP) {
  using id_dist = std::pair<indexType, float>;
  const uint32_t beam = (uint32_t)QP.beamSize;
  uint32_t vcap = std::max<uint32_t>(4 * beam, 256);
  for (;;) {
    std::vector<uint32_t> ids(beam), vids(vcap);
    std::vector<float> ds(beam), vds(vcap);
    uint32_t fs = 0, vc = 0, dc = 0;
    pann_search_out out{};
    out.ids = ids.data(); out.dists = ds.data(); out.out_k = beam; out.frontier_size = &fs; out.visited_count = &vc;
    out.dist_cmps = &dc; out.visited_ids = vids.data(); out.visited_dists = vds.data(); out.visited_cap = vcap;
    const pann_query_params q = to_pann(QP);
    const int rc = pann_batch_search(DI.h, p.values, nullptr, 1, (uint64_t)p.params.num_bytes(), starting_points.data(),
                                     (uint32_t)starting_points.size(), &q, &out);
    if (rc == PANN_ERR_OVERFLOW) { vcap *= 4; continue; }
    pann_check(rc);
    std::vector<id_dist> frontier(fs), visited(vc);
    for (uint32_t i = 0; i < fs; i++) frontier[i] = id_dist(ids[i], ds[i]);
    for (uint32_t i = 0; i < vc; i++) visited[i] = id_dist(vids[i], vds[i]);
    // the reference keeps `visited` sorted by (dist,id) (:112-113); the device emits visit order
    std::sort(visited.begin(), visited.end(), [](const id_dist& a, const id_dist& b) {
      return a.second < b.second || (a.second == b.second && a.first < b.first); });
    return std::make_pair(std::make_pair(frontier, visited), (size_t)dc);
  }
}

template <class PointRange, typename indexType>
auto beam_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI, const indexType starting_point,
                 const QueryParams& QP) {                                                         // :234-241
  std::vector<indexType> s = {starting_point};
  return beam_search<PointRange, indexType>(p, DI, s, QP);
}

// searchAll (:362-387): first k ids of every query's frontier + stats
template <class PointRange, typename indexType>
std::vector<std::vector<indexType>> searchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                              stats<indexType>& QueryStats, const std::vector<indexType>& starting_points,
                                              QueryParams& QP, std::vector<float>* dists_out = nullptr) {
  if (QP.k > QP.beamSize) {
    std::cout << "Error: beam search parameter Q = " << QP.beamSize << " same size or smaller than k = " << QP.k << std::endl;
    abort();
  }
  const size_t nq = Query_Points.size();
  const uint32_t k = (uint32_t)QP.k;
  std::vector<uint32_t> ids(nq * k), vc(nq), dc(nq);
  std::vector<float> ds(nq * k);
  pann_search_out out{};
  out.ids = ids.data(); out.dists = ds.data(); out.out_k = k; out.visited_count = vc.data(); out.dist_cmps = dc.data();
  const pann_query_params q = to_pann(QP);
  pann_check(pann_batch_search(DI.h, Query_Points.data(), nullptr, nq, Query_Points.get_aligned_bytes(), starting_points.data(),
                               (uint32_t)starting_points.size(), &q, &out));
  std::vector<std::vector<indexType>> all(nq);
  for (size_t i = 0; i < nq; i++) {
    all[i].assign(ids.begin() + i * k, ids.begin() + (i + 1) * k);
    QueryStats.increment_visited((indexType)i, vc[i]);
    QueryStats.increment_dist((indexType)i, dc[i]);
  }
  if (dists_out) *dists_out = ds;
  return all;
}

template <class PointRange, typename indexType>
std::vector<std::vector<indexType>> searchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                              stats<indexType>& QueryStats, indexType starting_point, QueryParams& QP) {
  std::vector<indexType> s = {starting_point};                                                    // :353-360
  return searchAll<PointRange, indexType>(Query_Points, DI, QueryStats, s, QP);
}

// qsearchAll (:537-565) for the un-quantised case (Q_ and QQ_ ranges equal the base ranges, so
// beam_search_rerank degenerates to beam search + exact distances of the first k, :445-452)
template <class PointRange, typename indexType>
std::vector<std::vector<indexType>> qsearchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                               stats<indexType>& QueryStats, const indexType starting_point,
                                               const QueryParams& QP) {
  QueryParams q = QP;
  return searchAll<PointRange, indexType>(Query_Points, DI, QueryStats, starting_point, q);
}

// qsearchAll<PR, QPR, QQPR> (:537-565) = beam_search_rerank (:390-454) for every query, as TWO launches: the beam
// search runs on the quantised mirror (Q_Query_Points against QDI), then the first min(k * rerank_factor, |beam|)
// frontier ids are re-scored with the full-precision query against the full-precision mirror, sorted by (dist,id),
// and the first k kept (:426-444).  When both ranges have the same num_bytes() nothing is re-sorted: the first k
// frontier ids with their exact distances (:445-452).  The second-level filter (QQ ranges, use_filtering) is
// out of scope; the QQ arguments of the reference equal the Q ones in every configuration mirrored here.
template <class PointRange, class QPointRange, typename indexType>
std::vector<std::vector<indexType>> qsearchAll(PointRange& Query_Points, QPointRange& Q_Query_Points,
                                               DeviceIndex<PointRange, indexType>& DI, DeviceIndex<QPointRange, indexType>& QDI,
                                               stats<indexType>& QueryStats, const indexType starting_point,
                                               const QueryParams& QP, std::vector<float>* dists_out = nullptr) {
  if (QP.k > QP.beamSize) {
    std::cout << "Error: beam search parameter Q = " << QP.beamSize << " same size or smaller than k = " << QP.k << std::endl;
    abort();
  }
  const size_t nq = Query_Points.size();
  const uint32_t k = (uint32_t)QP.k, beam = (uint32_t)QP.beamSize;
  const bool use_rerank = Query_Points.params.num_bytes() != Q_Query_Points.params.num_bytes();     // :409
  std::vector<uint32_t> ids(nq * beam), fs(nq), vc(nq), dc(nq);
  pann_search_out out{};
  out.ids = ids.data(); out.out_k = beam; out.frontier_size = fs.data(); out.visited_count = vc.data(); out.dist_cmps = dc.data();
  const pann_query_params q = to_pann(QP);
  const uint32_t start = starting_point;
  pann_check(pann_batch_search(QDI.h, Q_Query_Points.data(), nullptr, nq, Q_Query_Points.get_aligned_bytes(), &start, 1, &q, &out));
  std::vector<uint32_t> counts(nq);
  for (size_t i = 0; i < nq; i++) {
    if (fs[i] < k) {                                                                                  // :416-419
      std::cout << "Error: for point id " << i << " beam search returned " << fs[i] << " elements, which is less than k = " << k << std::endl;
      abort();
    }
    counts[i] = use_rerank ? (uint32_t)std::min<long>((long)QP.k * QP.rerank_factor, (long)fs[i]) : k;
    QueryStats.increment_visited((indexType)i, vc[i]);
    QueryStats.increment_dist((indexType)i, dc[i]);
  }
  std::vector<uint32_t> rid(nq * k);
  std::vector<float> rd(nq * k);
  pann_check(pann_rerank(DI.h, Query_Points.data(), nq, Query_Points.get_aligned_bytes(), ids.data(), beam, counts.data(), k,
                         use_rerank ? 1 : 0, rid.data(), rd.data()));
  std::vector<std::vector<indexType>> all(nq);
  for (size_t i = 0; i < nq; i++) all[i].assign(rid.begin() + i * k, rid.begin() + (i + 1) * k);
  if (dists_out) *dists_out = rd;
  return all;
}

// range_search(p, G, Points, starting_points, radius, radius_2, QP) -> (result in BFS order, distance comparisons)
// (:245-306).  `self` = the query's own vertex when p is a base point (Point::same_as is pointer equality), else -1.
// One pann_range_search; the result row grows until it is not truncated.
template <class PointRange, typename indexType>
std::pair<std::vector<indexType>, long> range_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI,
                                                     const std::vector<indexType>& starting_points, float /*radius: unused, :250*/,
                                                     float radius_2, const QueryParams& /*QP*/, long self = -1) {
  if (starting_points.empty()) return {std::vector<indexType>(), 0L};
  for (uint32_t cap = 1024;; cap *= 8) {
    std::vector<uint32_t> ids(cap);
    uint32_t cnt = 0, cmps = 0, trunc = 0;
    const uint32_t qid = (uint32_t)self;
    pann_check(pann_range_search(DI.h, self < 0 ? p.values : nullptr, self < 0 ? nullptr : &qid, 1, (uint64_t)p.params.num_bytes(),
                                 starting_points.data(), (uint32_t)starting_points.size(), 0, radius_2, cap, ids.data(), &cnt,
                                 &cmps, &trunc));
    if (trunc && cap < (1u << 30)) continue;
    ids.resize(cnt);
    return {std::vector<indexType>(ids.begin(), ids.end()), (long)cmps};
  }
}

// the loop of vamana/neighbors.h:95-101 as ONE launch: base point i searches from starts[i] (a row of nstarts ids,
// 0xFFFFFFFF = padding); returns (counts, distance comparisons) per point
template <class PointRange, typename indexType>
std::pair<std::vector<long>, std::vector<long>> self_range_search(DeviceIndex<PointRange, indexType>& DI, size_t n,
                                                                  const std::vector<uint32_t>& starts, uint32_t nstarts,
                                                                  float radius_2, uint32_t max_results = 1024) {
  std::vector<uint32_t> qid(n), ids(n * (size_t)max_results), cnt(n), cmps(n), trunc(n);
  for (size_t i = 0; i < n; i++) qid[i] = (uint32_t)i;
  pann_check(pann_range_search(DI.h, nullptr, qid.data(), n, 0, starts.data(), nstarts, 1, radius_2, max_results, ids.data(),
                               cnt.data(), cmps.data(), trunc.data()));
  return {std::vector<long>(cnt.begin(), cnt.end()), std::vector<long>(cmps.begin(), cmps.end())};
}

}  // namespace parlayANN
