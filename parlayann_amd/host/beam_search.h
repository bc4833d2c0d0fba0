// beam_search.h -- host mirror of the search surface of algorithms/utils/beamSearch.h with the reference's own
// names and argument lists:
//   filtered_beam_search (:22-33, use_filtering == false)      beam_search (:217-223, :234-241)   beam_search_impl (:226-231)
//   range_search (:245-306)      beamSearchRandom (:309-351)   searchAll (:353-387)
//   beam_search_rerank (:390-454)   beam_search_rerank__ (:499-521)   qsearchAll (:537-565)
// The CPU loop bodies are gone: every function is one (or, with a rerank, two) C-ABI call on the device mirror of
// (G, Points) -- the batched ones replace the parallel_for over queries (:374, :556) by ONE launch for the batch.
// Overloads taking a DeviceIndex (an explicitly managed mirror) end in the same calls.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "device_index.h"
#include "parlay_compat.h"
#include "stats.h"
#include "types.h"

namespace parlayANN {

inline pann_query_params to_pann(const QueryParams& QP) {
  pann_query_params q;
  q.k = QP.k; q.beam = QP.beamSize; q.cut = QP.cut; q.limit = QP.limit; q.degree_limit = QP.degree_limit;
  q.rerank_factor = QP.rerank_factor; q.pad = QP.pad;
  return q;
}

template <typename indexType>
using id_dist_seq = parlay::sequence<std::pair<indexType, float>>;
template <typename indexType>
using beam_result = std::pair<std::pair<id_dist_seq<indexType>, id_dist_seq<indexType>>, size_t>;

namespace detail {

inline void check_k_le_beam(const QueryParams& QP) {                       // beamSearch.h:316-320, :368-372, :549-553
  if (QP.k > QP.beamSize) {
    std::cout << "Error: beam search parameter Q = " << QP.beamSize << " same size or smaller than k = " << QP.k << std::endl;
    abort();
  }
}

// one query on handle h: ((frontier, visited sorted by (dist,id)), dist_cmps).  self >= 0: the query is base point `self`
template <typename indexType>
beam_result<indexType> search_one(pann_index* h, const void* qvalues, uint64_t qbytes, long self, const indexType* starts,
                                  size_t nstarts, const QueryParams& QP) {
  using id_dist = std::pair<indexType, float>;
  if (nstarts == 0) { std::cout << "beam search expects at least one start point" << std::endl; abort(); }   // :38-41
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
    const uint32_t qid = (uint32_t)self;
    std::vector<uint32_t> st(starts, starts + nstarts);
    const int rc = pann_batch_search(h, self < 0 ? qvalues : nullptr, self < 0 ? nullptr : &qid, 1, qbytes, st.data(),
                                     (uint32_t)nstarts, &q, &out);
    if (rc == PANN_ERR_OVERFLOW && vc > vcap) { vcap = vc; continue; }
    if (rc == PANN_ERR_OVERFLOW) { vcap *= 4; continue; }
    pann_check(rc);
    id_dist_seq<indexType> frontier(fs), visited(vc);
    for (uint32_t i = 0; i < fs; i++) frontier[i] = id_dist(ids[i], ds[i]);
    for (uint32_t i = 0; i < vc; i++) visited[i] = id_dist(vids[i], vds[i]);
    // the reference keeps `visited` sorted by (dist,id) (:112-113); the device emits visit order
    std::sort(visited.begin(), visited.end(), [](const id_dist& a, const id_dist& b) {
      return a.second < b.second || (a.second == b.second && a.first < b.first); });
    return std::make_pair(std::make_pair(frontier, visited), (size_t)dc);
  }
}

// the batch: first k ids (and distances) of every query's frontier, counters into QueryStats.  starts: nstarts shared ids,
// or nq x nstarts when per_query
template <class PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> search_batch(pann_index* h, const PointRange& Query_Points, stats<indexType>& QueryStats,
                                                           const indexType* starts, size_t nstarts, bool per_query,
                                                           const QueryParams& QP, std::vector<float>* dists_out = nullptr) {
  check_k_le_beam(QP);
  const size_t nq = Query_Points.size();
  const uint32_t k = (uint32_t)QP.k;
  std::vector<uint32_t> ids(nq * k), fs(nq), vc(nq), dc(nq);
  std::vector<float> ds(nq * k);
  pann_search_out out{};
  out.ids = ids.data(); out.dists = ds.data(); out.out_k = k; out.frontier_size = fs.data(); out.visited_count = vc.data();
  out.dist_cmps = dc.data();
  const pann_query_params q = to_pann(QP);
  std::vector<uint32_t> st(starts, starts + (per_query ? nq * nstarts : nstarts));
  if (per_query) pann_check(pann_batch_search_per_query_starts(h, Query_Points.data(), nullptr, nq, Query_Points.get_aligned_bytes(), st.data(), (uint32_t)nstarts, &q, &out));
  else pann_check(pann_batch_search(h, Query_Points.data(), nullptr, nq, Query_Points.get_aligned_bytes(), st.data(), (uint32_t)nstarts, &q, &out));
  parlay::sequence<parlay::sequence<indexType>> all(nq);
  for (size_t i = 0; i < nq; i++) {
    all[i] = parlay::sequence<indexType>(ids.begin() + i * k, ids.begin() + (i + 1) * k);
    QueryStats.increment_visited((indexType)i, vc[i]);
    QueryStats.increment_dist((indexType)i, dc[i]);
  }
  if (dists_out) *dists_out = ds;
  return all;
}

// beam_search_rerank (:390-454) for every query as TWO launches: beam search of the Q_ queries on the quantised mirror
// qh, then the first min(k * rerank_factor, |beam|) frontier ids re-scored by the full-precision queries on mirror h,
// sorted by (dist,id), first k kept (:426-444).  Equal num_bytes(): nothing is re-sorted, the first k frontier ids get
// their exact distances (:445-452).
template <class PointRange, class QPointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> search_rerank_batch(pann_index* h, pann_index* qh, const PointRange& Query_Points,
                                                                  const QPointRange& Q_Query_Points, stats<indexType>& QueryStats,
                                                                  indexType starting_point, const QueryParams& QP, bool count_stats,
                                                                  std::vector<float>* dists_out = nullptr) {
  check_k_le_beam(QP);
  const size_t nq = Query_Points.size();
  const uint32_t k = (uint32_t)QP.k, beam = (uint32_t)QP.beamSize;
  const bool use_rerank = Query_Points.params.num_bytes() != Q_Query_Points.params.num_bytes();     // :409
  std::vector<uint32_t> ids(nq * beam), fs(nq), vc(nq), dc(nq);
  pann_search_out out{};
  out.ids = ids.data(); out.out_k = beam; out.frontier_size = fs.data(); out.visited_count = vc.data(); out.dist_cmps = dc.data();
  const pann_query_params q = to_pann(QP);
  const uint32_t start = starting_point;
  pann_check(pann_batch_search(qh, Q_Query_Points.data(), nullptr, nq, Q_Query_Points.get_aligned_bytes(), &start, 1, &q, &out));
  std::vector<uint32_t> counts(nq);
  for (size_t i = 0; i < nq; i++) {
    if (fs[i] < k) {                                                                                  // :416-419
      std::cout << "Error: for point id " << i << " beam search returned " << fs[i] << " elements, which is less than k = " << k << std::endl;
      abort();
    }
    counts[i] = use_rerank ? (uint32_t)std::min<long>((long)QP.k * QP.rerank_factor, (long)fs[i]) : k;
    if (count_stats) {                                                                                // :421-424
      QueryStats.increment_visited((indexType)i, vc[i]);
      QueryStats.increment_dist((indexType)i, dc[i]);
    }
  }
  std::vector<uint32_t> rid(nq * k);
  std::vector<float> rd(nq * k);
  pann_check(pann_rerank(h, Query_Points.data(), nq, Query_Points.get_aligned_bytes(), ids.data(), beam, counts.data(), k,
                         use_rerank ? 1 : 0, rid.data(), rd.data()));
  parlay::sequence<parlay::sequence<indexType>> all(nq);
  for (size_t i = 0; i < nq; i++) all[i] = parlay::sequence<indexType>(rid.begin() + i * k, rid.begin() + (i + 1) * k);
  if (dists_out) *dists_out = rd;
  return all;
}

template <typename indexType>
std::pair<std::vector<indexType>, long> range_one(pann_index* h, const void* qvalues, uint64_t qbytes, long self,
                                                  const indexType* starts, size_t nstarts, float radius_2) {
  if (nstarts == 0) return {std::vector<indexType>(), 0L};
  std::vector<uint32_t> st(starts, starts + nstarts);
  for (uint32_t cap = 1024;; cap *= 8) {
    std::vector<uint32_t> ids(cap);
    uint32_t cnt = 0, cmps = 0, trunc = 0;
    const uint32_t qid = (uint32_t)self;
    pann_check(pann_range_search(h, self < 0 ? qvalues : nullptr, self < 0 ? nullptr : &qid, 1, qbytes, st.data(), (uint32_t)nstarts,
                                 0, radius_2, cap, ids.data(), &cnt, &cmps, &trunc));
    if (trunc && cap < (1u << 30)) continue;
    ids.resize(cnt);
    return {std::vector<indexType>(ids.begin(), ids.end()), (long)cmps};
  }
}

// the draws of beamSearchRandom (:326-332): one start per query, uniform over [0, n).  The reference keys a
// parlay::random_generator by the query number; that generator is parlaylib-internal, so this build draws from
// splitmix64(i) -- the same role, not the same numbers (DESIGN.md "unpinned vs upstream").
inline uint64_t random_start(uint64_t i, uint64_t n) {
  uint64_t z = (i + 1) * 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (z ^ (z >> 31)) % n;
}

}  // namespace detail

// ============================================================================================================
// the reference's argument lists
// ============================================================================================================

// filtered_beam_search(G, p, Points, qp, Q_Points, starting_points, QP, use_filtering)   (:22-33).  The second-level
// filter (use_filtering == true: :119-123,140-146) is outside this build's scope (only reachable
// with -quantize_mode >= 2 or d > 800); without it qp / Q_Points are not read (:150-159 use p / Points).
template <typename indexType, typename Point, typename PointRange, typename QPoint, typename QPointRange>
beam_result<indexType> filtered_beam_search(const Graph<indexType>& G, const Point p, const PointRange& Points, const QPoint /*qp*/,
                                            const QPointRange& /*Q_Points*/, const parlay::sequence<indexType> starting_points,
                                            const QueryParams& QP, bool use_filtering = false) {
  if (use_filtering) { std::cout << "Error: filtered_beam_search with use_filtering is not mirrored on the device" << std::endl; abort(); }
  auto L = device_mirror(G, Points);
  return detail::search_one<indexType>(L.h(), p.values, (uint64_t)p.params.num_bytes(), own_vertex(p, Points), starting_points.data(),
                                       starting_points.size(), QP);
}

// beam_search(p, G, Points, starting_points, QP) -> ((frontier, visited), dist_cmps)   (:217-223)
template <typename Point, typename PointRange, typename indexType>
beam_result<indexType> beam_search(const Point p, const Graph<indexType>& G, const PointRange& Points,
                                   const parlay::sequence<indexType> starting_points, const QueryParams& QP) {
  return filtered_beam_search(G, p, Points, p, Points, starting_points, QP, false);
}

// beam_search_impl(p, G, Points, starting_points, QP)   (:226-231; GT is the Graph here)
template <typename indexType, typename Point, typename PointRange>
beam_result<indexType> beam_search_impl(Point p, Graph<indexType>& G, PointRange& Points, parlay::sequence<indexType> starting_points,
                                        QueryParams& QP) {
  return filtered_beam_search(G, p, Points, p, Points, starting_points, QP, false);
}

// single start point (:234-241)
template <typename Point, typename PointRange, typename indexType>
beam_result<indexType> beam_search(const Point p, const Graph<indexType>& G, const PointRange& Points, const indexType starting_point,
                                   const QueryParams& QP) {
  parlay::sequence<indexType> start_points = {starting_point};
  return beam_search(p, G, Points, start_points, QP);
}

// range_search(p, G, Points, starting_points, radius, radius_2, QP, use_existing) -> (result in BFS order, distance comparisons)
// (:245-306; `radius` is unused upstream, :250; use_existing only switches a commented-out branch, :260-262)
template <typename indexType, typename Point, typename PointRange>
std::pair<std::vector<indexType>, long> range_search(Point p, Graph<indexType>& G, PointRange& Points,
                                                     parlay::sequence<indexType> starting_points, float /*radius*/, float radius_2,
                                                     QueryParams& /*QP*/, bool /*use_existing*/ = false) {
  auto L = device_mirror(G, Points);
  return detail::range_one<indexType>(L.h(), p.values, (uint64_t)p.params.num_bytes(), own_vertex(p, Points), starting_points.data(),
                                      starting_points.size(), radius_2);
}

// beamSearchRandom(Query_Points, G, Base_Points, QueryStats, QP)   (:309-351)
template <typename PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> beamSearchRandom(const PointRange& Query_Points, const Graph<indexType>& G,
                                                               const PointRange& Base_Points, stats<indexType>& QueryStats,
                                                               const QueryParams& QP) {
  detail::check_k_le_beam(QP);
  std::vector<indexType> starts(Query_Points.size());
  for (size_t i = 0; i < starts.size(); i++) starts[i] = (indexType)detail::random_start(i, G.size());
  auto L = device_mirror(G, Base_Points);
  return detail::search_batch<PointRange, indexType>(L.h(), Query_Points, QueryStats, starts.data(), 1, true, QP);
}

// searchAll(Query_Points, G, Base_Points, QueryStats, starting_points, QP)   (:362-387)
template <typename PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> searchAll(PointRange& Query_Points, Graph<indexType>& G, PointRange& Base_Points,
                                                        stats<indexType>& QueryStats, parlay::sequence<indexType> starting_points,
                                                        QueryParams& QP) {
  auto L = device_mirror(G, Base_Points);
  return detail::search_batch<PointRange, indexType>(L.h(), Query_Points, QueryStats, starting_points.data(), starting_points.size(),
                                                     false, QP);
}
template <typename PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> searchAll(PointRange& Query_Points, Graph<indexType>& G, PointRange& Base_Points,
                                                        stats<indexType>& QueryStats, indexType starting_point, QueryParams& QP) {
  parlay::sequence<indexType> start_points = {starting_point};                                   // :353-360
  return searchAll<PointRange, indexType>(Query_Points, G, Base_Points, QueryStats, start_points, QP);
}

// beam_search_rerank(p, qp, qqp, G, Base_Points, Q_Base_Points, QQ_Base_Points, QueryStats, starting_points, QP, stats)
// -> k (id, exact distance) pairs   (:390-454).  QQ ranges must have the Q ranges' num_bytes() (no second-level filter).
template <typename Point, typename QPoint, typename QQPoint, typename PointRange, typename QPointRange, typename QQPointRange,
          typename indexType>
id_dist_seq<indexType> beam_search_rerank(const Point& p, const QPoint& qp, const QQPoint& /*qqp*/, const Graph<indexType>& G,
                                          const PointRange& Base_Points, const QPointRange& Q_Base_Points,
                                          const QQPointRange& QQ_Base_Points, stats<indexType>& QueryStats,
                                          const parlay::sequence<indexType> starting_points, const QueryParams& QP, bool stats_ = true) {
  if (Q_Base_Points.params.num_bytes() != QQ_Base_Points.params.num_bytes()) {
    std::cout << "Error: beam_search_rerank with a second-level filter range is not mirrored on the device" << std::endl; abort();
  }
  const bool use_rerank = Base_Points.params.num_bytes() != Q_Base_Points.params.num_bytes();
  beam_result<indexType> r;
  {
    auto QL = device_mirror(G, Q_Base_Points);
    r = detail::search_one<indexType>(QL.h(), qp.values, (uint64_t)qp.params.num_bytes(), own_vertex(qp, Q_Base_Points),
                                      starting_points.data(), starting_points.size(), QP);
  }
  const auto& beamElts = r.first.first;
  if ((long)beamElts.size() < QP.k) {
    std::cout << "Error: for point id " << p.id() << " beam search returned " << beamElts.size() << " elements, which is less than k = " << QP.k << std::endl;
    abort();
  }
  if (stats_) { QueryStats.increment_visited((indexType)p.id(), (indexType)r.first.second.size()); QueryStats.increment_dist((indexType)p.id(), (indexType)r.second); }
  const uint32_t c = use_rerank ? (uint32_t)std::min<long>((long)QP.k * QP.rerank_factor, (long)beamElts.size()) : (uint32_t)QP.k;
  std::vector<uint32_t> cand(c), oi(QP.k);
  std::vector<float> od(QP.k);
  for (uint32_t i = 0; i < c; i++) cand[i] = beamElts[i].first;
  auto L = device_mirror(G, Base_Points);
  pann_check(pann_rerank(L.h(), p.values, 1, (uint64_t)p.params.num_bytes(), cand.data(), c, nullptr, (uint32_t)QP.k, use_rerank ? 1 : 0,
                         oi.data(), od.data()));
  id_dist_seq<indexType> out(QP.k);
  for (long i = 0; i < QP.k; i++) out[i] = std::make_pair((indexType)oi[i], od[i]);
  return out;
}

// beam_search_rerank__(p, qp, G, Base_Points, Q_Base_Points, starting_point, QP) -> (visited list, dist_cmps): the
// build-time search (:499-521; vamana/index.h:250-259 calls it with the same range twice)
template <typename Point, typename QPoint, typename PointRange, typename QPointRange, typename indexType>
std::pair<id_dist_seq<indexType>, indexType> beam_search_rerank__(const Point& p, const QPoint& qp, const Graph<indexType>& G,
                                                                  const PointRange& Base_Points, const QPointRange& Q_Base_Points,
                                                                  indexType starting_point, const QueryParams& QP) {
  parlay::sequence<indexType> starting_points = {starting_point};
  const bool use_filtering = Base_Points.params.num_bytes() != Q_Base_Points.params.num_bytes();
  auto r = filtered_beam_search(G, p, Base_Points, qp, Q_Base_Points, starting_points, QP, use_filtering);
  return std::make_pair(r.first.second, (indexType)r.second);
}

// qsearchAll(Query_Points, Q_Query_Points, QQ_Query_Points, G, Base_Points, Q_Base_Points, QQ_Base_Points, QueryStats,
//            starting_point, QP)   (:537-565)
template <typename PointRange, typename QPointRange, typename QQPointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> qsearchAll(const PointRange& Query_Points, const QPointRange& Q_Query_Points,
                                                         const QQPointRange& /*QQ_Query_Points*/, const Graph<indexType>& G,
                                                         const PointRange& Base_Points, const QPointRange& Q_Base_Points,
                                                         const QQPointRange& QQ_Base_Points, stats<indexType>& QueryStats,
                                                         const indexType starting_point, const QueryParams& QP,
                                                         std::vector<float>* dists_out = nullptr) {
  detail::check_k_le_beam(QP);
  if (Q_Base_Points.params.num_bytes() != QQ_Base_Points.params.num_bytes()) {
    std::cout << "Error: qsearchAll with a second-level filter range is not mirrored on the device" << std::endl; abort();
  }
  if ((const void*)Base_Points.data() == (const void*)Q_Base_Points.data()) {      // un-quantised: one mirror serves both steps
    auto L = device_mirror(G, Base_Points);
    return detail::search_rerank_batch<PointRange, QPointRange, indexType>(L.h(), L.h(), Query_Points, Q_Query_Points, QueryStats,
                                                                            starting_point, QP, true, dists_out);
  }
  auto QL = device_mirror(G, Q_Base_Points);
  auto L = device_mirror(G, Base_Points);
  return detail::search_rerank_batch<PointRange, QPointRange, indexType>(L.h(), QL.h(), Query_Points, Q_Query_Points, QueryStats,
                                                                          starting_point, QP, true, dists_out);
}

// ============================================================================================================
// the same calls on an explicitly managed mirror
// ============================================================================================================

template <class PointRange, typename indexType>
beam_result<indexType> beam_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI,
                                   const std::vector<indexType>& starting_points, const QueryParams& QP) {
  return detail::search_one<indexType>(DI.h, p.values, (uint64_t)p.params.num_bytes(), -1, starting_points.data(), starting_points.size(), QP);
}
template <class PointRange, typename indexType>
beam_result<indexType> beam_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI, const indexType starting_point,
                                   const QueryParams& QP) {
  std::vector<indexType> s = {starting_point};
  return beam_search<PointRange, indexType>(p, DI, s, QP);
}

template <class PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> searchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                                        stats<indexType>& QueryStats, const std::vector<indexType>& starting_points,
                                                        QueryParams& QP, std::vector<float>* dists_out = nullptr) {
  return detail::search_batch<PointRange, indexType>(DI.h, Query_Points, QueryStats, starting_points.data(), starting_points.size(),
                                                     false, QP, dists_out);
}
template <class PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> searchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                                        stats<indexType>& QueryStats, indexType starting_point, QueryParams& QP) {
  std::vector<indexType> s = {starting_point};
  return searchAll<PointRange, indexType>(Query_Points, DI, QueryStats, s, QP);
}

template <class PointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> qsearchAll(PointRange& Query_Points, DeviceIndex<PointRange, indexType>& DI,
                                                         stats<indexType>& QueryStats, const indexType starting_point,
                                                         const QueryParams& QP) {
  QueryParams q = QP;
  return searchAll<PointRange, indexType>(Query_Points, DI, QueryStats, starting_point, q);
}
template <class PointRange, class QPointRange, typename indexType>
parlay::sequence<parlay::sequence<indexType>> qsearchAll(PointRange& Query_Points, QPointRange& Q_Query_Points,
                                                         DeviceIndex<PointRange, indexType>& DI, DeviceIndex<QPointRange, indexType>& QDI,
                                                         stats<indexType>& QueryStats, const indexType starting_point,
                                                         const QueryParams& QP, std::vector<float>* dists_out = nullptr) {
  return detail::search_rerank_batch<PointRange, QPointRange, indexType>(DI.h, QDI.h, Query_Points, Q_Query_Points, QueryStats,
                                                                          starting_point, QP, true, dists_out);
}

template <class PointRange, typename indexType>
std::pair<std::vector<indexType>, long> range_search(const typename PointRange::Point p, DeviceIndex<PointRange, indexType>& DI,
                                                     const std::vector<indexType>& starting_points, float /*radius: unused, :250*/,
                                                     float radius_2, const QueryParams& /*QP*/, long self = -1) {
  return detail::range_one<indexType>(DI.h, p.values, (uint64_t)p.params.num_bytes(), self, starting_points.data(), starting_points.size(), radius_2);
}

// the loop of vamana/neighbors.h:95-101 as ONE launch: base point i searches from starts[i] (a row of nstarts ids,
// 0xFFFFFFFF = padding); returns (counts, distance comparisons) per point
inline std::pair<std::vector<long>, std::vector<long>> self_range_search(pann_index* h, size_t n, const std::vector<uint32_t>& starts,
                                                                         uint32_t nstarts, float radius_2, uint32_t max_results = 1024) {
  std::vector<uint32_t> qid(n), ids(n * (size_t)max_results), cnt(n), cmps(n), trunc(n);
  for (size_t i = 0; i < n; i++) qid[i] = (uint32_t)i;
  pann_check(pann_range_search(h, nullptr, qid.data(), n, 0, starts.data(), nstarts, 1, radius_2, max_results, ids.data(),
                               cnt.data(), cmps.data(), trunc.data()));
  return {std::vector<long>(cnt.begin(), cnt.end()), std::vector<long>(cmps.begin(), cmps.end())};
}
template <class PointRange, typename indexType>
std::pair<std::vector<long>, std::vector<long>> self_range_search(DeviceIndex<PointRange, indexType>& DI, size_t n,
                                                                  const std::vector<uint32_t>& starts, uint32_t nstarts,
                                                                  float radius_2, uint32_t max_results = 1024) {
  return self_range_search(DI.h, n, starts, nstarts, radius_2, max_results);
}

}  // namespace parlayANN
