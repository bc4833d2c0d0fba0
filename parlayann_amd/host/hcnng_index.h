// hcnng_index.h -- host mirror of algorithms/HCNNG/hcnng_index.h + clusterEdge.h.
//
//   cluster::random_clustering / recurse  clusterEdge.h:66-132  -> level-synchronous tree on the host, the
//        two-pivot distance tests of every splitting cluster in ONE pann_pivot_split call per level
//   hcnng_index::MSTk                     hcnng_index.h:134-229 -> all-pairs + 10-NN of EVERY leaf of a tree
//        in one pann_leaf_knn_batch call; de-duplication, degree-bounded Kruskal and process_edges
//        (:117-131,:202-228) stay on the host (threads over leaves), as SURVEY section 2 #8 scopes it
//   hcnng_index::build_index              hcnng_index.h:273-281
//
// Deviations, all forced or harmless: the reference seeds each tree from std::random_device
// (clusterEdge.h:137-140); here the seed is explicit.  A split that would leave one side empty
// (possible for MIPS, where d(s,s) < d(s,f) is not guaranteed) falls back to the reference's
// "split in half" rule (:108-127) instead of recursing forever.  Ties in the per-row 10-NN are
// broken by point id (the reference's heap order is unspecified).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <thread>
#include <tuple>
#include <vector>

#include "device_index.h"

namespace parlayANN {

namespace hcnng_detail {
inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
struct rnd_t {                       // same roles as parlay::random ith_rand / fork (clusterEdge.h:43-44,85-86)
  uint64_t state;
  uint64_t ith_rand(uint64_t i) const { return mix64(state + i); }
  rnd_t fork(uint64_t i) const { return rnd_t{mix64(mix64(state) + i + 17)}; }
};
template <typename F>
inline void parallel_for(size_t lo, size_t hi, F f) {
  const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (hi - lo < 2 || nt == 1) { for (size_t i = lo; i < hi; i++) f(i); return; }
  std::atomic<size_t> next(lo);
  std::vector<std::thread> ts;
  for (unsigned t = 0; t < nt; t++)
    ts.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < hi;) f(i); });
  for (auto& t : ts) t.join();
}
// hcnng_index.h:36-89, including _union's use of rank[x] rather than rank[root]
struct DisjointSet {
  std::vector<int> parent, rank;
  size_t N;
  explicit DisjointSet(size_t size) : parent(size), rank(size, 0), N(size) { for (size_t i = 0; i < N; i++) parent[i] = (int)i; }
  int find(int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; }
  void _union(int x, int y) {
    const int xroot = find(x), yroot = find(y);
    const int xrank = rank[x], yrank = rank[y];
    if (xroot == yroot) return;
    else if (xrank < yrank) parent[xroot] = yroot;
    else { parent[yroot] = xroot; if (xrank == yrank) rank[xroot] = rank[xroot] + 1; }
  }
  bool is_full() { const int r = find(0); for (size_t i = 1; i < N; i++) if (find((int)i) != r) return false; return true; }
};
}  // namespace hcnng_detail

template <typename Point, typename PointRange, typename indexType>
struct hcnng_index {
  using GraphI = Graph<indexType>;
  using DI = DeviceIndex<PointRange, indexType>;
  using edge = std::pair<indexType, indexType>;
  uint64_t seed = 1;
  int device = 0;
  double t_tree_s = 0, t_leaf_s = 0, t_mst_s = 0;

  hcnng_index() {}

  // One cluster tree (clusterEdge.h:99-144): returns the leaves as (ids, offsets)
  void cluster_tree(DI& D, PointRange& Points, size_t cluster_size, uint64_t tree_seed, std::vector<uint32_t>& ids,
                    std::vector<uint64_t>& leaf_off) {
    using namespace hcnng_detail;
    const size_t n = Points.size();
    struct Cl { uint64_t lo, len; rnd_t rnd; };
    std::vector<uint32_t> cur(n), nxt(n);
    for (size_t i = 0; i < n; i++) cur[i] = (uint32_t)i;
    std::vector<Cl> level = {Cl{0, n, rnd_t{mix64(tree_seed)}}};
    std::vector<std::pair<uint64_t, uint64_t>> leaves;   // (lo, len) in `done`
    std::vector<uint32_t> done; done.reserve(n);
    while (!level.empty()) {
      // leaves leave the level; the rest choose pivots (select_two_random :40-50)
      std::vector<Cl> split; std::vector<uint32_t> pa, pb; std::vector<uint8_t> halve;
      for (const Cl& c : level) {
        if (c.len <= cluster_size) {
          leaves.push_back({done.size(), c.len});
          done.insert(done.end(), cur.begin() + c.lo, cur.begin() + c.lo + c.len);
          continue;
        }
        const size_t fi = c.rnd.ith_rand(0) % c.len;
        const size_t su = c.rnd.ith_rand(1) % (c.len - 1);
        const size_t si = (su < fi) ? su : su + 1;
        const uint32_t f = cur[c.lo + fi], s = cur[c.lo + si];
        split.push_back(c); pa.push_back(f); pb.push_back(s);
        halve.push_back(Points[f] == Points[s] ? 1 : 0);                       // :107
      }
      if (split.empty()) break;
      // one device call: which pivot is closer, for every member of every splitting cluster (:71-83)
      std::vector<uint32_t> sid; std::vector<uint64_t> soff = {0};
      for (const Cl& c : split) { sid.insert(sid.end(), cur.begin() + c.lo, cur.begin() + c.lo + c.len); soff.push_back(sid.size()); }
      std::vector<uint8_t> side(sid.size());
      pann_check(pann_pivot_split(D.h, sid.data(), soff.data(), split.size(), pa.data(), pb.data(), side.data()));
      std::vector<Cl> next_level(2 * split.size());
      parallel_for(0, split.size(), [&](size_t ci) {
        const Cl& c = split[ci];
        const uint32_t* src = sid.data() + soff[ci];
        const uint8_t* sd = side.data() + soff[ci];
        size_t n0 = 0;
        for (size_t i = 0; i < c.len; i++) n0 += (sd[i] == 0);
        bool half = halve[ci] || n0 == 0 || n0 == c.len;
        if (half) n0 = c.len / 2;                                              // :108-115
        uint32_t* dst = nxt.data() + c.lo;
        size_t w0 = 0, w1 = n0;
        for (size_t i = 0; i < c.len; i++) {
          const bool first = half ? (i < c.len / 2) : (sd[i] == 0);
          if (first) dst[w0++] = src[i]; else dst[w1++] = src[i];              // parlay::filter keeps order
        }
        next_level[2 * ci] = Cl{c.lo, n0, c.rnd.fork(0)};                      // :85-86
        next_level[2 * ci + 1] = Cl{c.lo + n0, c.len - n0, c.rnd.fork(1)};
      });
      for (const Cl& c : split) std::memcpy(cur.data() + c.lo, nxt.data() + c.lo, c.len * 4);
      level.swap(next_level);
    }
    ids.swap(done);
    leaf_off.assign(1, 0);
    for (auto& l : leaves) leaf_off.push_back(l.first + l.second);
  }

  // MSTk for every leaf of one tree (hcnng_index.h:134-229)
  void mst_leaves(GraphI& G, DI& D, const std::vector<uint32_t>& ids, const std::vector<uint64_t>& leaf_off, long MSTDeg) {
    using namespace hcnng_detail;
    const uint32_t m = 10;                                                     // :140
    const size_t total = ids.size(), nleaves = leaf_off.size() - 1;
    std::vector<uint32_t> nn_ids(total * m); std::vector<float> nn_d(total * m);
    auto t0 = std::chrono::steady_clock::now();
    pann_check(pann_leaf_knn_batch(D.h, ids.data(), leaf_off.data(), nleaves, m, nn_ids.data(), nn_d.data()));
    auto t1 = std::chrono::steady_clock::now();
    std::vector<uint32_t> pos(G.size());
    for (size_t i = 0; i < total; i++) pos[ids[i]] = (uint32_t)i;
    const long maxDeg = G.max_degree();
    parallel_for(0, nleaves, [&](size_t li) {
      const size_t lo = leaf_off[li], N = leaf_off[li + 1] - lo;
      if (N < 2) return;
      using ledge = std::tuple<float, int, int>;                               // (dist, i, j) with i < j, local indices
      std::vector<ledge> edges;
      edges.reserve(N * m);
      for (size_t i = 0; i < N; i++)
        for (uint32_t t = 0; t < m; t++) {
          const uint32_t nb = nn_ids[(lo + i) * m + t];
          if (nb == 0xFFFFFFFFu) break;
          const int j = (int)(pos[nb] - lo);
          edges.emplace_back(nn_d[(lo + i) * m + t], std::min((int)i, j), std::max((int)i, j));   // :160-170
        }
      std::sort(edges.begin(), edges.end());                                   // less_dup order (:183-201)
      edges.erase(std::unique(edges.begin(), edges.end()), edges.end());      // remove_duplicates_ordered (:202-203)
      DisjointSet ds(N);
      std::vector<int> degrees(N, 0);
      std::vector<edge> mst;
      for (size_t e = 0; e < edges.size(); e++) {                              // :208-226
        const int a = std::get<1>(edges[e]), b = std::get<2>(edges[e]);
        if (ds.find(a) != ds.find(b) && degrees[a] < MSTDeg && degrees[b] < MSTDeg) {
          mst.push_back({ids[lo + a], ids[lo + b]});
          mst.push_back({ids[lo + b], ids[lo + a]});
          degrees[a]++; degrees[b]++;
          ds._union(a, b);
        }
        if (e % N == 0 && ds.is_full()) break;
      }
      // process_edges (:117-131): leaves of one tree are vertex-disjoint, so rows are private here.
      // (remove_edge_duplicates :102-109 writes back the unfiltered list, i.e. it is a no-op.)
      for (auto& ed : mst) {
        auto row = G[ed.first];
        if ((long)row.size() < maxDeg) row.append_neighbor(ed.second);
      }
    });
    auto t2 = std::chrono::steady_clock::now();
    t_leaf_s += std::chrono::duration<double>(t1 - t0).count();
    t_mst_s += std::chrono::duration<double>(t2 - t1).count();
  }

  // build_index(G, Points, cluster_rounds, cluster_size, MSTDeg)   (:273-281): trees, leaf kNN and Kruskal all on the
  // device (one pann_hcnng_build call on the mirror of (G, Points)); the edges are appended to G's rows
  void build_index(GraphI& G, PointRange& Points, long cluster_rounds, long cluster_size, long MSTDeg) {
    auto L = device_mirror(G, Points);
    double t3[3] = {0, 0, 0};
    pann_check(pann_hcnng_build(L.h(), (uint32_t)cluster_rounds, (uint32_t)cluster_size, (uint32_t)MSTDeg, seed, t3));
    t_tree_s += t3[0]; t_leaf_s += t3[1]; t_mst_s += t3[2];
    MirrorCache::download_graph(L, G);
    // remove_all_duplicates (:111-114) is a no-op in the reference (remove_edge_duplicates writes back the
    // unfiltered list, :102-109); kept as such.
  }
  void build_index_on_device(GraphI& G, PointRange& Points, long cluster_rounds, long cluster_size, long MSTDeg) {
    build_index(G, Points, cluster_rounds, cluster_size, MSTDeg);
  }

  // The same graph with the reference's host structure kept: cluster tree recursion and Kruskal on the host
  // (threads over clusters / leaves), the distance work of every level and of every leaf in batched device calls
  // (pann_pivot_split, pann_leaf_knn_batch).  ~10x slower than build_index at 10M points; a cross-check of it.
  void build_index_host_tree(GraphI& G, PointRange& Points, long cluster_rounds, long cluster_size, long MSTDeg) {
    DI D(Points, nullptr, G.max_degree(), device);
    for (long r = 0; r < cluster_rounds; r++) {
      std::vector<uint32_t> ids; std::vector<uint64_t> off;
      auto t0 = std::chrono::steady_clock::now();
      cluster_tree(D, Points, (size_t)cluster_size, hcnng_detail::mix64(seed + (uint64_t)r), ids, off);
      t_tree_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      mst_leaves(G, D, ids, off, MSTDeg);
      std::cout << "Built cluster " << r << " of " << cluster_rounds << std::endl;   // clusterEdge.h:151
    }
    G.touch();
  }
};

}  // namespace parlayANN
