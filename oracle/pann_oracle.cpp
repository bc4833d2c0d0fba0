// pann_oracle.cpp -- CPU restatement of ParlayANN's beam-search / robustPrune / batch_insert /
// HCNNG-leaf path.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the product (parlayann_amd, libpann.so)
// never links, loads or calls it.
//
// PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or fixtures for this
// path (its only tests are placeholders, algorithms/vamana/index_test.cc:1-5), and the reference
// itself cannot be built here: every header includes parlaylib, an un-vendored dependency
// (empty submodule, .gitmodules:1-3; CMakeLists.txt:14-22 fetches GIT_TAG master).  This file is
// therefore a restatement written from reading the reference sources, each function citing the
// file:line it follows; the one third-party function that influences search results,
// parlay::hash64_2 (called at beamSearch.h:55), is restated from parlaylib's published
// include/parlay/utilities.h (splitmix64 finaliser) and isolated in hash64_2() below.
//
// Plain C++17, no dependencies; threads over queries with std::thread exactly where the reference
// has parlay::parallel_for (beamSearch.h:374,556; vamana/index.h:247,268,289).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <unordered_set>
#include <vector>
#include <chrono>

namespace {

enum { DT_U8 = 0, DT_I8 = 1, DT_F32 = 2, DT_F16 = 3, DT_BF16 = 4 };
enum { M_L2 = 0, M_MIPS = 1 };

// parlaylib include/parlay/utilities.h hash64_2 (used at beamSearch.h:55)
inline uint64_t hash64_2(uint64_t x) {
  x = (x ^ (x >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  x = (x ^ (x >> 27)) * UINT64_C(0x94d049bb133111eb);
  x = x ^ (x >> 31);
  return x;
}

// IEEE binary16 -> binary32, exact (the F16 extension stores halves; arithmetic is in f32)
inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal
      int e = -1;
      do { e++; man <<= 1; } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// bfloat16 -> binary32: the 16 bits are the float's upper half, exact
inline float bf16_to_float(uint16_t h) {
  const uint32_t bits = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// euclidian_point.h:54-62 (u8), :74-81 (i8), :83-90 (f32); mips_point.h:43-65.
// Integer types accumulate in int32 and cast once; float accumulates strictly left to right with
// one rounding per multiply and per add (this file is compiled with -ffp-contract=off).
float distance(int dtype, int metric, const void* pa, const void* pb, unsigned d) {
  if (dtype == DT_U8) {
    const uint8_t* p = (const uint8_t*)pa; const uint8_t* q = (const uint8_t*)pb;
    int32_t r = 0;
    if (metric == M_L2) {
      for (unsigned i = 0; i < d; i++) { int32_t t = (int32_t)p[i] - (int32_t)q[i]; r += t * t; }
      return (float)r;
    }
    for (unsigned i = 0; i < d; i++) r += (int32_t)p[i] * (int32_t)q[i];
    return -((float)r);
  }
  if (dtype == DT_I8) {
    const int8_t* p = (const int8_t*)pa; const int8_t* q = (const int8_t*)pb;
    int32_t r = 0;
    if (metric == M_L2) {
      for (unsigned i = 0; i < d; i++) { int32_t t = (int32_t)q[i] - (int32_t)p[i]; r += t * t; }
      return (float)r;
    }
    for (unsigned i = 0; i < d; i++) r += (int32_t)q[i] * (int32_t)p[i];
    return -((float)r);
  }
  if (dtype == DT_F32) {
    const float* p = (const float*)pa; const float* q = (const float*)pb;
    float r = 0.0f;
    if (metric == M_L2) {
      for (unsigned i = 0; i < d; i++) { float t = q[i] - p[i]; r += t * t; }
      return r;
    }
    for (unsigned i = 0; i < d; i++) r += q[i] * p[i];
    return -r;
  }
  // DT_F16 / DT_BF16 (extensions of this build: two-byte storage, arithmetic in f32): convert exactly, then the f32 rule
  const uint16_t* p = (const uint16_t*)pa; const uint16_t* q = (const uint16_t*)pb;
  auto cv = [dtype](uint16_t h) { return dtype == DT_BF16 ? bf16_to_float(h) : half_to_float(h); };
  float r = 0.0f;
  if (metric == M_L2) {
    for (unsigned i = 0; i < d; i++) { float t = cv(q[i]) - cv(p[i]); r += t * t; }
    return r;
  }
  for (unsigned i = 0; i < d; i++) r += cv(q[i]) * cv(p[i]);
  return -r;
}

// diagnostic counters (tools/ only): iterations that skipped the merge (:162-168) / merges, summed over all searches
static std::atomic<uint64_t> g_dbg_skips{0}, g_dbg_merges{0};

struct IdDist { uint32_t id; float dist; };
// total order used everywhere: beamSearch.h:46-48, vamana/index.h:80-82
inline bool less_id_dist(const IdDist& a, const IdDist& b) {
  return a.dist < b.dist || (a.dist == b.dist && a.id < b.id);
}

struct Dataset {
  const uint8_t* points; uint64_t n; uint32_t d; int dtype; uint64_t stride; int metric;
  const uint32_t* graph; uint32_t maxdeg;  // reference layout: n x (maxdeg+1), slot 0 = degree
  const void* row(uint64_t i) const { return points + i * stride; }
  const uint32_t* grow(uint64_t i) const { return graph + i * (uint64_t)(maxdeg + 1); }
  float dist(const void* q, uint32_t i) const { return distance(dtype, metric, row(i), q, d); }
  float dist_ids(uint32_t a, uint32_t b) const { return distance(dtype, metric, row(a), row(b), d); }
};

struct SearchParams { int64_t k, beam; double cut; int64_t limit, degree_limit; };

struct SearchResult {
  std::vector<IdDist> frontier;  // sorted by (dist,id), <= beam
  std::vector<IdDist> visited;   // sorted by (dist,id) (beamSearch.h:112-113)
  std::vector<IdDist> visit_order;  // same elements, in the order they were visited
  uint64_t dist_cmps = 0;        // full_dist_cmps (beamSearch.h:213); == dist_cmps without filtering
  uint64_t degree_sum = 0;       // sum of min(deg, degree_limit) over visited (roofline numerator)
};

// beamSearch.h:22-214 with use_filtering == false (the only mode reachable for the in-scope
// configurations).  self_id >= 0 reproduces Points[a].same_as(p)
// (:133) for build-time searches where p is a base point.
void beam_search(const Dataset& D, const void* q, int64_t self_id, const uint32_t* starts,
                 uint32_t nstarts, const SearchParams& QP, SearchResult& R) {
  const int beam = (int)QP.beam;
  // :52-59 lossy direct-mapped filter
  int bits = std::max<int>(10, (int)std::ceil(std::log2((double)beam * (double)beam)) - 2);
  std::vector<uint32_t> table((size_t)1 << bits, 0xFFFFFFFFu);
  const uint64_t mask = ((uint64_t)1 << bits) - 1;
  auto seen = [&](uint32_t a) -> bool {
    size_t loc = (size_t)(hash64_2((uint64_t)a) & mask);
    if (table[loc] == a) return true;
    table[loc] = a;
    return false;
  };

  std::vector<IdDist>& frontier = R.frontier;
  frontier.clear(); R.visited.clear(); R.visit_order.clear(); R.degree_sum = 0;
  frontier.reserve(beam);
  for (uint32_t s = 0; s < nstarts; s++) {  // :66-70
    frontier.push_back(IdDist{starts[s], D.dist(q, starts[s])});
    seen(starts[s]);
  }
  std::sort(frontier.begin(), frontier.end(), less_id_dist);

  std::vector<IdDist> unvisited(std::max<size_t>(beam, nstarts));  // :74-76
  for (size_t i = 0; i < frontier.size(); i++) unvisited[i] = frontier[i];

  std::vector<IdDist>& visited = R.visited;
  uint64_t dist_cmps = nstarts;  // :83-84
  int remain = (int)frontier.size();
  int64_t num_visited = 0;
  std::vector<IdDist> merged(2 * std::max<size_t>(beam, nstarts) + D.maxdeg);  // :89-90
  std::vector<IdDist> cand;
  std::vector<uint32_t> keep;
  int offset = 0;
  const float big = (float)std::numeric_limits<int>::max();  // :152

  while (remain > offset && num_visited < QP.limit) {  // :107
    IdDist cur = unvisited[offset];
    __builtin_prefetch(D.grow(cur.id));               // G[current.first].prefetch() (:110)
    visited.insert(std::upper_bound(visited.begin(), visited.end(), cur, less_id_dist), cur);
    R.visit_order.push_back(cur);
    num_visited++;
    bool full = (int)frontier.size() == beam;  // :115

    keep.clear();
    const uint32_t* row = D.grow(cur.id);
    int64_t ne = std::min<int64_t>((int64_t)row[0], QP.degree_limit);  // :130
    R.degree_sum += (uint64_t)std::max<int64_t>(ne, 0);
    for (int64_t i = 0; i < ne; i++) {
      uint32_t a = row[1 + i];
      if (seen(a) || (int64_t)a == self_id) continue;  // :133 (filter is updated before same_as)
      // Q_Points[a].prefetch() (:134; euclidian_point.h:129-133): one prefetch per 64-byte line of the row
      for (uint64_t off = 0; off < (uint64_t)D.d * (D.dtype == DT_F32 ? 4 : (D.dtype == DT_F16 || D.dtype == DT_BF16) ? 2 : 1); off += 64)
        __builtin_prefetch((const char*)D.row(a) + off);
      keep.push_back(a);
    }
    dist_cmps += keep.size();  // :137 (and :155: one full distance per survivor)

    float cutoff = full ? frontier.back().dist : big;  // :150-152
    for (uint32_t a : keep) {
      float dist = D.dist(q, a);
      if (dist >= cutoff) continue;  // :157
      cand.push_back(IdDist{a, dist});
    }
    // :162-168 -- note: candidates persist across skipped iterations
    if (cand.empty() ||
        (QP.limit >= 2 * (int64_t)beam && (int64_t)cand.size() < beam / 8 && offset + 1 < remain)) {
      offset++;
      g_dbg_skips.fetch_add(1, std::memory_order_relaxed);
      continue;
    }
    offset = 0;

    std::sort(cand.begin(), cand.end(), less_id_dist);  // :173
    auto cend = std::unique(cand.begin(), cand.end(),
                            [](const IdDist& a, const IdDist& b) { return a.id == b.id; });
    size_t msize = std::set_union(frontier.begin(), frontier.end(), cand.begin(), cend,
                                  merged.begin(), less_id_dist) - merged.begin();  // :178-181
    cand.clear();
    g_dbg_merges.fetch_add(1, std::memory_order_relaxed);
    msize = std::min<size_t>((size_t)beam, msize);  // :185

    if (QP.k > 0 && (int64_t)msize > QP.k && D.metric == M_L2) {  // :190 (is_metric(): L2 only)
      // :191-195: the bound is pair{0, cut * merged[k].dist}; the product is formed in double and
      // narrowed to float when the pair is converted to (indexType, float) for `less`.
      IdDist thr{0u, (float)(QP.cut * (double)merged[QP.k].dist)};
      size_t ub = std::upper_bound(merged.begin(), merged.begin() + msize, thr, less_id_dist) -
                  merged.begin();
      msize = std::max<size_t>(ub, frontier.size());
    }
    frontier.assign(merged.begin(), merged.begin() + msize);  // :198-200

    remain = (int)(std::set_difference(frontier.begin(),
                                       frontier.begin() + std::min<size_t>(frontier.size(), beam),
                                       visited.begin(), visited.end(), unvisited.begin(),
                                       less_id_dist) - unvisited.begin());  // :203-208
  }
  R.dist_cmps = dist_cmps;
}

// vamana/index.h:63-120.  cand: (id, dist to p).  Returns new neighbours, adds to *dcmps.
void robust_prune(const Dataset& D, uint32_t p, std::vector<IdDist>& cand, double alpha, uint32_t R,
                  bool add, std::vector<uint32_t>& out, uint64_t* dcmps) {
  uint64_t dc = 0;
  if (add) {  // :72-77
    const uint32_t* row = D.grow(p);
    for (uint32_t i = 0; i < row[0]; i++) {
      dc++;
      cand.push_back(IdDist{row[1 + i], D.dist_ids(row[1 + i], p)});
    }
  }
  std::sort(cand.begin(), cand.end(), less_id_dist);  // :83
  cand.erase(std::unique(cand.begin(), cand.end(),
                         [](const IdDist& a, const IdDist& b) { return a.id == b.id; }),
             cand.end());  // :86-88
  out.clear();
  size_t idx = 0;
  const uint32_t DEAD = 0xFFFFFFFFu;  // the reference's -1 sentinel (:99,112)
  while (out.size() < R && idx < cand.size()) {  // :95
    uint32_t ps = cand[idx].id;
    idx++;
    if (ps == p || ps == DEAD) continue;
    out.push_back(ps);
    for (size_t i = idx; i < cand.size(); i++) {  // :105-115
      uint32_t pp = cand[i].id;
      if (pp != DEAD) {
        dc++;
        float d_sp = D.dist_ids(ps, pp);
        float d_pp = cand[i].dist;
        if (alpha * (double)d_sp <= (double)d_pp) cand[i].id = DEAD;  // :111, in double
      }
    }
  }
  if (dcmps) *dcmps += dc;
}

template <typename F>
void parallel_for(size_t lo, size_t hi, int nthreads, F f) {
  if (nthreads <= 1 || hi - lo < 2) { for (size_t i = lo; i < hi; i++) f(i); return; }
  std::atomic<size_t> next(lo);
  const size_t chunk = std::max<size_t>(1, (hi - lo) / ((size_t)nthreads * 16));
  std::vector<std::thread> ts;
  for (int t = 0; t < nthreads; t++)
    ts.emplace_back([&]() {
      for (;;) {
        size_t b = next.fetch_add(chunk);
        if (b >= hi) break;
        size_t e = std::min(hi, b + chunk);
        for (size_t i = b; i < e; i++) f(i);
      }
    });
  for (auto& t : ts) t.join();
}

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += UINT64_C(0x9e3779b97f4a7c15));
  z = (z ^ (z >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  z = (z ^ (z >> 27)) * UINT64_C(0x94d049bb133111eb);
  return z ^ (z >> 31);
}

}  // namespace

extern "C" {

void pann_oracle_debug_counters(uint64_t* out2, int reset) {
  out2[0] = g_dbg_skips.load(); out2[1] = g_dbg_merges.load();
  if (reset) { g_dbg_skips = 0; g_dbg_merges = 0; }
}

int pann_oracle_hw_threads(void) { return (int)std::thread::hardware_concurrency(); }

uint64_t pann_oracle_hash64_2(uint64_t x) { return hash64_2(x); }

float pann_oracle_distance(int dtype, int metric, const void* a, const void* b, uint32_t d) {
  return distance(dtype, metric, a, b, d);
}

// Batched beam search = searchAll / qsearchAll (beamSearch.h:353-387,537-565): one task per query.
// queries: nq rows (stride q_stride) or query_ids (base points; self excluded).  Outputs as in
// pann_search_out, except that visited lists are returned SORTED by (dist,id) (reference order)
// in visited_* and in visit order in visit_order_* (either may be NULL).
int pann_oracle_batch_search(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride,
                             int metric, const uint32_t* graph, uint32_t maxdeg,
                             const void* queries, const uint32_t* query_ids, uint64_t nq,
                             uint64_t q_stride, const uint32_t* starts, uint32_t nstarts,
                             int64_t k, int64_t beam, double cut, int64_t limit,
                             int64_t degree_limit, uint32_t out_k, uint32_t* out_ids,
                             float* out_dists, uint32_t* frontier_size, uint32_t* visited_count,
                             uint32_t* dist_cmps, uint32_t* degree_sum, uint32_t visited_cap,
                             uint32_t* visited_ids, float* visited_dists,
                             uint32_t* visit_order_ids, int nthreads) {
  if (nstarts == 0 || beam <= 0 || nstarts > (uint32_t)beam) return 1;
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
  SearchParams QP{k, beam, cut, limit, degree_limit};
  std::atomic<int> overflow(0);
  parallel_for(0, nq, nthreads, [&](size_t i) {
    SearchResult R;
    const void* q = query_ids ? D.row(query_ids[i]) : (const void*)((const uint8_t*)queries + i * q_stride);
    beam_search(D, q, query_ids ? (int64_t)query_ids[i] : -1, starts, nstarts, QP, R);
    for (uint32_t j = 0; j < out_k; j++) {
      bool ok = j < R.frontier.size();
      if (out_ids) out_ids[i * out_k + j] = ok ? R.frontier[j].id : 0xFFFFFFFFu;
      if (out_dists) out_dists[i * out_k + j] = ok ? R.frontier[j].dist : std::numeric_limits<float>::infinity();
    }
    if (frontier_size) frontier_size[i] = (uint32_t)R.frontier.size();
    if (visited_count) visited_count[i] = (uint32_t)R.visited.size();
    if (dist_cmps) dist_cmps[i] = (uint32_t)R.dist_cmps;
    if (degree_sum) degree_sum[i] = (uint32_t)R.degree_sum;
    if (visited_cap) {
      if (R.visited.size() > visited_cap) overflow = 1;
      size_t m = std::min<size_t>(R.visited.size(), visited_cap);
      for (size_t j = 0; j < m; j++) {
        if (visited_ids) visited_ids[i * visited_cap + j] = R.visited[j].id;
        if (visited_dists) visited_dists[i * visited_cap + j] = R.visited[j].dist;
        if (visit_order_ids) visit_order_ids[i * visited_cap + j] = R.visit_order[j].id;
      }
    }
  });
  return overflow ? 5 : 0;
}

// Batched robustPrune; same contract as pann_robust_prune_batch (include/pann.h).
int pann_oracle_robust_prune_batch(const void* points, uint64_t n, uint32_t d, int dtype,
                                   uint64_t stride, int metric, const uint32_t* graph,
                                   uint32_t maxdeg, const uint32_t* owners, uint64_t m,
                                   const uint32_t* cand_ids, const float* cand_dists,
                                   const uint64_t* cand_offsets, double alpha, uint32_t R,
                                   int add_out_nbrs, uint32_t* out_rows, uint32_t* out_dist_cmps,
                                   int nthreads) {
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
  parallel_for(0, m, nthreads, [&](size_t i) {
    std::vector<IdDist> cand;
    uint64_t dc = 0;
    for (uint64_t j = cand_offsets[i]; j < cand_offsets[i + 1]; j++) {
      if (cand_dists) cand.push_back(IdDist{cand_ids[j], cand_dists[j]});
      else { dc++; cand.push_back(IdDist{cand_ids[j], D.dist_ids(cand_ids[j], owners[i])}); }  // :131-134
    }
    std::vector<uint32_t> out;
    robust_prune(D, owners[i], cand, alpha, R, add_out_nbrs != 0, out, &dc);
    uint32_t* row = out_rows + i * (uint64_t)(R + 1);
    row[0] = (uint32_t)out.size();
    for (uint32_t j = 0; j < R; j++) row[1 + j] = j < out.size() ? out[j] : 0;
    if (out_dist_cmps) out_dist_cmps[i] = (uint32_t)dc;
  });
  return 0;
}

// The insertion order used by this build (oracle and product alike; DESIGN.md "build determinism"):
// Fisher-Yates driven by splitmix64(seed).  parlay::random_permutation (vamana/index.h:212) is not
// reproducible without parlaylib, so graph identity with upstream is out of reach by construction.
void pann_oracle_permutation(uint64_t m, uint64_t seed, uint32_t* out) {
  for (uint64_t i = 0; i < m; i++) out[i] = (uint32_t)i;
  uint64_t s = seed;
  for (uint64_t i = m; i > 1; i--) {
    uint64_t j = splitmix64(s) % i;
    std::swap(out[i - 1], out[j]);
  }
}

// BuildStats of the reference (stats.h:63-73), per point: visited += |visited| (vamana/index.h:262), distances +=
// beam-search + robustPrune comparisons (:261,266) and, for a re-pruned reverse-edge target, that prune's (:298).
// Optional: the arrays (n entries, accumulated) are registered before a build / insert call.
static uint32_t* g_pp_visited = nullptr;
static uint32_t* g_pp_dists = nullptr;
void pann_oracle_set_build_point_stats(uint32_t* visited, uint32_t* dists) { g_pp_visited = visited; g_pp_dists = dists; }

// One batch of vamana/index.h:188-316 (steps 1-4) on a HOST graph in the reference layout.
// stats6: [search_dist_cmps, prune_dist_cmps, visited_total, t_search_us, t_prune_us, t_bidirect_us]
// The batch in two phases, the split the multi-GPU build makes (tests/test_distributed_cpu.py drives
// parlayann_amd/distributed.py with these as the per-rank workers): phase A = :247-266 for the points given (reads the
// graph only), rows_out m x R, unused slots 0xFFFFFFFF; phase B = :268-300 for the whole batch.
int pann_oracle_vamana_phase_a(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride, int metric,
                               const uint32_t* graph, uint32_t maxdeg, const uint32_t* batch, uint64_t m, uint32_t start,
                               uint32_t R, uint32_t L, double alpha, uint32_t* rows_out, uint64_t* stats6, int nthreads) {
  if (R > maxdeg) return 1;
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, const_cast<uint32_t*>(graph), maxdeg};
  SearchParams QP{0, (int64_t)L, 0.0, (int64_t)n, (int64_t)maxdeg};  // :250
  std::atomic<uint64_t> sdc(0), pdc(0), vis(0);
  auto t0 = std::chrono::steady_clock::now();
  parallel_for(0, m, nthreads, [&](size_t i) {  // :247-266
    uint32_t p = batch[i];
    SearchResult Rs;
    beam_search(D, D.row(p), (int64_t)p, &start, 1, QP, Rs);
    sdc += Rs.dist_cmps; vis += Rs.visited.size();
    uint64_t dc = 0;
    std::vector<IdDist> cand = Rs.visited;
    std::vector<uint32_t> out;
    robust_prune(D, p, cand, alpha, R, true, out, &dc);
    pdc += dc;
    if (g_pp_visited) g_pp_visited[p] += (uint32_t)Rs.visited.size();     // one writer per p within a batch
    if (g_pp_dists) g_pp_dists[p] += (uint32_t)(Rs.dist_cmps + dc);
    for (uint32_t j = 0; j < R; j++) rows_out[i * (uint64_t)R + j] = j < out.size() ? out[j] : 0xFFFFFFFFu;
  });
  if (stats6) {
    stats6[0] += sdc; stats6[1] += pdc; stats6[2] += vis;
    stats6[3] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  return 0;
}

int pann_oracle_vamana_phase_b(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride, int metric,
                               uint32_t* graph, uint32_t maxdeg, const uint32_t* batch, uint64_t m, const uint32_t* rows,
                               uint32_t R, double alpha, uint64_t* stats6, int nthreads) {
  if (R > maxdeg) return 1;
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
  std::atomic<uint64_t> pdc(0);
  auto t1 = std::chrono::steady_clock::now();
  const uint64_t rs = (uint64_t)maxdeg + 1;
  auto row_len = [&](size_t i) { uint32_t c = 0; while (c < R && rows[i * (uint64_t)R + c] != 0xFFFFFFFFu) c++; return c; };
  parallel_for(0, m, nthreads, [&](size_t i) {  // :268-270
    uint32_t* row = graph + batch[i] * rs;
    const uint32_t c = row_len(i);
    row[0] = c;
    for (uint32_t j = 0; j < c; j++) row[1 + j] = rows[i * (uint64_t)R + j];
  });
  // :278-282 reverse edges grouped by target.  Order inside a group (unspecified upstream):
  // ascending position of the source in the batch -- the rule the product follows too.
  std::vector<std::pair<uint32_t, uint32_t>> edges;  // (target, batch position)
  for (size_t i = 0; i < m; i++) {
    const uint32_t c = row_len(i);
    for (uint32_t j = 0; j < c; j++) edges.push_back({rows[i * (uint64_t)R + j], (uint32_t)i});
  }
  std::sort(edges.begin(), edges.end());
  std::vector<size_t> gstart;
  for (size_t e = 0; e < edges.size(); e++)
    if (e == 0 || edges[e].first != edges[e - 1].first) gstart.push_back(e);
  gstart.push_back(edges.size());
  auto t2 = std::chrono::steady_clock::now();
  parallel_for(0, gstart.size() - 1, nthreads, [&](size_t g) {  // :289-300
    uint32_t v = edges[gstart[g]].first;
    uint32_t* row = graph + v * rs;
    std::vector<uint32_t> cids;
    for (size_t e = gstart[g]; e < gstart[g + 1]; e++) cids.push_back(batch[edges[e].second]);
    size_t newsize = cids.size() + row[0];
    if (newsize <= R) {  // :292-294 add_neighbors_without_repeats(G[index], candidates)
      std::vector<uint32_t> res = cids;
      for (uint32_t j = 0; j < row[0]; j++)
        if (std::find(cids.begin(), cids.end(), row[1 + j]) == cids.end()) res.push_back(row[1 + j]);
      row[0] = (uint32_t)res.size();
      for (size_t j = 0; j < res.size(); j++) row[1 + j] = res[j];
    } else {  // :296-298 id-only robustPrune overload (:124-137)
      uint64_t dc = 0;
      std::vector<IdDist> cand;
      for (uint32_t c : cids) { dc++; cand.push_back(IdDist{c, D.dist_ids(c, v)}); }
      std::vector<uint32_t> out;
      robust_prune(D, v, cand, alpha, R, true, out, &dc);
      pdc += dc;
      if (g_pp_dists) g_pp_dists[v] += (uint32_t)dc;                        // one group per target v
      row[0] = (uint32_t)out.size();
      for (size_t j = 0; j < out.size(); j++) row[1 + j] = out[j];
    }
  });
  auto t3 = std::chrono::steady_clock::now();
  if (stats6) {
    auto us = [](auto a, auto b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    stats6[1] += pdc; stats6[4] += us(t2, t3); stats6[5] += us(t1, t2);
  }
  return 0;
}

int pann_oracle_vamana_insert_batch(const void* points, uint64_t n, uint32_t d, int dtype,
                                    uint64_t stride, int metric, uint32_t* graph, uint32_t maxdeg,
                                    const uint32_t* batch, uint64_t m, uint32_t start, uint32_t R,
                                    uint32_t L, double alpha, uint64_t* stats6, int nthreads) {
  std::vector<uint32_t> rows((size_t)m * R);
  if (int rc = pann_oracle_vamana_phase_a(points, n, d, dtype, stride, metric, graph, maxdeg, batch, m, start, R, L, alpha,
                                          rows.data(), stats6, nthreads)) return rc;
  return pann_oracle_vamana_phase_b(points, n, d, dtype, stride, metric, graph, maxdeg, batch, m, rows.data(), R, alpha, stats6, nthreads);
}

// vamana/index.h:150-186 build_index + the batch schedule of :200-234.
// build_index with BP.single_batch = degree (vamana/index.h:156-170,236-240): `degree` random out-edges per vertex, then every
// pass inserts all points as ONE batch.  The start edges come from this build's own generator (the reference's
// parlay::random_generator is not reproducible offline): edge j of vertex i = splitmix64(seed + golden * (i*degree + j + 1)) mod n.
int pann_oracle_vamana_build_single_batch(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride,
                                          int metric, uint32_t* graph, uint32_t maxdeg, uint32_t R, uint32_t L,
                                          double alpha, int num_passes, uint32_t degree, uint64_t seed, int sort_neighbors,
                                          uint64_t* stats6, int nthreads) {
  if (degree == 0 || degree > maxdeg) return 1;
  std::vector<uint32_t> perm(n);
  pann_oracle_permutation(n, seed, perm.data());
  for (uint64_t i = 0; i < n; i++) {
    uint32_t* row = graph + i * (uint64_t)(maxdeg + 1);
    row[0] = degree;
    for (uint32_t j = 0; j < degree; j++) {
      uint64_t z = seed + 0x9e3779b97f4a7c15ull * (i * degree + j + 1);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      z ^= z >> 31;
      row[1 + j] = (uint32_t)(z % n);
    }
  }
  for (int pass = 0; pass < num_passes; pass++) {
    const double a = (pass == num_passes - 1) ? alpha : 1.0;  // :173-178
    int rc = pann_oracle_vamana_insert_batch(points, n, d, dtype, stride, metric, graph, maxdeg, perm.data(), n, 0, R, L, a,
                                             stats6, nthreads);      // floor = 0, ceiling = m (:236-240)
    if (rc) return rc;
  }
  if (sort_neighbors) {  // :180-185; ties broken by id
    Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
    parallel_for(0, n, nthreads, [&](size_t i) {
      uint32_t* row = graph + i * (uint64_t)(maxdeg + 1);
      std::vector<IdDist> v;
      for (uint32_t j = 0; j < row[0]; j++) v.push_back(IdDist{row[1 + j], D.dist_ids((uint32_t)i, row[1 + j])});
      std::sort(v.begin(), v.end(), less_id_dist);
      for (uint32_t j = 0; j < row[0]; j++) row[1 + j] = v[j].id;
    });
  }
  return 0;
}

int pann_oracle_vamana_build(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride,
                             int metric, uint32_t* graph, uint32_t maxdeg, uint32_t R, uint32_t L,
                             double alpha, int num_passes, uint64_t seed, int sort_neighbors,
                             uint64_t* stats6, int nthreads) {
  std::vector<uint32_t> perm(n);
  pann_oracle_permutation(n, seed, perm.data());
  // :206-209
  size_t max_batch = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);
  if (max_batch == 0) max_batch = n;
  for (int pass = 0; pass < num_passes; pass++) {
    double a = (pass == num_passes - 1) ? alpha : 1.0;  // :173-178
    size_t count = 0, inc = 0, m = n;
    while (count < m) {  // :223-234
      size_t floor, ceiling;
      if (std::pow(2.0, (double)inc) <= (double)max_batch) {
        floor = (size_t)std::pow(2.0, (double)inc) - 1;
        ceiling = std::min((size_t)std::pow(2.0, (double)(inc + 1)) - 1, m);
        count = ceiling;
      } else {
        floor = count;
        ceiling = std::min(count + max_batch, m);
        count += max_batch;
      }
      int rc = pann_oracle_vamana_insert_batch(points, n, d, dtype, stride, metric, graph, maxdeg,
                                               perm.data() + floor, ceiling - floor, 0, R, L, a,
                                               stats6, nthreads);
      if (rc) return rc;
      inc++;
    }
  }
  if (sort_neighbors) {  // :180-185; ties broken by id so the result is a function of the input
    Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
    parallel_for(0, n, nthreads, [&](size_t i) {
      uint32_t* row = graph + i * (uint64_t)(maxdeg + 1);
      std::vector<IdDist> v;
      for (uint32_t j = 0; j < row[0]; j++) v.push_back(IdDist{row[1 + j], D.dist_ids((uint32_t)i, row[1 + j])});
      std::sort(v.begin(), v.end(), less_id_dist);
      for (uint32_t j = 0; j < row[0]; j++) row[1 + j] = v[j].id;
    });
  }
  return 0;
}

// data_tools/compute_groundtruth.cpp:22-59: exact k nearest base points per query, sorted (dist,id).
int pann_oracle_bruteforce_knn(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride,
                               int metric, const void* queries, uint64_t nq, uint64_t q_stride,
                               uint32_t k, uint32_t* out_ids, float* out_dists, int nthreads) {
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, nullptr, 0};
  parallel_for(0, nq, nthreads, [&](size_t qi) {
    const void* q = (const uint8_t*)queries + qi * q_stride;
    std::vector<IdDist> all(n);
    for (uint64_t i = 0; i < n; i++) all[i] = IdDist{(uint32_t)i, D.dist(q, (uint32_t)i)};
    size_t kk = std::min<size_t>(k, n);
    std::partial_sort(all.begin(), all.begin() + kk, all.end(), less_id_dist);
    for (uint32_t j = 0; j < k; j++) {
      out_ids[qi * k + j] = j < kk ? all[j].id : 0xFFFFFFFFu;
      out_dists[qi * k + j] = j < kk ? all[j].dist : std::numeric_limits<float>::infinity();
    }
  });
  return 0;
}

// hcnng_index.h:145-181: within one leaf (N ids) the m smallest (dist,id) among the other members.
int pann_oracle_leaf_knn(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride,
                         int metric, const uint32_t* ids, uint32_t N, uint32_t m, uint32_t* out_ids,
                         float* out_dists, int nthreads) {
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, nullptr, 0};
  parallel_for(0, N, nthreads, [&](size_t i) {
    std::vector<IdDist> v;
    for (uint32_t j = 0; j < N; j++)
      if (j != i) v.push_back(IdDist{ids[j], D.dist_ids(ids[i], ids[j])});
    size_t kk = std::min<size_t>(m, v.size());
    std::partial_sort(v.begin(), v.begin() + kk, v.end(), less_id_dist);
    for (uint32_t j = 0; j < m; j++) {
      out_ids[i * m + j] = j < kk ? v[j].id : 0xFFFFFFFFu;
      out_dists[i * m + j] = j < kk ? v[j].dist : std::numeric_limits<float>::infinity();
    }
  });
  return 0;
}

// beamSearch.h:245-306 range_search: starts within radius_2 seed `result` (:271-277), then a BFS over
// result[position++] (:280-297) with an exact `seen` set; same_as(p) (:272,287) holds only when the query is
// base point `self` (pointer equality, euclidian_point.h:178-180).  Rows of out_ids have capacity cap; a
// longer result is cut there and flagged (the product does the same; the reference's vector is unbounded).
int pann_oracle_range_search(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride, int metric,
                             const uint32_t* graph, uint32_t maxdeg, const void* queries, uint64_t q_stride,
                             const uint32_t* query_ids, uint64_t nq, const uint32_t* starts, uint32_t nstarts,
                             int starts_per_query, float radius_2, uint32_t cap, uint32_t* out_ids,
                             uint32_t* out_counts, uint32_t* out_cmps, uint32_t* out_trunc, int nthreads) {
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
  parallel_for(0, nq, nthreads, [&](size_t qi) {
    const void* q = query_ids ? D.row(query_ids[qi]) : (const void*)((const uint8_t*)queries + qi * q_stride);
    const int64_t self = query_ids ? (int64_t)query_ids[qi] : -1;
    std::vector<uint32_t> result;
    std::unordered_set<uint32_t> seen;
    long cmps = 0;
    bool trunc = false;
    const uint32_t* sp = starts + (starts_per_query ? qi * nstarts : 0);
    for (uint32_t j = 0; j < nstarts && !trunc; j++) {
      const uint32_t v = sp[j];
      if (v == 0xFFFFFFFFu) continue;                                   // padding (not in the reference)
      if (seen.count(v) > 0 || (int64_t)v == self) continue;
      cmps++;
      if (D.dist(q, v) > radius_2) continue;
      if (result.size() == cap) { trunc = true; break; }
      result.push_back(v);
      seen.insert(v);
    }
    size_t position = 0;
    while (position < result.size() && !trunc) {
      const uint32_t next = result[position++];
      const uint32_t* row = D.grow(next);
      std::vector<uint32_t> unseen;
      for (uint32_t i = 0; i < row[0]; i++) {
        const uint32_t v = row[1 + i];
        if (seen.count(v) > 0 || (int64_t)v == self) continue;
        unseen.push_back(v);
        seen.insert(v);
      }
      for (uint32_t v : unseen) {
        cmps++;
        if (D.dist(q, v) <= radius_2) {
          if (result.size() == cap) { trunc = true; continue; }
          result.push_back(v);
        }
      }
    }
    for (size_t j = 0; j < result.size(); j++) out_ids[qi * cap + j] = result[j];
    out_counts[qi] = (uint32_t)result.size();
    if (out_cmps) out_cmps[qi] = (uint32_t)cmps;
    if (out_trunc) out_trunc[qi] = trunc ? 1u : 0u;
  });
  return 0;
}

// check_nn_recall.h:83-109 tie-aware recall: a returned id counts if it is among the first k
// ground-truth ids or any later ground-truth entry whose distance equals the k-th distance.
double pann_oracle_recall(const uint32_t* result_ids, uint32_t res_stride, const uint32_t* gt_ids,
                          const float* gt_dists, uint32_t gt_k, uint64_t nq, uint32_t k) {
  uint64_t hits = 0;
  for (uint64_t i = 0; i < nq; i++) {
    const uint32_t* g = gt_ids + i * gt_k; const float* gd = gt_dists + i * gt_k;
    // the k first ground-truth entries plus every later one at the k-th distance (:89-98);
    // each is counted once if it appears among the k reported ids (:99-106)
    for (uint32_t t = 0; t < gt_k; t++) {
      if (t >= k && !(gd[t] == gd[k - 1])) continue;
      for (uint32_t j = 0; j < k; j++)
        if (result_ids[i * res_stride + j] == g[t]) { hits++; break; }
    }
  }
  return (double)hits / (double)(nq * k);
}


// ---- scalar quantisation ----
// Euclidian_Point<uint8_t>::generate_parameters (euclidian_point.h:211-235); out2 = {slope, (float)offset}
void pann_oracle_euclid_u8_params(const float* x, uint64_t n, uint32_t d, float* out2) {
  float min_val = 0.0f, max_val = 0.0f;
  bool all_ints = true;
  for (uint64_t i = 0; i < n * d; i++) {
    float v = x[i];
    all_ints = all_ints && (v >= 0) && (v - (long)v) == 0;
    min_val = std::min(min_val, v); max_val = std::max(max_val, v);
  }
  if (all_ints) { if (max_val < 256) max_val = 255; min_val = 0; }
  const long range = 255;
  float slope = range / (max_val - min_val);                       // :106
  int32_t offset = (int32_t)std::round(min_val * slope);           // :107
  out2[0] = slope; out2[1] = (float)offset;
}
// Euclidian_Point<uint8_t>::translate_point (euclidian_point.h:182-209)
void pann_oracle_euclid_u8_translate(const float* x, uint64_t n, uint32_t d, float slope, int32_t offset, uint8_t* out) {
  const long range = 255;
  for (uint64_t i = 0; i < n * d; i++) {
    if (slope == 1.0 && offset == 0) { out[i] = (uint8_t)x[i]; continue; }
    int64_t r = (int64_t)(std::round(x[i] * slope)) - offset;
    if (r < 0) r = 0;
    if (r > range) r = range;
    out[i] = (uint8_t)r;
  }
}
// Mips_Point<float>::normalize (mips_point.h:113-122)
void pann_oracle_normalize(float* x, uint64_t n, uint32_t d) {
  for (uint64_t i = 0; i < n; i++) {
    float* v = x + i * d;
    double norm = 0.0;
    for (uint32_t j = 0; j < d; j++) norm += v[j] * v[j];
    norm = std::sqrt(norm);
    if (norm == 0) norm = 1.0;
    float inv_norm = 1.0 / norm;
    for (uint32_t j = 0; j < d; j++) v[j] = v[j] * inv_norm;
  }
}
// Quantized_Mips_Point<8,trim>::generate_parameters (mips_point.h:433-486) -> max_val
float pann_oracle_mips_i8_maxval(const float* x, uint64_t n, uint32_t d, int trim) {
  long len = (long)(n * d);
  std::vector<float> vals(x, x + len);
  std::sort(vals.begin(), vals.end());
  float min_val, max_val;
  if (trim) {
    float cutoff = .0001;
    min_val = vals[(long)(cutoff * len)];
    max_val = vals[(long)((1.0 - cutoff) * (len - 1))];
  } else { min_val = vals[0]; max_val = vals[len - 1]; }
  return std::max(max_val, -min_val);
}
// Quantized_Mips_Point<8>::translate_point (mips_point.h:416-430), range = 255
void pann_oracle_mips_i8_translate(const float* x, uint64_t n, uint32_t d, float mv, int8_t* out) {
  const int range = 255;
  for (uint64_t i = 0; i < n * d; i++) {
    float scale = (range / 2) / mv;
    float pj = x[i];
    if (pj < -mv) out[i] = (int8_t)(-range / 2);
    else if (pj > mv) out[i] = (int8_t)(range / 2);
    else { int32_t v = std::round(pj * scale); out[i] = (int8_t)v; }
  }
}


// ---- HCNNG (clusterEdge.h:40-153, hcnng_index.h:102-281) with this build's explicit seeding rules
// (DESIGN.md "Build determinism"): node RNG r: ith_rand(i) = mix(r + i), fork(i) = mix(mix(r) + i + 17),
// tree t seeded with mix(mix(seed + t)); a split that leaves one side empty falls back to "halve".
namespace {
inline uint64_t hc_mix(uint64_t x) {
  x += UINT64_C(0x9e3779b97f4a7c15);
  x = (x ^ (x >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  x = (x ^ (x >> 27)) * UINT64_C(0x94d049bb133111eb);
  return x ^ (x >> 31);
}
void hc_cluster(const Dataset& D, std::vector<uint32_t>& act, uint64_t rnd, size_t cluster_size,
                std::vector<std::vector<uint32_t>>& leaves) {
  if (act.size() <= cluster_size) { leaves.push_back(act); return; }          // clusterEdge.h:103-104
  const size_t fi = hc_mix(rnd + 0) % act.size();                              // select_two_random :40-50
  const size_t su = hc_mix(rnd + 1) % (act.size() - 1);
  const size_t si = su < fi ? su : su + 1;
  const uint32_t f = act[fi], s = act[si];
  std::vector<uint32_t> a, b;
  bool same = std::memcmp(D.row(f), D.row(s), (size_t)D.d * (D.dtype == DT_F32 ? 4 : (D.dtype == DT_F16 || D.dtype == DT_BF16) ? 2 : 1)) == 0;   // Points[f] == Points[s] :107
  if (!same) {
    for (uint32_t id : act) {                                                  // :71-83
      float df = D.dist_ids(id, f), ds = D.dist_ids(id, s);
      if (df <= ds) a.push_back(id); else b.push_back(id);
    }
  }
  if (same || a.empty() || b.empty()) {                                        // :108-115
    a.clear(); b.clear();
    for (size_t i = 0; i < act.size(); i++) (i < act.size() / 2 ? a : b).push_back(act[i]);
  }
  std::vector<uint32_t>().swap(act);
  hc_cluster(D, a, hc_mix(hc_mix(rnd) + 0 + 17), cluster_size, leaves);       // rnd.fork(0) :85
  hc_cluster(D, b, hc_mix(hc_mix(rnd) + 1 + 17), cluster_size, leaves);       // rnd.fork(1) :86
}
struct HcDS {   // hcnng_index.h:36-89 (incl. the use of rank[x] rather than rank[root])
  std::vector<int> parent, rank;
  explicit HcDS(size_t n) : parent(n), rank(n, 0) { for (size_t i = 0; i < n; i++) parent[i] = (int)i; }
  int find(int x) { if (parent[x] != x) parent[x] = find(parent[x]); return parent[x]; }
  void unite(int x, int y) {
    int xr = find(x), yr = find(y), xk = rank[x], yk = rank[y];
    if (xr == yr) return;
    if (xk < yk) parent[xr] = yr; else { parent[yr] = xr; if (xk == yk) rank[xr]++; }
  }
  bool full() { int r = find(0); for (size_t i = 1; i < parent.size(); i++) if (find((int)i) != r) return false; return true; }
};
}  // namespace

int pann_oracle_hcnng_build(const void* points, uint64_t n, uint32_t d, int dtype, uint64_t stride, int metric,
                            uint32_t* graph, uint32_t maxdeg, long num_clusters, long cluster_size, long mst_deg,
                            uint64_t seed, int nthreads) {
  Dataset D{(const uint8_t*)points, n, d, dtype, stride, metric, graph, maxdeg};
  const uint32_t m = 10;                                                       // hcnng_index.h:140
  for (long t = 0; t < num_clusters; t++) {
    std::vector<uint32_t> all(n);
    for (uint64_t i = 0; i < n; i++) all[i] = (uint32_t)i;
    std::vector<std::vector<uint32_t>> leaves;
    hc_cluster(D, all, hc_mix(hc_mix(seed + (uint64_t)t)), (size_t)cluster_size, leaves);
    parallel_for(0, leaves.size(), nthreads, [&](size_t li) {                 // MSTk :134-229
      const std::vector<uint32_t>& ids = leaves[li];
      const size_t N = ids.size();
      if (N < 2) return;
      struct E { float w; int i, j; };
      std::vector<E> edges;
      for (size_t i = 0; i < N; i++) {
        std::vector<IdDist> v;
        for (size_t j = 0; j < N; j++) if (j != i) v.push_back(IdDist{ids[j], D.dist_ids(ids[i], ids[j])});
        size_t kk = std::min<size_t>(m, v.size());
        std::partial_sort(v.begin(), v.begin() + kk, v.end(), less_id_dist);  // 10 nearest, ties by id
        for (size_t t2 = 0; t2 < kk; t2++) {
          int j = (int)(std::find(ids.begin(), ids.end(), v[t2].id) - ids.begin());
          edges.push_back(E{v[t2].dist, std::min((int)i, j), std::max((int)i, j)});
        }
      }
      auto lt = [](const E& a, const E& b) { return a.w < b.w || (a.w == b.w && (a.i < b.i || (a.i == b.i && a.j < b.j))); };
      std::sort(edges.begin(), edges.end(), lt);                               // less_dup :183-201
      edges.erase(std::unique(edges.begin(), edges.end(), [](const E& a, const E& b) { return a.w == b.w && a.i == b.i && a.j == b.j; }),
                  edges.end());
      HcDS ds(N);
      std::vector<int> deg(N, 0);
      for (size_t e = 0; e < edges.size(); e++) {                              // :208-226
        const int a = edges[e].i, b = edges[e].j;
        if (ds.find(a) != ds.find(b) && deg[a] < mst_deg && deg[b] < mst_deg) {
          for (int dir = 0; dir < 2; dir++) {                                  // process_edges :117-131
            uint32_t* row = graph + (uint64_t)ids[dir ? b : a] * (maxdeg + 1);
            if (row[0] < maxdeg) { row[1 + row[0]] = ids[dir ? a : b]; row[0]++; }
          }
          deg[a]++; deg[b]++;
          ds.unite(a, b);
        }
        if (e % N == 0 && ds.full()) break;
      }
    });
  }
  return 0;
}

}  // extern "C"
