// range_search.hip -- the BFS range search of algorithms/utils/beamSearch.h:245-306 on the device.
//
// Reference behaviour (one query p, a list of start vertices, radius_2):
//   * starts (:271-277): skip a start that is already in `seen` or is the query's own vertex; otherwise count
//     one distance comparison; if dist <= radius_2 append it to `result` and insert it into `seen`
//     (a start that fails the test is NOT remembered, so a repeated failing start is counted again);
//   * BFS (:280-297): pop result[position++]; every neighbour (row order) that is not in `seen` and is not the
//     query's own vertex goes to `unseen` and into `seen`; then every unseen vertex costs one distance
//     comparison and is appended to `result` iff dist <= radius_2;
//   * returns (result in BFS order, distance_comparisons).  `seen` is an EXACT set (std::unordered_set).
//
// Device mapping: one wavefront per query (persistent over the batch).  `seen` is an exact open-addressing
// hash set in HBM, one region per wave, entries tagged with the query number so that the table is never
// cleared between queries; lookups and inserts are agent-scope atomics (all lanes of the wave work on the
// table concurrently; duplicates inside one 64-neighbour chunk are removed first, the lowest lane wins, so
// the order of `unseen` is the reference's first-occurrence order).  Distances use the same gather_tile as
// the beam search.  `result` is the caller's output row (capacity max_results): a query whose result would
// exceed it stops there and is reported as truncated.
#include <algorithm>

#include "pann_device.h"
#include "pann_internal.h"

namespace pann {

struct RSArgs {
  PointsView pv; uint32_t dbytes;
  const uint32_t* graph; uint32_t gstride; uint64_t n;
  const uint8_t* q_ext; uint64_t q_stride; const uint32_t* q_ids; uint64_t nq;
  const uint32_t* starts; uint32_t nstarts; int starts_per_query;
  float radius_2; uint32_t cap;
  unsigned long long* table; uint64_t hsize;      // per wave: hsize entries (power of two)
  uint32_t* out_ids; uint32_t* out_counts; uint32_t* out_cmps; uint32_t* out_trunc;
};

__device__ __forceinline__ unsigned long long rs_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// is `id` in this query's set?
__device__ __forceinline__ bool rs_contains(const unsigned long long* T, uint64_t mask, uint64_t tag, uint32_t id) {
  uint64_t h = hash64_2(id) & mask;
  const unsigned long long want = (tag << 32) | id;
  for (;;) {
    const unsigned long long e = rs_load(T + h);
    if (e == want) return true;
    if ((e >> 32) != tag) return false;
    h = (h + 1) & mask;
  }
}

// insert `id` (known to be absent; concurrent lanes insert distinct ids)
__device__ __forceinline__ void rs_insert(unsigned long long* T, uint64_t mask, uint64_t tag, uint32_t id) {
  uint64_t h = hash64_2(id) & mask;
  const unsigned long long mine = (tag << 32) | id;
  for (;;) {
    unsigned long long e = rs_load(T + h);
    if ((e >> 32) != tag) {
      if (__hip_atomic_compare_exchange_strong(T + h, &e, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return;
      if ((e >> 32) != tag) continue;      // lost to a stale-slot race that still left it free: retry this slot
    }
    h = (h + 1) & mask;
  }
}

// true on lanes whose (valid) id also sits on a lower valid lane
__device__ __forceinline__ bool rs_dup_of_lower_lane(uint32_t id, bool valid, int lane) {
  bool dup = false;
  const uint64_t vm = __ballot(valid);
  for (int l = 0; l < PANN_WAVE - 1; l++) {
    if (!((vm >> l) & 1)) continue;                       // wave-uniform
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)id, l);
    dup |= valid && lane > l && o == id;
  }
  return dup;
}

template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) range_search_kernel(RSArgs A) {
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  __shared__ float Dl[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  unsigned long long* T = A.table + (uint64_t)blockIdx.x * A.hsize;
  const uint64_t mask = A.hsize - 1;

  for (uint64_t qi = blockIdx.x; qi < A.nq; qi += gridDim.x) {
    const uint64_t tag = qi + 1;                                        // table entries of other queries are "empty"
    const uint32_t self = A.q_ids ? A.q_ids[qi] : SENTINEL;             // same_as(p): only a base-point query has a vertex
    QReg<DT> qreg{};
    __syncthreads();
    if (A.q_ids) load_query<DT, LPC, NCH1>(A.pv.points + (uint64_t)self * A.pv.pstride, A.pv.pstride, A.pv.nch, qreg, qlds, lane);
    else load_query<DT, LPC, NCH1>(A.q_ext + qi * A.q_stride, A.dbytes, A.pv.nch, qreg, qlds, lane);
    __syncthreads();
    uint32_t* res = A.out_ids + qi * A.cap;
    uint32_t count = 0, cmps = 0;
    bool trunc = false;

    // distances of the m ids in Pl -> Dl, then append those within the radius (in Pl order)
    auto score_and_append = [&](uint32_t m) {
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, m, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) Dl[ci] = dist; });
      __syncthreads();
      const bool in = lane < (int)m && Dl[lane] <= A.radius_2;
      const uint64_t im = __ballot(in);
      const uint32_t pos = count + lanes_below(im, lane);
      if (in && pos < A.cap) __hip_atomic_store(res + pos, Pl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t add = (uint32_t)__popcll(im);
      if (count + add > A.cap) trunc = true;
      count = min(count + add, A.cap);
    };

    // ---- starts (:271-277) ----
    const uint32_t* sp = A.starts + (A.starts_per_query ? qi * A.nstarts : 0);
    for (uint32_t s0 = 0; s0 < A.nstarts && !trunc; s0 += PANN_WAVE) {
      const uint32_t v = (s0 + lane < A.nstarts) ? sp[s0 + lane] : SENTINEL;
      bool live = v != SENTINEL && v != self && !rs_contains(T, mask, tag, v);
      // a start is skipped only if an EARLIER equal start passed the test: score all live ones, then drop
      // later duplicates of passing lanes (their comparison is not counted either)
      const uint64_t lm = __ballot(live);
      const uint32_t m = (uint32_t)__popcll(lm);
      if (m == 0) continue;
      if (live) Pl[lanes_below(lm, lane)] = v;
      __syncthreads();
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, m, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) Dl[ci] = dist; });
      __syncthreads();
      const bool me = lane < (int)m;
      const uint32_t id = me ? Pl[lane] : SENTINEL;
      const bool pass = me && Dl[lane] <= A.radius_2;
      bool shadowed = false;                                // an earlier equal start passed
      {
        const uint64_t pm = __ballot(pass);
        for (int l = 0; l < PANN_WAVE - 1; l++) {
          if (!((pm >> l) & 1)) continue;
          const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)id, l);
          shadowed |= me && lane > l && o == id;
        }
      }
      cmps += (uint32_t)__popcll(__ballot(me && !shadowed));
      const bool in = pass && !shadowed;
      const uint64_t im = __ballot(in);
      const uint32_t pos = count + lanes_below(im, lane);
      if (in && pos < A.cap) __hip_atomic_store(res + pos, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (in) rs_insert(T, mask, tag, id);
      const uint32_t add = (uint32_t)__popcll(im);
      if (count + add > A.cap) trunc = true;
      count = min(count + add, A.cap);
      __syncthreads();
    }

    // ---- BFS (:280-297) ----
    uint32_t position = 0;
    while (position < count && !trunc) {
      const uint32_t next = __hip_atomic_load(res + position, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      position++;
      const uint32_t* row = A.graph + (uint64_t)next * A.gstride;
      for (uint32_t j0 = 0; j0 < A.gstride && !trunc; j0 += PANN_WAVE) {
        const uint32_t v = (j0 + lane < A.gstride) ? row[j0 + lane] : SENTINEL;
        const bool valid = v != SENTINEL;
        if (__ballot(valid) == 0) break;                    // neighbours are packed at the front of the row
        bool unseen = valid && v != self && !rs_contains(T, mask, tag, v);
        unseen = unseen && !rs_dup_of_lower_lane(v, unseen, lane);
        const uint64_t um = __ballot(unseen);
        const uint32_t m = (uint32_t)__popcll(um);
        if (m == 0) continue;
        if (unseen) { Pl[lanes_below(um, lane)] = v; rs_insert(T, mask, tag, v); }
        __syncthreads();
        cmps += m;
        score_and_append(m);
        __syncthreads();
      }
    }
    if (lane == 0) {
      A.out_counts[qi] = count;
      if (A.out_cmps) A.out_cmps[qi] = cmps;
      if (A.out_trunc) A.out_trunc[qi] = trunc ? 1u : 0u;
    }
  }
}

// hsize entries per wave hold every vertex a query can put into `seen`: the passing starts and all
// neighbours of at most `cap` result vertices, and never more than n distinct ids
uint64_t range_search_table_entries(const DeviceIndex& ix, uint32_t nstarts, uint32_t cap) {
  const uint64_t worst = std::min<uint64_t>(ix.n, (uint64_t)nstarts + (uint64_t)cap * ix.max_deg);
  uint64_t h = 1024;
  while (h < 2 * worst) h <<= 1;
  return h;
}

int range_search_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint8_t* d_q, uint64_t q_stride,
                     const uint32_t* d_qids, uint64_t nq, const uint32_t* d_starts, uint32_t nstarts,
                     int starts_per_query, float radius_2, uint32_t cap, uint32_t* d_out_ids, uint32_t* d_out_counts,
                     uint32_t* d_out_cmps, uint32_t* d_out_trunc) {
  if (nq == 0) return PANN_OK;
  if (!ix.graph) { set_error("pann_range_search: the index has no graph"); return PANN_ERR_BAD_ARG; }
  const uint64_t hsize = range_search_table_entries(ix, nstarts, cap);
  const uint64_t budget = 4ull << 30;                                   // bytes of seen-set tables per launch
  uint64_t waves = std::min<uint64_t>(std::min<uint64_t>(nq, 8192), std::max<uint64_t>(1, budget / (hsize * 8)));
  if (int rc = ws.ensure(waves * hsize * 8)) return rc;
  PANN_HIP(hipMemsetAsync(ws.buf, 0, waves * hsize * 8, st));           // tag 0 = never used
  RSArgs A{};
  A.pv = PointsView{ix.points, ix.pstride, ix.nch, ix.exact}; A.dbytes = ix.dbytes;
  A.graph = ix.graph; A.gstride = ix.gstride; A.n = ix.n;
  A.q_ext = d_q; A.q_stride = q_stride; A.q_ids = d_qids; A.nq = nq;
  A.starts = d_starts; A.nstarts = nstarts; A.starts_per_query = starts_per_query;
  A.radius_2 = radius_2; A.cap = cap;
  A.table = (unsigned long long*)ws.buf; A.hsize = hsize;
  A.out_ids = d_out_ids; A.out_counts = d_out_counts; A.out_cmps = d_out_cmps; A.out_trunc = d_out_trunc;
  const size_t lds = query_lds_bytes(ix);
#define CALL_RS(DT, MT, L, N1) hipLaunchKernelGGL((range_search_kernel<DT, MT, L, N1>), dim3((uint32_t)waves), dim3(PANN_WAVE), lds, st, A)
  PANN_TYPE_SWITCH(ix, CALL_RS);
#undef CALL_RS
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

}  // namespace pann
