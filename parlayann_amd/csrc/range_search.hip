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
#include <vector>

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
  // work list: item i is query qlist[i] (null: query i); waves take items from *work_counter.  A query whose `seen` set
  // would fill more than half of the wave's table is appended to ovf_list (when given) and left for a pass with larger tables.
  const uint32_t* qlist; uint64_t nlist; uint32_t* work_counter;
  uint32_t* ovf_list; uint32_t* ovf_count;
};

__device__ __forceinline__ unsigned long long rs_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
}

// is `id` in this query's set?  If not, *slot / *stale are where the probe ended: the first slot of the chain that does not
// belong to this query, and the (stale) entry found there -- rs_insert starts from them without reading the slot again.
// (h, e): the id's home slot and the entry already read from it -- the caller may have requested that read earlier
__device__ __forceinline__ bool rs_contains_from(const unsigned long long* T, uint64_t mask, uint64_t tag, uint32_t id, uint64_t h,
                                                 unsigned long long e, uint64_t* slot, unsigned long long* stale) {
  const unsigned long long want = (tag << 32) | id;
  for (;;) {
    if (e == want) return true;
    if ((e >> 32) != tag) { *slot = h; *stale = e; return false; }
    h = (h + 1) & mask;
    e = rs_load(T + h);
  }
}
__device__ __forceinline__ bool rs_contains(const unsigned long long* T, uint64_t mask, uint64_t tag, uint32_t id,
                                            uint64_t* slot, unsigned long long* stale) {
  const uint64_t h = hash64_2(id) & mask;
  return rs_contains_from(T, mask, tag, id, h, rs_load(T + h), slot, stale);
}

// insert `id` (known to be absent; concurrent lanes of the wave insert distinct ids), starting at the slot rs_contains ended on
__device__ __forceinline__ void rs_insert(unsigned long long* T, uint64_t mask, uint64_t tag, uint32_t id, uint64_t h,
                                          unsigned long long e) {
  const unsigned long long mine = (tag << 32) | id;
  for (;;) {
    if ((e >> 32) != tag) {
      if (__hip_atomic_compare_exchange_strong(T + h, &e, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE))
        return;
      if ((e >> 32) != tag) continue;      // the slot changed under us but is still free: try it again
    }
    h = (h + 1) & mask;                    // another lane of this query took it: next slot
    e = rs_load(T + h);
  }
}

// true on lanes whose (valid) id also sits on a lower valid lane.  Every valid lane writes its number into a byte table slot
// chosen by its id; a lane that reads back another lane's number shares the slot with somebody -- only those slot groups
// (about two per 64 ids at 1024 slots) are walked member by member.  W: 1024 bytes of LDS, contents irrelevant on entry.
__device__ __forceinline__ bool rs_dup_of_lower_lane(uint32_t id, bool valid, int lane, uint8_t* W) {
  const uint32_t s = (id * 0x9E3779B1u) >> 22;              // 10 bits
  if (valid) W[s] = (uint8_t)lane;
  wave_lds_sync();
  const uint32_t w = valid ? (uint32_t)W[s] : (uint32_t)lane;
  uint64_t losers = __ballot(valid && w != (uint32_t)lane);
  bool dup = false;
  while (losers) {
    const int L = __ffsll((unsigned long long)losers) - 1;
    const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)s, L);
    uint64_t grp = __ballot(valid && s == sL);              // every lane of that slot, winners included
    losers &= ~grp;
    const bool mine = valid && s == sL;
    while (grp) {                                           // ascending: a member is a duplicate of any EARLIER member with its id
      const int M = __ffsll((unsigned long long)grp) - 1;
      grp &= grp - 1;
      const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)id, M);
      dup |= mine && lane > M && o == id;
    }
  }
  wave_lds_sync();                                          // W may be rewritten by the next call
  return dup;
}

// (One wave per workgroup: every barrier below only orders LDS traffic -- wave_lds_sync, which unlike __syncthreads() does not
// wait for the result stores and table updates still in flight.)
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) range_search_kernel(RSArgs A) {
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  __shared__ float Dl[PANN_WAVE];
  // the youngest RS_RING entries of `result` also sit here: the BFS pops what this wave appended a moment ago, and reading it
  // back from HBM would put one more memory round trip on the per-vertex chain (entries older than the ring come from HBM)
  constexpr uint32_t RS_RING = 512;
  __shared__ uint32_t Rq[RS_RING];
  __shared__ uint8_t Wd[1024];                              // scratch of rs_dup_of_lower_lane
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  unsigned long long* T = A.table + (uint64_t)blockIdx.x * A.hsize;
  const uint64_t mask = A.hsize - 1;

  const uint32_t ins_limit = (uint32_t)min((uint64_t)0xFFFFFFFFu, A.hsize / 2);
  uint32_t item = 0;
  if (lane == 0) item = atomicAdd(A.work_counter, 1u);
  item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
  while (item < A.nlist) {
    const uint64_t qi = A.qlist ? A.qlist[item] : item;
    const uint64_t tag = qi + 1;                                        // table entries of other queries are "empty"
    uint32_t ins = 0;                                                   // ids this query has put into the table
    bool overflow = false;
    const uint32_t self = A.q_ids ? A.q_ids[qi] : SENTINEL;             // same_as(p): only a base-point query has a vertex
    QReg<DT> qreg{};
    wave_lds_sync();
    if (A.q_ids) load_query<DT, LPC, NCH1>(A.pv.points + (uint64_t)self * A.pv.pstride, A.pv.pstride, A.pv.nch, qreg, qlds, lane);
    else load_query<DT, LPC, NCH1>(A.q_ext + qi * A.q_stride, A.dbytes, A.pv.nch, qreg, qlds, lane);
    wave_lds_sync();
    uint32_t* res = A.out_ids + qi * A.cap;
    uint32_t count = 0, cmps = 0;
    bool trunc = false;

    // distances of the m ids in Pl -> Dl, then append those within the radius (in Pl order)
    auto score_and_append = [&](uint32_t m) {
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, m, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) Dl[ci] = dist; });
      wave_lds_sync();
      const bool in = lane < (int)m && Dl[lane] <= A.radius_2;
      const uint64_t im = __ballot(in);
      const uint32_t pos = count + lanes_below(im, lane);
      if (in && pos < A.cap) { __hip_atomic_store(res + pos, Pl[lane], __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE); Rq[pos % RS_RING] = Pl[lane]; }
      const uint32_t add = (uint32_t)__popcll(im);
      if (count + add > A.cap) trunc = true;
      count = min(count + add, A.cap);
    };
    auto result_at = [&](uint32_t pos) -> uint32_t {        // pos < count (wave-uniform)
      if (pos + RS_RING >= count) return Rq[pos % RS_RING];
      return __hip_atomic_load(res + pos, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
    };

    // ---- starts (:271-277) ----
    const uint32_t* sp = A.starts + (A.starts_per_query ? qi * A.nstarts : 0);
    for (uint32_t s0 = 0; s0 < A.nstarts && !trunc; s0 += PANN_WAVE) {
      const uint32_t v = (s0 + lane < A.nstarts) ? sp[s0 + lane] : SENTINEL;
      uint64_t slot = 0; unsigned long long stale = 0;
      bool live = v != SENTINEL && v != self && !rs_contains(T, mask, tag, v, &slot, &stale);
      // a start is skipped only if an EARLIER equal start passed the test: score all live ones, then drop
      // later duplicates of passing lanes (their comparison is not counted either)
      const uint64_t lm = __ballot(live);
      const uint32_t m = (uint32_t)__popcll(lm);
      if (m == 0) continue;
      if (live) Pl[lanes_below(lm, lane)] = v;
      wave_lds_sync();
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, m, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) Dl[ci] = dist; });
      wave_lds_sync();
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
      if (in && pos < A.cap) { __hip_atomic_store(res + pos, id, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE); Rq[pos % RS_RING] = id; }
      const uint32_t add = (uint32_t)__popcll(im);
      if (A.ovf_list && ins + add > ins_limit) { overflow = true; break; }
      if (in) {            // (the passing start sits on another lane than the probe did: probe again from its home slot)
        uint64_t h2 = 0; unsigned long long e2 = 0;
        (void)rs_contains(T, mask, tag, id, &h2, &e2);
        rs_insert(T, mask, tag, id, h2, e2);
      }
      ins += add;
      if (count + add > A.cap) trunc = true;
      count = min(count + add, A.cap);
      wave_lds_sync();
    }

    // ---- BFS (:280-297) ----
    // The first 64 neighbours of the vertex after the current one are requested one iteration ahead whenever that vertex is
    // already in the result (nearly always: the queue is longer than one), which takes the row fetch off the per-vertex chain.
    // ... and the first probe of those neighbours is requested while the current vertex's candidate rows are in flight (the
    // table does not change between that request and its use: inserts happen before the candidates are scored), which leaves two
    // dependent round trips per vertex -- max(probe, candidate rows) and the compare-and-swap -- instead of three.
    uint32_t position = 0;
    uint32_t pre_row = SENTINEL; bool pre_ok = false;
    bool pp_ok = false; uint64_t pp_h = 0; unsigned long long pp_e = 0;
    while (position < count && !trunc && !overflow) {
      const uint32_t next = result_at(position);
      position++;
      const uint32_t* row = A.graph + (uint64_t)next * A.gstride;
      const uint32_t first = pre_ok ? pre_row : ((uint32_t)lane < A.gstride ? row[lane] : SENTINEL);
      const bool have_pp = pp_ok; const uint64_t my_h = pp_h; const unsigned long long my_e = pp_e;
      pp_ok = false;
      pre_ok = position < count;                            // the next vertex is known now (entries below count are final)
      if (pre_ok) {
        const uint32_t nn = result_at(position);
        pre_row = (uint32_t)lane < A.gstride ? A.graph[(uint64_t)nn * A.gstride + lane] : SENTINEL;
      }
      for (uint32_t j0 = 0; j0 < A.gstride && !trunc; j0 += PANN_WAVE) {
        const uint32_t v = j0 == 0 ? first : ((j0 + lane < A.gstride) ? row[j0 + lane] : SENTINEL);
        const bool valid = v != SENTINEL;
        if (__ballot(valid) == 0) break;                    // neighbours are packed at the front of the row
        uint64_t slot = 0; unsigned long long stale = 0;
        bool unseen = valid && v != self;
        if (unseen) unseen = (j0 == 0 && have_pp) ? !rs_contains_from(T, mask, tag, v, my_h, my_e, &slot, &stale)
                                                  : !rs_contains(T, mask, tag, v, &slot, &stale);
        const bool dup = rs_dup_of_lower_lane(v, unseen, lane, Wd);       // (all lanes call it: it synchronises on LDS)
        unseen = unseen && !dup;
        const uint64_t um = __ballot(unseen);
        const uint32_t m = (uint32_t)__popcll(um);
        if (m == 0) continue;
        if (A.ovf_list && ins + m > ins_limit) { overflow = true; break; }
        if (unseen) { Pl[lanes_below(um, lane)] = v; rs_insert(T, mask, tag, v, slot, stale); }
        ins += m;
        wave_lds_sync();
        cmps += m;
        if (pre_ok && A.gstride <= PANN_WAVE) {             // one chunk per row: nothing more is inserted before the next vertex
          pp_h = hash64_2(pre_row) & mask;
          pp_e = (pre_row != SENTINEL && pre_row != self) ? rs_load(T + pp_h) : 0ull;
          pp_ok = true;
        }
        score_and_append(m);
        wave_lds_sync();
      }
    }
    if (lane == 0) {
      if (overflow) {                                      // this query goes to the pass with worst-case tables
        A.ovf_list[atomicAdd(A.ovf_count, 1u)] = (uint32_t)qi;
      } else {
        A.out_counts[qi] = count;
        if (A.out_cmps) A.out_cmps[qi] = cmps;
        if (A.out_trunc) A.out_trunc[qi] = trunc ? 1u : 0u;
      }
    }
    wave_lds_sync();
    item = 0;
    if (lane == 0) item = atomicAdd(A.work_counter, 1u);
    item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
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
  // Tables sized for the worst case (every result vertex contributes a full row of new ids) are megabytes per wave and leave room
  // for few waves; almost every query needs a small fraction of that.  Pass 1 runs all queries on many waves with small tables
  // and sets aside the queries that would fill more than half of one; pass 2 runs those with worst-case tables.
  const uint64_t hs_worst = range_search_table_entries(ix, nstarts, cap);
  const uint64_t budget = 4ull << 30;                                   // bytes of seen-set tables per launch
  const uint64_t waves1 = std::min<uint64_t>(nq, 8192);                 // 32 waves per CU: the kernel is a chain of dependent round trips
  uint64_t hs_small = 1024;                                             // the largest table that lets all of them run (>= 16384 entries)
  while (hs_small < hs_worst && (hs_small * 2) * 8 * waves1 <= budget) hs_small <<= 1;
  hs_small = std::min(hs_worst, std::max<uint64_t>(hs_small, 16384));
  const size_t head = 256 + ((size_t)nq * 4 + 255) / 256 * 256;         // counters + overflow list
  if (int rc = ws.ensure(head + waves1 * hs_small * 8)) return rc;
  RSArgs A{};
  A.pv = PointsView{ix.points, ix.pstride, ix.nch, ix.exact}; A.dbytes = ix.dbytes;
  A.graph = ix.graph; A.gstride = ix.gstride; A.n = ix.n;
  A.q_ext = d_q; A.q_stride = q_stride; A.q_ids = d_qids; A.nq = nq;
  A.starts = d_starts; A.nstarts = nstarts; A.starts_per_query = starts_per_query;
  A.radius_2 = radius_2; A.cap = cap;
  A.out_ids = d_out_ids; A.out_counts = d_out_counts; A.out_cmps = d_out_cmps; A.out_trunc = d_out_trunc;
  const size_t lds = query_lds_bytes(ix);
  auto launch = [&](uint64_t waves, uint64_t hsize, const uint32_t* qlist, uint64_t nlist, bool may_overflow) -> int {
    uint8_t* w = static_cast<uint8_t*>(ws.buf);
    A.work_counter = reinterpret_cast<uint32_t*>(w); A.ovf_count = reinterpret_cast<uint32_t*>(w + 64);
    A.ovf_list = may_overflow ? reinterpret_cast<uint32_t*>(w + 256) : nullptr;
    A.qlist = qlist; A.nlist = nlist;
    A.table = reinterpret_cast<unsigned long long*>(w + head); A.hsize = hsize;
    PANN_HIP(hipMemsetAsync(w, 0, 256, st));
    PANN_HIP(hipMemsetAsync(w + head, 0, waves * hsize * 8, st));       // tag 0 = never used
#define CALL_RS(DT, MT, L, N1) hipLaunchKernelGGL((range_search_kernel<DT, MT, L, N1>), dim3((uint32_t)waves), dim3(PANN_WAVE), lds, st, A)
    PANN_TYPE_SWITCH(ix, CALL_RS);
#undef CALL_RS
    PANN_HIP(hipGetLastError());
    return PANN_OK;
  };
  if (int rc = launch(waves1, hs_small, nullptr, nq, hs_small < hs_worst)) return rc;
  if (hs_small < hs_worst) {
    uint32_t novf = 0;
    PANN_HIP(hipMemcpyAsync(&novf, static_cast<uint8_t*>(ws.buf) + 64, 4, hipMemcpyDeviceToHost, st));
    PANN_HIP(hipStreamSynchronize(st));
    if (novf) {
      const uint64_t waves2 = std::min<uint64_t>(novf, std::max<uint64_t>(1, budget / (hs_worst * 8)));
      // the overflow list sits in the head of the workspace: growing the workspace would move it, so copy it out first
      std::vector<uint32_t> h_ovf(novf);
      PANN_HIP(hipMemcpy(h_ovf.data(), static_cast<uint8_t*>(ws.buf) + 256, (size_t)novf * 4, hipMemcpyDeviceToHost));
      if (int rc = ws.ensure(head + waves2 * hs_worst * 8)) return rc;
      PANN_HIP(hipMemcpyAsync(static_cast<uint8_t*>(ws.buf) + 256, h_ovf.data(), (size_t)novf * 4, hipMemcpyHostToDevice, st));
      PANN_HIP(hipStreamSynchronize(st));                                 // h_ovf goes out of scope
      if (int rc = launch(waves2, hs_worst, reinterpret_cast<const uint32_t*>(static_cast<uint8_t*>(ws.buf) + 256), novf, false)) return rc;
    }
  }
  return PANN_OK;
}

}  // namespace pann
