// beam_search.hip -- batched greedy beam search on gfx950: ONE WAVEFRONT PER QUERY.
//
// Replaces the parallel_for over queries of searchAll/qsearchAll (beamSearch.h:374,556) and the
// per-insert searches of Vamana's batch_insert (vamana/index.h:247-259); each wave runs the exact
// state machine of filtered_beam_search (beamSearch.h:22-214, use_filtering == false):
//
//   frontier F[<=beam]  sorted (dist,id) keys in LDS, one "visited" flag per entry
//   filter   H[1<<bits] the reference's lossy direct-mapped table (LDS up to 16 KB, else HBM)
//   cands    C[...]     survivors of `dist < cutoff`, accumulated across skipped merges (:162-168)
//
// Per iteration: pick first unvisited frontier entry -> one aligned read of its adjacency row
// (lane i <- neighbour i) -> wave-parallel replay of the SEQUENTIAL filter update (:130-136) ->
// gather distances, LPC lanes x 16 B per candidate vector so every load instruction covers whole
// rows (coalesced 64..512 B segments) -> ballot-compaction of candidates -> rank-based merge.
//
// Equivalences used (proved in DESIGN.md "Kernel 1"):
//  * unvisited_frontier[offset] (:109) == first frontier entry whose flag is clear, because the
//    frontier does not change between merges and the skipped entries are exactly the flagged ones.
//  * set_difference(frontier, visited) (:203-208) == entries with flag clear, provided an entry
//    that re-enters the frontier after having been visited gets its flag back.  That can only
//    happen while the frontier is not full (cut-prune :190-195 dropped it); such entries are kept
//    in a small per-query "dropped" list and candidates are checked against it.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pann_device.h"

namespace pann {

// Diagnostic build only (make STAMPS=1 -> lib/libpann_stamps.so): s_memtime at the phase boundaries
// of an iteration, summed per query into a buffer nothing else reads (guide section 7, In-kernel
// stamps).  In the shipped library PANN_STAMP() expands to nothing.
#ifndef PANN_GU
#define PANN_GU 4   /* candidate groups in flight per lane in the main gather */
#endif
#ifndef PANN_GU_B128
/* beam 65..128: a whole adjacency row per memory round trip (16 groups in flight) was measured SLOWER than 4
   (C3 build search phase 1.50 s vs 1.19 s); rows of several chunks per lane halve the group count inside
   gather_tile, so they ask for 8 here (= 4 groups x 3 chunks in flight) */
#define PANN_GU_B128(LPC, NCH1) 4   /* round 3, with the filter-code table: 4 / 8 / 16 for multi-chunk rows = 0.720 / 0.743 / 0.885 s (C3 2M search phase) */
#endif
#ifndef PANN_MINWAVES
#define PANN_MINWAVES 1
#endif
#ifndef PANN_B128_PAIR
/* beam 65..128: two vertices per memory round trip while merges are skipped.  Exact (the whole GPU suite and the fuzz soak pass
   with it on) but OFF: measured on the C3-shaped 2M build, search phase 0.864 s without vs 0.91..0.93 s with it -- the kernel
   already moves 5.1 TB/s on the HBM side (81 % of the 6.3 TB/s random-row ceiling), so more bytes in flight buy nothing and the
   rows fetched for nothing when row 1 ends the skipping cost bandwidth.  It does pay when the table sits wholly in LDS and
   occupancy is low (PANN_B128_SPLIT=3: 1.03 -> 0.98 s), which is slower than the default split anyway.
   Round 3: with the filter-code table (CODES: the whole filter in 6 KB of LDS, no table traffic) it is ON for that variant:
   C3 2M search phase 0.743 -> 0.712 s (PANN_B128_PAIR_CODES). */
#define PANN_B128_PAIR 0
#endif
#ifndef PANN_B128_PAIR_CODES
#define PANN_B128_PAIR_CODES 1
#endif
#ifndef PANN_B64_PREFETCH
#define PANN_B64_PREFETCH 1   /* speculative adjacency-row fetch in the beam-64 kernel (0: A/B library builds) */
#endif
#ifndef PANN_B64_PAIR
/* beam <= 64: two vertices per memory round trip while merges are skipped (the b128 form above, on the register frontier of
   the beam-64 kernel): one query's chain of dependent round trips gets a third shorter, which is what the tail of a 10K-query
   launch consists of.  A/B in profiles/r03_b64_pair.txt. */
#define PANN_B64_PAIR 0
#endif
#ifndef PANN_MINWAVES_B64
#define PANN_MINWAVES_B64 7   /* at least 7 waves per SIMD (<= 72 VGPRs); the kernel uses 61 -> 8 waves, LDS caps a CU at 30 queries */
#endif
#ifdef PANN_STAMPS
#define PANN_STAMP(slot)                                                                        \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    stamp_sum[slot] += t_ - stamp_prev;                                                         \
    stamp_prev = t_;                                                                            \
  } while (0)
#else
#define PANN_STAMP(slot) do { } while (0)
#endif

// every kernel of this file runs ONE wavefront per workgroup
#ifdef PANN_FULL_BARRIERS
#define PANN_WSYNC() __syncthreads()
#else
#define PANN_WSYNC() wave_lds_sync()
#endif

struct BSParams {
  const uint8_t* points; uint32_t pstride; uint32_t dbytes; uint32_t nch; uint32_t exact;
  const uint32_t* graph; uint32_t gstride; uint32_t max_deg;
  const uint8_t* queries; uint64_t qstride; const uint32_t* query_ids;
  const uint32_t* starts; uint32_t nstarts; uint32_t starts_stride;   // starts_stride: 0 shared, nstarts per query
  uint32_t nq;
  uint32_t k, beam, limit, degree_limit; double cut;
  uint32_t skip_enabled;   // QP.limit >= 2*beam (:163)
  uint32_t cut_enabled;    // QP.k > 0 && is_metric() (:190)
  uint32_t bits;           // log2 of filter size (:52)
  uint32_t bcap, ccap;     // LDS capacities (multiples of 64)
  uint32_t* hash_global;   // [slots][1<<bits] when the filter does not fit LDS
  uint32_t prio_start;     // first block index that runs with raised issue priority (register-frontier beam-64 kernel)
  uint32_t hsplit;         // beam 65..128 in HBM mode: 0 whole table in HBM, 1 half, 2 three quarters of it in LDS
  uint32_t p24;            // ... whose LDS part holds planar 24-bit entries (n < 2^24 - 1)
  const uint32_t* order;   // launch slot -> query (or null: identity).  Results do not depend on it: every output is indexed by the query
  const uint16_t* gcode;   // filter-code table (filter_codes.hip): codes of the neighbours, slot-aligned with the graph rows
  const uint16_t* rank16;  // ... and of every id (start points)
  uint64_t* dropped; uint32_t dcap;  // [nq][dcap] visited entries that left a non-full frontier
  uint32_t* work_counter;  // persistent variant: next query to take
  uint32_t* status;        // [0] |= 1 on visited-list overflow, |= 2 on dropped-list overflow
  pann_search_out out;
  unsigned long long* stamps;   // diagnostic build: [nq][8] cycle sums
};

template <bool HASH_LDS>
__device__ __forceinline__ uint32_t hload(const uint32_t* H, uint32_t s) {
  if constexpr (HASH_LDS) return H[s];
  else return __hip_atomic_load(H + s, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
}
template <bool HASH_LDS>
__device__ __forceinline__ void hstore(uint32_t* H, uint32_t s, uint32_t v) {
  if constexpr (HASH_LDS) H[s] = v;
  else __hip_atomic_store(H + s, v, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
}
template <bool HASH_LDS>
__device__ __forceinline__ void hsync() {
  if constexpr (!HASH_LDS) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): global table ops retire in order
  PANN_WSYNC();
}

// first index in sorted A[0..n) with A[i] >= key
__device__ __forceinline__ uint32_t lower_bound_lds(const uint64_t* A, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (A[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Replay of `for a in row: has_been_seen(a)` (:54-59,:131-136) for up to 64 ids held one per lane.
// Sequential semantics: after lane j is processed table[s_j] == a_j whether or not it was a hit, so
// lane i is a hit iff the LAST earlier lane with the same slot holds the same id, or, when there is
// none, iff the table held a_i on entry.  Returns "seen" per lane and leaves the table as the
// sequential loop would.
// Lanes that share a slot are found by writing the lane number somewhere slot-indexed and reading it
// back: into the table itself when it lives in LDS; into a 1 KB LDS scratch T (index = slot mod 1024,
// false sharing only costs a loop trip) when the table lives in HBM, which then sees ONE dependent
// load (the entry values) and one fire-and-forget store per call instead of four round trips.
// Split mode (HBM branch, hb = 1 or 2): the slots whose low hb bits are all ones live in HBM (index s >> hb), the
// other half / three quarters in LDS right after T -- fewer single-word L2 requests per row for a smaller LDS
// footprint than the whole table.
// hb == 3: the WHOLE table in LDS as 24-bit planes (12 KB at beam 128) -- no table traffic to HBM at all
__device__ __forceinline__ uint32_t split_lds_index(uint32_t s, uint32_t hb) {
  return hb == 1 ? (s >> 1) : hb == 3 ? s : (s >> 2) * 3 + (s & 3);
}
// Planar 24-bit entries (p24 != 0: n < 2^24 - 1, the LDS part only): a low plane of uint16 followed by a high plane of uint8,
// 3 bytes per slot instead of 4 -- exact (an id IS 24 bits), and 2 KB less LDS per query at beam 65..128.
struct LdsPart { uint8_t* base; uint32_t nslots; uint32_t p24; };
__device__ __forceinline__ uint32_t lds_part_load(const LdsPart& L, uint32_t i) {
  if (L.p24) {
    const uint32_t lo = reinterpret_cast<const uint16_t*>(L.base)[i];
    const uint32_t hi = (L.base + 2 * L.nslots)[i];
    return lo | (hi << 16);                                              // empty slot = 0xFFFFFF, never equal to an id
  }
  return reinterpret_cast<const uint32_t*>(L.base)[i];
}
__device__ __forceinline__ void lds_part_store(const LdsPart& L, uint32_t i, uint32_t a) {
  if (L.p24) {
    reinterpret_cast<uint16_t*>(L.base)[i] = (uint16_t)a;
    (L.base + 2 * L.nslots)[i] = (uint8_t)(a >> 16);
  } else {
    reinterpret_cast<uint32_t*>(L.base)[i] = a;
  }
}
__device__ __forceinline__ void lds_part_clear(const LdsPart& L, int lane) {
  if (L.p24) {      // 3 * nslots bytes of 0xFF, nslots a multiple of 64
    uint32_t* w = reinterpret_cast<uint32_t*>(L.base);
    for (uint32_t i = lane; i < (3 * L.nslots) / 4; i += PANN_WAVE) w[i] = 0xFFFFFFFFu;
  } else {
    uint32_t* w = reinterpret_cast<uint32_t*>(L.base);
    for (uint32_t i = lane; i < L.nslots; i += PANN_WAVE) w[i] = SENTINEL;
  }
}

// What one filter_update call changed, per lane: enough to put the table back (a row that was scanned speculatively and
// turns out not to be the next vertex must leave no trace in the lossy table, whose state decides later hits and misses).
struct FilterUndo { uint32_t s, old; bool wrote; };
template <bool HASH_LDS>
__device__ __forceinline__ void filter_undo(uint32_t* H, const FilterUndo& u, uint32_t hb, const LdsPart& Lp) {
  if (u.wrote) {
    const uint32_t hm = (1u << hb) - 1u;
    if constexpr (HASH_LDS) H[u.s] = u.old;
    else if (hb != 3 && (u.s & hm) == hm) __hip_atomic_store(H + (u.s >> hb), u.old, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
    else lds_part_store(Lp, split_lds_index(u.s, hb), u.old);
  }
  PANN_WSYNC();
}

template <bool HASH_LDS>
__device__ __forceinline__ bool filter_update(uint32_t* H, uint32_t hmask, bool active, uint32_t a, int lane,
                                              uint8_t* T = nullptr, uint32_t hb = 0, LdsPart Lp = LdsPart{nullptr, 0, 0},
                                              FilterUndo* undo = nullptr) {
  const uint32_t s = (uint32_t)hash64_2((uint64_t)a) & hmask;
  uint32_t old, w;
  const uint32_t hm = (1u << hb) - 1u;
  const bool in_hbm = hb != 3 && (s & hm) == hm;                        // always true when hb == 0, never when hb == 3
  if constexpr (HASH_LDS) {
    old = active ? H[s] : 0u;
    PANN_WSYNC();
    if (active) H[s] = (uint32_t)lane;                   // some lane of each slot group wins
    PANN_WSYNC();
    w = active ? H[s] : (uint32_t)lane;
  } else {
    __builtin_amdgcn_s_waitcnt(0);                       // the previous call's table stores have been acknowledged
    old = 0u;
    if (active) old = in_hbm ? __hip_atomic_load(H + (s >> hb), __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE) : lds_part_load(Lp, split_lds_index(s, hb));
    const uint32_t t = s & 1023u;
    if (active) T[t] = (uint8_t)lane;
    PANN_WSYNC();
    w = active ? (uint32_t)T[t] : (uint32_t)lane;
  }
  uint64_t losers = __ballot(active && w != (uint32_t)lane);
  int prev = -1;       // last earlier lane with my slot
  bool last = true;    // no later lane with my slot
  while (losers) {     // one trip per slot shared by >1 lane (about 2 per 64 ids at 1024 slots)
    const int L = __ffsll((unsigned long long)losers) - 1;
    const uint32_t sL = __builtin_amdgcn_readlane(s, L);
    const uint64_t grp = __ballot(active && s == sL);
    losers &= ~grp;
    if (active && s == sL) {
      const uint64_t below = grp & ((1ull << lane) - 1ull);
      prev = below ? 63 - __clzll((unsigned long long)below) : -1;
      last = (lane == 63) ? true : ((grp >> (lane + 1)) == 0ull);
    }
  }
  const uint32_t a_prev = __shfl(a, prev < 0 ? lane : prev);
  const bool seen = active && (prev >= 0 ? (a_prev == a) : (old == a));
  if (undo) { undo->s = s; undo->old = old; undo->wrote = active && last; }   // `old` of a slot group: the value on entry
  if constexpr (HASH_LDS) {
    PANN_WSYNC();
    if (active && last) H[s] = a;
    PANN_WSYNC();
  } else {
    if (active && last) {
      if (in_hbm) __hip_atomic_store(H + (s >> hb), a, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
      else lds_part_store(Lp, split_lds_index(s, hb), a);
    }
    PANN_WSYNC();                                     // T (and the LDS part) may be touched by the next call
  }
  return seen;
}

// The same replay on a table of 12-bit CLASS CODES (filter_codes.hip): slot s remembers the code of the id written last, and
// "table[s] == a" is "table[s] == code(a)" because (slot, code) <-> id is a bijection.  4 096 slots take 6 KB of LDS: an 8-bit
// plane P8[s] and a 4-bit plane, eight slots per dword of P4.  Two lanes of one instruction may own different nibbles of one
// dword, so the nibble is written with ds_mskor_b32 (D = (D & ~mask) | data, atomic per lane).  Empty slot = 0xFFF (no code is).
// Lanes that share a slot are found through the 1 KB byte scratch T as in the HBM-table form above.
__device__ __forceinline__ void code_store(uint8_t* P8, uint32_t* P4, uint32_t s, uint32_t code) {
  const uint32_t sh = (s & 7u) * 4u;
  P8[s] = (uint8_t)code;
  const uint32_t addr = (uint32_t)(uintptr_t)(P4 + (s >> 3));            // LDS byte address (low 32 bits of the flat address)
  asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(addr), "v"(0xFu << sh), "v"((code >> 8) << sh) : "memory");
}
__device__ __forceinline__ void filter_undo_codes(uint8_t* P8, uint32_t* P4, const FilterUndo& u) {
  if (u.wrote) code_store(P8, P4, u.s, u.old);
  PANN_WSYNC();
}
__device__ __forceinline__ bool filter_update_codes(uint8_t* P8, uint32_t* P4, uint32_t hmask, bool active, uint32_t a, uint32_t code,
                                                    int lane, uint8_t* T, FilterUndo* undo = nullptr) {
  const uint32_t s = (uint32_t)hash64_2((uint64_t)a) & hmask;
  const uint32_t sh = (s & 7u) * 4u;
  uint32_t old = 0x1000u;
  if (active) old = (uint32_t)P8[s] | (((P4[s >> 3] >> sh) & 0xFu) << 8);
  const uint32_t t = s & 1023u;
  if (active) T[t] = (uint8_t)lane;
  PANN_WSYNC();
  const uint32_t w = active ? (uint32_t)T[t] : (uint32_t)lane;
  uint64_t losers = __ballot(active && w != (uint32_t)lane);
  int prev = -1;       // last earlier lane with my slot
  bool last = true;    // no later lane with my slot
  while (losers) {
    const int L = __ffsll((unsigned long long)losers) - 1;
    const uint32_t sL = __builtin_amdgcn_readlane(s, L);
    const uint64_t grp = __ballot(active && s == sL);
    losers &= ~grp;
    if (active && s == sL) {
      const uint64_t below = grp & ((1ull << lane) - 1ull);
      prev = below ? 63 - __clzll((unsigned long long)below) : -1;
      last = (lane == 63) ? true : ((grp >> (lane + 1)) == 0ull);
    }
  }
  const uint32_t a_prev = __shfl(a, prev < 0 ? lane : prev);
  const bool seen = active && (prev >= 0 ? (a_prev == a) : (old == code));
  if (undo) { undo->s = s; undo->old = old; undo->wrote = active && last; }   // `old` of a slot group: the code on entry
  if (active && last) code_store(P8, P4, s, code);
  PANN_WSYNC();                                          // T and the table may be touched by the next call
  return seen;
}

// Distances from the query to the m ids in Pl[0..m); survivors of `dist < cutoff` (:157) are
// appended to C in row order.
template <int DT, int METRIC, int LPC, bool NCH1, int U, bool BATCH_EMIT = true>
__device__ __forceinline__ uint32_t gather_distances(const BSParams& P, const QReg<DT>& qreg,
                                                     const uint4* qlds, const uint32_t* Pl, uint32_t m,
                                                     uint32_t cutoff_ord, uint64_t* C, uint32_t c, int lane) {
  const PointsView PV{P.points, P.pstride, P.nch, P.exact};
  if constexpr (BATCH_EMIT) {
    // one append per gather iteration (G*U candidates): +4 % on the beam-64 kernel at 10K queries
    gather_tile<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, lane,
      [&](bool has, uint32_t, uint32_t id, float dist, uint64_t before) {
        const uint32_t ord = f2ord(dist);
        const bool pass = has && (ord < cutoff_ord);
        const uint64_t pm = __ballot(pass);
        if (pass) C[c + (uint32_t)__popcll(pm & before)] = ((uint64_t)ord << 32) | id;
        c += __popcll(pm);
      });
  } else {
    // one append per candidate group (the beam-128 kernel measured 2.5 % slower with the batched form)
    gather_tile<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, lane,
      [&](bool has, uint32_t, uint32_t id, float dist) {
        const uint32_t ord = f2ord(dist);
        const bool pass = has && (ord < cutoff_ord);
        const uint64_t pm = __ballot(pass);
        if (pass) C[c + lanes_below(pm, lane)] = ((uint64_t)ord << 32) | id;
        c += __popcll(pm);
      });
  }
  return c;
}

// The same for the survivors of TWO rows listed one after the other in Pl (row 1: Pl[0..split), row 2: the rest):
// *c_split = size of C after the candidates of row 1 alone (the appends keep Pl order).
template <int DT, int METRIC, int LPC, bool NCH1, int U, bool BATCH_EMIT = false>
__device__ __forceinline__ uint32_t gather_distances_split(const BSParams& P, const QReg<DT>& qreg, const uint4* qlds,
                                                           const uint32_t* Pl, uint32_t m, uint32_t split, uint32_t cutoff_ord,
                                                           uint64_t* C, uint32_t c, int lane, uint32_t* c_split) {
  const PointsView PV{P.points, P.pstride, P.nch, P.exact};
  uint32_t c1 = c;
  if constexpr (BATCH_EMIT) {
    gather_tile<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, lane,
      [&](bool has, uint32_t ci, uint32_t id, float dist, uint64_t before) {
        const uint32_t ord = f2ord(dist);
        const bool pass = has && (ord < cutoff_ord);
        const uint64_t pm = __ballot(pass);
        if (pass) C[c + (uint32_t)__popcll(pm & before)] = ((uint64_t)ord << 32) | id;
        c += __popcll(pm);
        c1 += __popcll(__ballot(pass && ci < split));
      });
  } else {
    gather_tile<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, lane,
      [&](bool has, uint32_t ci, uint32_t id, float dist) {
        const uint32_t ord = f2ord(dist);
        const bool pass = has && (ord < cutoff_ord);
        const uint64_t pm = __ballot(pass);
        if (pass) C[c + lanes_below(pm, lane)] = ((uint64_t)ord << 32) | id;
        c += __popcll(pm);
        c1 += __popcll(__ballot(pass && ci < split));
      });
  }
  *c_split = c1;
  return c;
}


// Generic kernel: any beam (frontier in LDS), filter in LDS or in HBM scratch.
template <int DT, int METRIC, int LPC, bool NCH1, bool HASH_LDS>
__global__ void __launch_bounds__(PANN_WAVE, PANN_MINWAVES) beam_search_kernel(BSParams P) {
  const int lane = threadIdx.x;
  extern __shared__ __align__(16) uint8_t smem[];
  // ---- LDS carve (all regions 16 B aligned): 6.4 KB at beam 64 / degree 64 -> 24 queries per CU ----
  uint64_t* F = reinterpret_cast<uint64_t*>(smem);       // [bcap] frontier keys
  uint64_t* NF = F + P.bcap;                             // [bcap] merge output (swapped with F); between
                                                         //        merges its first 256 B hold Pl
  uint64_t* C = NF + P.bcap;                             // [ccap] candidates (unsorted)
  uint8_t* Fv = reinterpret_cast<uint8_t*>(C + P.ccap);  // [bcap] 0 unvisited, 1 visited, 2 visited+listed
  uint8_t* NFv = Fv + P.bcap;                            // [bcap]
  uint16_t* CP = reinterpret_cast<uint16_t*>(NFv + P.bcap);  // [ccap] candidate's lower_bound in F
  uint4* qlds = reinterpret_cast<uint4*>(CP + P.ccap);   // [nch*LPC] query (generic variant)
  uint32_t* Hl = reinterpret_cast<uint32_t*>(qlds + (NCH1 ? 0 : P.nch * LPC));  // [1<<bits] if HASH_LDS, else 1 KB replay scratch
#define PANN_PL (reinterpret_cast<uint32_t*>(NF))        /* [64] filter survivors of one row chunk */

  const uint32_t hsize = 1u << P.bits, hmask = hsize - 1u;
  const uint32_t beam = P.beam;
  const uint32_t BIG_ORD = f2ord(2147483648.0f);  // (distanceType) numeric_limits<int>::max()  (:152)

  uint32_t slot = blockIdx.x;
  if constexpr (!HASH_LDS) {  // persistent: one filter slot per block, queries pulled from a counter
    slot = 0;
    if (lane == 0) slot = atomicAdd(P.work_counter, 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
  }
  while (slot < P.nq) {
    const uint32_t qi = P.order ? P.order[slot] : slot;
    uint32_t* H = HASH_LDS ? Hl : P.hash_global + ((size_t)blockIdx.x << P.bits);
    // ---- filter init: hash_filter(1<<bits, -1) (:53) ----
    for (uint32_t i = lane; i < hsize; i += PANN_WAVE) hstore<HASH_LDS>(H, i, SENTINEL);

    // ---- query vector -> registers (one chunk) or LDS (generic) ----
    const int64_t self = P.query_ids ? (int64_t)P.query_ids[qi] : -1;
    const uint8_t* qrow = P.query_ids ? P.points + (uint64_t)self * P.pstride : P.queries + (uint64_t)qi * P.qstride;
    QReg<DT> qreg{};
    load_query<DT, LPC, NCH1>(qrow, P.dbytes, P.nch, qreg, qlds, lane);
    hsync<HASH_LDS>();

    uint32_t f = 0;        // frontier size
    uint32_t c = 0;        // accumulated candidates
    uint32_t nvis = 0;     // num_visited
    uint32_t dcmps = P.nstarts;  // dist_cmps == full_dist_cmps (:83-84)
    uint32_t degsum = 0;
    uint32_t ndrop = 0;    // entries in the dropped list
#ifdef PANN_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    uint64_t* DL = P.dropped + (size_t)qi * P.dcap;

    // ---- start points (:66-70): distance for each, filter insert in order, then "merge" into
    // the empty frontier (which sorts them) ----
    for (uint32_t s0 = 0; s0 < P.nstarts; s0 += PANN_WAVE) {
      const uint32_t i = s0 + lane;
      const bool act = i < P.nstarts;
      const uint32_t a = act ? P.starts[(size_t)qi * P.starts_stride + i] : 0u;
      (void)filter_update<HASH_LDS>(H, hmask, act, a, lane, reinterpret_cast<uint8_t*>(Hl));
      if (act) PANN_PL[lane] = a;
      PANN_WSYNC();
      const uint32_t m = min(P.nstarts - s0, (uint32_t)PANN_WAVE);
      // every start enters the frontier: cutoff above any finite distance
      c = gather_distances<DT, METRIC, LPC, NCH1, 4>(P, qreg, qlds, PANN_PL, m, 0xFFFFFFFFu, C, c, lane);
      PANN_WSYNC();
    }

    bool first = true;  // the start-point pseudo-merge: no cut-prune, no visit
    for (;;) {
      bool do_merge = first;
      if (!first) {
        // ---- next vertex: first unvisited frontier entry (:107-109) ----
        int cur_idx = -1;
        uint32_t unv_total = 0;
        for (uint32_t e0 = 0; e0 < f; e0 += PANN_WAVE) {
          const uint32_t e = e0 + lane;
          const uint64_t um = __ballot(e < f && Fv[e] == 0);
          if (cur_idx < 0 && um) cur_idx = (int)e0 + __ffsll((unsigned long long)um) - 1;
          unv_total += __popcll(um);
        }
        if (cur_idx < 0 || nvis >= P.limit) break;
        PANN_STAMP(0);   // head scan
        const uint64_t cur_key = F[cur_idx];
        const uint32_t cur = key_id(cur_key);
        // ---- visited.insert(current) (:112-114) ----
        if (lane == 0) {
          Fv[cur_idx] = 1;
          if (P.out.visited_cap) {
            if (nvis < P.out.visited_cap) {
              if (P.out.visited_ids) P.out.visited_ids[(size_t)qi * P.out.visited_cap + nvis] = cur;
              if (P.out.visited_dists) P.out.visited_dists[(size_t)qi * P.out.visited_cap + nvis] = key_dist(cur_key);
            } else {
              atomicOr(P.status, 1u);
            }
          }
        }
        nvis++;
        const bool more_unvisited = unv_total > 1;         // offset + 1 < remain (:165)
        const bool full = (f == beam);                      // :115
        uint32_t cutoff_ord = BIG_ORD;                      // :150-152
        if (full) cutoff_ord = (uint32_t)(F[f - 1] >> 32);

        // ---- adjacency row: lane i <- slot i; degree = number of non-sentinel slots ----
        const uint32_t* row = P.graph + (size_t)cur * P.gstride;
        for (uint32_t i0 = 0; i0 < P.gstride; i0 += PANN_WAVE) {
          const uint32_t i = i0 + lane;
          uint32_t a = SENTINEL;
          if (i < P.gstride) a = row[i];
          const bool act = (a != SENTINEL) && (i < P.degree_limit);   // min(size, degree_limit) (:130)
          const uint64_t am = __ballot(act);
          PANN_STAMP(1);   // adjacency row arrived
          if (am == 0ull) break;
          degsum += __popcll(am);
          const bool seen = filter_update<HASH_LDS>(H, hmask, act, a, lane, reinterpret_cast<uint8_t*>(Hl));
          const bool keep = act && !seen && ((int64_t)a != self);     // :133
          const uint64_t km = __ballot(keep);
          const uint32_t m = __popcll(km);
          if (keep) PANN_PL[lanes_below(km, lane)] = a;
          dcmps += m;                                                 // :137,155
          PANN_WSYNC();
          PANN_STAMP(2);   // filter + compaction
          if (m) c = gather_distances<DT, METRIC, LPC, NCH1, PANN_GU>(P, qreg, qlds, PANN_PL, m, cutoff_ord, C, c, lane);
          PANN_WSYNC();
          PANN_STAMP(3);   // gather + distances
        }
        // ---- skip the merge while too few candidates (:162-168) ----
        PANN_WSYNC();
        const bool skip = (c == 0) || (P.skip_enabled && c < beam / 8 && more_unvisited);
        do_merge = !skip;
      }
      if (do_merge) {
        {
          // ================= merge: sort+unique(C), set_union with F, trim (:173-185) =========
          // A1: kill duplicates (same id <=> same key) and entries already in F; remember rank in F
          for (uint32_t j0 = 0; j0 < c; j0 += PANN_WAVE) {
            const uint32_t j = j0 + lane;
            uint64_t key = KEY_INF;
            uint32_t p = 0;
            if (j < c) {
              key = C[j];
              bool dead = false;
              for (uint32_t i = 0; i < j; i++) dead |= (C[i] == key);
              p = lower_bound_lds(F, f, key);
              dead |= (p < f && F[p] == key);
              if (dead) key = KEY_INF;
            }
            PANN_WSYNC();       // all reads of C[0..j) by this chunk are done
            if (j < c) { C[j] = key; CP[j] = (uint16_t)p; }
            // later chunks compare against earlier ORIGINAL keys; a killed earlier key was itself a
            // duplicate of a still earlier live one (or of F), so the verdict is unchanged.
            PANN_WSYNC();
          }
          // A2: rank among live candidates -> direct placement into NF
          uint32_t nvalid = 0;
          for (uint32_t j0 = 0; j0 < c; j0 += PANN_WAVE) {
            const uint32_t j = j0 + lane;
            const uint64_t key = j < c ? C[j] : KEY_INF;
            nvalid += __popcll(__ballot(key != KEY_INF));
            if (key != KEY_INF) {
              uint32_t r = 0;
              for (uint32_t i = 0; i < c; i++) r += (C[i] < key) ? 1u : 0u;   // dead entries are KEY_INF
              const uint32_t pos = r + CP[j];
              if (pos < beam) {
                uint32_t flag = 0u;   // re-entry of an already visited vertex? (only while not full)
                for (uint32_t t = 0; t < ndrop; t++)
                  flag |= (__hip_atomic_load(DL + t, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE) == key) ? 2u : 0u;
                NF[pos] = key; NFv[pos] = (uint8_t)flag;
              }
            }
          }
          // B: old frontier entries move right by the number of live candidates below them
          for (uint32_t e0 = 0; e0 < f; e0 += PANN_WAVE) {
            const uint32_t e = e0 + lane;
            if (e < f) {
              const uint64_t key = F[e];
              uint32_t below = 0;
              for (uint32_t i = 0; i < c; i++) below += (C[i] < key) ? 1u : 0u;
              const uint32_t pos = e + below;
              if (pos < beam) { NF[pos] = key; NFv[pos] = Fv[e]; }
            }
          }
          PANN_WSYNC();
          const uint32_t f_old = f;
          uint32_t f_new = min(f_old + nvalid, beam);   // :185
          // ---- cut-prune (:190-195) ----
          if (!first && P.cut_enabled && f_new > P.k) {
            const float dk = key_dist(NF[P.k]);
            const float thr = (float)(P.cut * (double)dk);
            const uint64_t thr_key = (uint64_t)f2ord(thr) << 32;   // pair{0, thr}: kept iff key <= thr_key
            uint32_t ub = 0;
            for (uint32_t e0 = 0; e0 < f_new; e0 += PANN_WAVE) {
              const uint32_t e = e0 + lane;
              ub += __popcll(__ballot(e < f_new && NF[e] <= thr_key));
            }
            f_new = max(ub, f_old);
          }
          // ---- visited entries that fall off a frontier that is still not full can come back:
          // remember them (see file header).  Once full, nothing dropped can ever re-enter. ----
          if (f_new == beam) {
            ndrop = 0;
          } else if (P.cut_enabled && !first) {
            for (uint32_t e0 = 0; e0 < f_old; e0 += PANN_WAVE) {
              const uint32_t e = e0 + lane;
              bool lost = false;
              uint64_t key = 0;
              if (e < f_old && Fv[e] == 1) {
                key = F[e];
                uint32_t below = 0;
                for (uint32_t i = 0; i < c; i++) below += (C[i] < key) ? 1u : 0u;
                lost = (e + below) >= f_new;
              }
              const uint64_t lm = __ballot(lost);
              if (lost) {
                const uint32_t at = ndrop + lanes_below(lm, lane);
                if (at < P.dcap) DL[at] = key; else atomicOr(P.status, 2u);
              }
              ndrop = min(ndrop + (uint32_t)__popcll(lm), P.dcap);
            }
            // entries already listed (flag 2) stay listed; nothing to do for them
            __builtin_amdgcn_s_waitcnt(0);   // dropped-list stores visible to this wave's later loads
          }
          // visited entries with flag 2 that stay in the frontier keep flag 2 (== visited)
          { uint64_t* t = F; F = NF; NF = t; uint8_t* tv = Fv; Fv = NFv; NFv = tv; }
          f = f_new;
          c = 0;                      // candidates.clear() (:182)
          PANN_WSYNC();
        }
      }
      PANN_STAMP(4);     // merge (or nothing when skipped)
      first = false;
    }

    // ---- outputs (:211-213; searchAll takes the first k ids :378-380) ----
    const size_t qo = (size_t)qi * P.out.out_k;
    for (uint32_t j = lane; j < P.out.out_k; j += PANN_WAVE) {
      const bool ok = j < f;
      uint64_t key = 0ull;
      if (ok) key = F[j];
      if (P.out.ids) P.out.ids[qo + j] = ok ? key_id(key) : SENTINEL;
      if (P.out.dists) P.out.dists[qo + j] = ok ? key_dist(key) : __builtin_inff();
    }
#ifdef PANN_STAMPS
    if (lane == 0 && P.stamps) for (int i = 0; i < 8; i++) P.stamps[(size_t)qi * 8 + i] = stamp_sum[i];
#endif
    if (lane == 0) {
      if (P.out.frontier_size) P.out.frontier_size[qi] = f;
      if (P.out.visited_count) P.out.visited_count[qi] = nvis;
      if (P.out.dist_cmps) P.out.dist_cmps[qi] = dcmps;
      if (P.out.degree_sum) P.out.degree_sum[qi] = degsum;
    }
    PANN_WSYNC();
    if constexpr (HASH_LDS) break;
    else {
      slot = 0;
      if (lane == 0) slot = atomicAdd(P.work_counter, 1u);
      slot = __builtin_amdgcn_readfirstlane(slot);
    }
  }
}

// =============================================================================================
// beam <= 64 specialisation: the frontier lives in REGISTERS (lane e <-> entry e: key + visited
// flag).  Head scan = one ballot, cutoff = one readlane, every merge = loops of readlane/ballot
// steps over <= 64 candidates at a time (top-beam of a union is associative, so candidate chunks
// are merged one after the other; cut-prune runs once at the end).  LDS per query shrinks to the
// filter (4 KB) + a 64-entry scatter scratch + the candidate list: 5.2 KB -> 31 queries per CU.
// =============================================================================================
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE, PANN_MINWAVES_B64) beam_search_b64_kernel(BSParams P) {
  const int lane = threadIdx.x;
  extern __shared__ __align__(16) uint8_t smem[];
  uint64_t* S = reinterpret_cast<uint64_t*>(smem);          // [64] merge scatter scratch; first 256 B double as Pl
  uint64_t* C = S + 64;                                     // [ccap] candidates (unsorted)
  uint8_t* Sv = reinterpret_cast<uint8_t*>(C + P.ccap);     // [64] flags of the scatter scratch
  uint4* qlds = reinterpret_cast<uint4*>(Sv + 64);          // [nch*LPC] query (generic variant)
  uint32_t* H = reinterpret_cast<uint32_t*>(qlds + (NCH1 ? 0 : P.nch * LPC));  // [1<<bits]
  uint32_t* Pl = reinterpret_cast<uint32_t*>(S);            // [64] filter survivors of one row chunk

  const uint32_t hsize = 1u << P.bits, hmask = hsize - 1u;
  const uint32_t beam = P.beam;
  const uint32_t BIG_ORD = f2ord(2147483648.0f);            // (:152)
  const uint32_t qi = P.order ? P.order[blockIdx.x] : blockIdx.x;
  // Blocks are dispatched in index order, so the highest indices start last and form the tail of the launch: they
  // get instruction-issue priority over the queries that are already under way (measured +1..3 % at 10K queries,
  // inside the box-to-box noise but never negative).
  if (blockIdx.x >= P.prio_start) __builtin_amdgcn_s_setprio(3);

  for (uint32_t i = lane; i < hsize; i += PANN_WAVE) H[i] = SENTINEL;       // :53
  const int64_t self = P.query_ids ? (int64_t)P.query_ids[qi] : -1;
  const uint8_t* qrow = P.query_ids ? P.points + (uint64_t)self * P.pstride : P.queries + (uint64_t)qi * P.qstride;
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(qrow, P.dbytes, P.nch, qreg, qlds, lane);
  PANN_WSYNC();

  uint32_t f = 0, c = 0, nvis = 0, dcmps = P.nstarts, degsum = 0, ndrop = 0;
  uint64_t fkey = KEY_INF;
  uint32_t fflag = 0;
  uint64_t* DL = P.dropped + (size_t)qi * P.dcap;
#ifdef PANN_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif

  // ---- start points (:66-70); nstarts <= 64 here ----
  {
    const bool act = lane < (int)P.nstarts;
    const uint32_t a = act ? P.starts[(size_t)qi * P.starts_stride + lane] : 0u;
    (void)filter_update<true>(H, hmask, act, a, lane);
    if (act) Pl[lane] = a;
    PANN_WSYNC();
    c = gather_distances<DT, METRIC, LPC, NCH1, 4>(P, qreg, qlds, Pl, P.nstarts, 0xFFFFFFFFu, C, c, lane);
    PANN_WSYNC();
  }

  uint32_t pref_id = SENTINEL, pref_row = SENTINEL;            // speculative row fetch (see the loop)
  bool first = true;
  for (;;) {
    bool do_merge = first;
    if (!first) {
      // ---- next vertex: first unvisited frontier entry (:107-109) ----
      const uint64_t um = __ballot(lane < (int)f && fflag == 0);
      if (um == 0ull || nvis >= P.limit) break;
      const int cur_idx = __ffsll((unsigned long long)um) - 1;
      const bool more_unvisited = (um & (um - 1)) != 0ull;       // offset + 1 < remain (:165)
      PANN_STAMP(0);
      const uint64_t cur_key = readlane64(fkey, cur_idx);
      const uint32_t cur = key_id(cur_key);
      if (lane == cur_idx) fflag = 1;                            // visited.insert(current) (:112-114)
      if (lane == 0 && P.out.visited_cap) {
        if (nvis < P.out.visited_cap) {
          if (P.out.visited_ids) P.out.visited_ids[(size_t)qi * P.out.visited_cap + nvis] = cur;
          if (P.out.visited_dists) P.out.visited_dists[(size_t)qi * P.out.visited_cap + nvis] = key_dist(cur_key);
        } else {
          atomicOr(P.status, 1u);
        }
      }
      nvis++;
      uint32_t cutoff_ord = BIG_ORD;                             // :150-152
      if (f == beam) cutoff_ord = (uint32_t)(readlane64(fkey, (int)f - 1) >> 32);

      // Speculative adjacency-row fetch: the frontier's NEXT unvisited entry is the next vertex unless this iteration's
      // merge puts something in front of it (hit rate ~80 %), so its row is requested now, a whole iteration ahead -- one
      // dependent memory round trip less per iteration, which is what a query costs once the chip is no longer full (the
      // tail of a 10K-query launch).  The load is issued by hand (asm): hipcc drains vmcnt(0) at loop headers, which made a
      // compiler-visible prefetch wait at once (round 1: -2..-8 %).  An unknown older load only makes the compiler's
      // counted waits wait for more, never less; the register is drained (vmcnt(0), free by then) before it is reused.
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pref_row));
      const bool pref_hit = PANN_B64_PREFETCH && (pref_id == cur);
      const uint32_t pref_val = pref_row;
      pref_id = SENTINEL;
      const uint64_t rest1 = um & (um - 1);
      // two vertices this iteration? (nvis already counts `cur`: a second visit needs nvis < limit)
      const bool pair_try = PANN_B64_PAIR && P.skip_enabled && rest1 != 0ull && P.gstride <= PANN_WAVE && nvis < P.limit;
      if (PANN_B64_PREFETCH) {
        const uint64_t rest = pair_try ? (rest1 & (rest1 - 1)) : rest1;      // the entry after the one(s) visited now
        if (rest) {
          pref_id = key_id(readlane64(fkey, __ffsll((unsigned long long)rest) - 1));
          const uint32_t* np = P.graph + (size_t)pref_id * P.gstride + min((uint32_t)lane, P.gstride - 1);
          asm volatile("global_load_dword %0, %1, off" : "=v"(pref_row) : "v"(np));
        }
      }

      const uint32_t* row = P.graph + (size_t)cur * P.gstride;
      if (pair_try) {
        // While merges are skipped (:162-168) the frontier and the cutoff do not change, so the vertex after `cur` is known:
        // the second unvisited entry.  Both adjacency rows are fetched together, the filter is replayed row after row and ONE
        // gather fetches the survivors of both.  If the candidates of row 1 alone end the skipping (or row 2 does not fit the
        // candidate list), row 2 never happened: its candidates are cut off and its table writes are put back.
        const int idx2 = __ffsll((unsigned long long)rest1) - 1;
        const bool more_after2 = (rest1 & (rest1 - 1)) != 0ull;
        const uint64_t key2 = readlane64(fkey, idx2);
        const uint32_t cur2 = key_id(key2);
        uint32_t a1 = SENTINEL, a2 = SENTINEL;
        if (lane < (int)P.gstride) {
          a1 = pref_hit ? pref_val : row[lane];
          a2 = P.graph[(size_t)cur2 * P.gstride + lane];
        }
        const bool act1 = (a1 != SENTINEL) && ((uint32_t)lane < P.degree_limit);
        const bool act2 = (a2 != SENTINEL) && ((uint32_t)lane < P.degree_limit);
        PANN_STAMP(1);
        const bool seen1 = filter_update<true>(H, hmask, act1, a1, lane);
        FilterUndo u2;
        const bool seen2 = filter_update<true>(H, hmask, act2, a2, lane, nullptr, 0, LdsPart{nullptr, 0, 0}, &u2);
        const bool keep1 = act1 && !seen1 && ((int64_t)a1 != self);
        const bool keep2 = act2 && !seen2 && ((int64_t)a2 != self);
        const uint64_t km1 = __ballot(keep1), km2 = __ballot(keep2);
        const uint32_t m1 = __popcll(km1), m2 = __popcll(km2);
        const bool both = (c + m1 + m2 <= P.ccap);                   // the candidate list is sized for beam/8 - 1 + ONE row
        if (keep1) Pl[lanes_below(km1, lane)] = a1;
        if (both && keep2) Pl[m1 + lanes_below(km2, lane)] = a2;
        PANN_WSYNC();
        PANN_STAMP(2);
        uint32_t c_after1 = c;
        if (both) {
          if (m1 + m2) c = gather_distances_split<DT, METRIC, LPC, NCH1, PANN_GU, true>(P, qreg, qlds, Pl, m1 + m2, m1, cutoff_ord, C, c, lane, &c_after1);
        } else {
          if (m1) c = gather_distances<DT, METRIC, LPC, NCH1, PANN_GU>(P, qreg, qlds, Pl, m1, cutoff_ord, C, c, lane);
          c_after1 = c;
        }
        PANN_WSYNC();
        PANN_STAMP(3);
        // the reference's decision after vertex 1 (another unvisited entry exists): skip iff its list is still short
        const bool two = both && (c_after1 == 0 || c_after1 < beam / 8);
        if (!two) { c = c_after1; filter_undo<true>(H, u2, 0, LdsPart{nullptr, 0, 0}); }
        if (two) {
          if (lane == idx2) fflag = 1;
          if (lane == 0 && P.out.visited_cap) {
            if (nvis < P.out.visited_cap) {
              if (P.out.visited_ids) P.out.visited_ids[(size_t)qi * P.out.visited_cap + nvis] = cur2;
              if (P.out.visited_dists) P.out.visited_dists[(size_t)qi * P.out.visited_cap + nvis] = key_dist(key2);
            } else {
              atomicOr(P.status, 1u);
            }
          }
          nvis++;
        }
        degsum += __popcll(__ballot(act1)) + (two ? __popcll(__ballot(act2)) : 0u);
        dcmps += m1 + (two ? m2 : 0u);
        const bool more_left = two ? more_after2 : true;
        const bool skip = (c == 0) || (c < beam / 8 && more_left);
        do_merge = !skip;
      } else {
      for (uint32_t i0 = 0; i0 < P.gstride; i0 += PANN_WAVE) {
        const uint32_t i = i0 + lane;
        uint32_t a = SENTINEL;
        if (i0 == 0 && pref_hit) { if (i < P.gstride) a = pref_val; }
        else if (i < P.gstride) a = row[i];
        const bool act = (a != SENTINEL) && (i < P.degree_limit);
        const uint64_t am = __ballot(act);
        PANN_STAMP(1);
        if (am == 0ull) break;
        degsum += __popcll(am);
        const bool seen = filter_update<true>(H, hmask, act, a, lane);
        const bool keep = act && !seen && ((int64_t)a != self);   // :133
        const uint64_t km = __ballot(keep);
        const uint32_t m = __popcll(km);
        if (keep) Pl[lanes_below(km, lane)] = a;
        dcmps += m;
        PANN_WSYNC();
        PANN_STAMP(2);
        if (m) c = gather_distances<DT, METRIC, LPC, NCH1, PANN_GU>(P, qreg, qlds, Pl, m, cutoff_ord, C, c, lane);
        PANN_WSYNC();
        PANN_STAMP(3);
      }
      const bool skip = (c == 0) || (P.skip_enabled && c < beam / 8 && more_unvisited);   // :162-168
      do_merge = !skip;
      }   // single-vertex path
    }
    if (do_merge) {
      const uint32_t f_old = f;
      const bool track = P.cut_enabled && !first;
      uint32_t lt = 0;                                           // tentative appends to the dropped list
      for (uint32_t c0 = 0; c0 < c; c0 += PANN_WAVE) {           // sort+unique(C) U frontier, trimmed (:173-185)
        const uint32_t cc = min(c - c0, (uint32_t)PANN_WAVE);
        const uint64_t ckey = (lane < (int)cc) ? C[c0 + lane] : KEY_INF;
        const bool fl = lane < (int)f;
        uint32_t myp = 0, rank_c = 0, below_f = 0;
        uint64_t live_mask = 0;
        // 64-bit compares issue at 2.5 x the cost of 32-bit ones on gfx950 (tools/valu_rate.hip), so the loop makes two of them
        // per candidate instead of five: equal keys <=> equal ids (a vertex's distance to this query is one deterministic
        // value, whichever row offered it), and "frontier entry below the candidate" is the complement of "candidate below the
        // entry" once no entry is equal (free lanes hold KEY_INF: above every candidate, masked out by vmask).
        const uint64_t vmask = __ballot(fl);
        const uint32_t cid = (uint32_t)ckey, fid = (uint32_t)fkey;
        for (uint32_t i = 0; i < cc; i++) {
          const uint64_t kk = readlane64(ckey, (int)i);
          const uint32_t kid = (uint32_t)kk;
          const bool g = kk < fkey;
          const uint64_t gtm = __ballot(g);
          const uint64_t eqm = __ballot(fid == kid);                     // already in the frontier (set_union)
          const uint64_t dupm = __ballot(cid == kid) & ((1ull << i) - 1ull);   // duplicate (std::unique)
          if (eqm == 0ull && dupm == 0ull) {
            live_mask |= 1ull << i;
            if (lane == (int)i) myp = f - (uint32_t)__popcll(gtm & vmask);
            rank_c += (kk < ckey) ? 1u : 0u;
            below_f += g ? 1u : 0u;
          }
        }
        const uint32_t nvalid = __popcll(live_mask);
        const bool clive = (lane < (int)cc) && ((live_mask >> lane) & 1ull);
        const uint32_t cpos = rank_c + myp, fpos = (uint32_t)lane + below_f;
        if (clive && cpos < beam) {
          uint32_t flag = 0u;   // re-entry of an already visited vertex (only while the frontier is not full)
          for (uint32_t t = 0; t < ndrop; t++)
            flag |= (__hip_atomic_load(DL + t, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE) == ckey) ? 2u : 0u;
          S[cpos] = ckey; Sv[cpos] = (uint8_t)flag;
        }
        if (fl && fpos < beam) { S[fpos] = fkey; Sv[fpos] = (uint8_t)fflag; }
        if (track) {            // visited entries pushed past the beam by this chunk
          const bool lost = fl && fflag == 1u && fpos >= beam;
          const uint64_t lm = __ballot(lost);
          if (lost) {
            const uint32_t at = ndrop + lt + lanes_below(lm, lane);
            if (at < P.dcap) DL[at] = fkey; else atomicOr(P.status, 2u);
          }
          lt += __popcll(lm);
        }
        PANN_WSYNC();
        f = min(f + nvalid, beam);
        fkey = (lane < (int)f) ? S[lane] : KEY_INF;
        fflag = (lane < (int)f) ? (uint32_t)Sv[lane] : 0u;
        PANN_WSYNC();
      }
      uint32_t f_new = f;
      if (!first && P.cut_enabled && f_new > P.k) {              // cut-prune (:190-195)
        const float dk = key_dist(readlane64(fkey, (int)P.k));
        const float thr = (float)(P.cut * (double)dk);
        const uint64_t thr_key = (uint64_t)f2ord(thr) << 32;     // pair{0, thr}: kept iff key <= thr_key
        const uint32_t ub = __popcll(__ballot(lane < (int)f_new && fkey <= thr_key));
        f_new = max(ub, f_old);
      }
      if (f_new == beam) {
        ndrop = 0;                                               // a full frontier never re-admits anything
      } else if (track) {
        const bool lost = lane >= (int)f_new && lane < (int)f && fflag == 1u;
        const uint64_t lm = __ballot(lost);
        if (lost) {
          const uint32_t at = ndrop + lt + lanes_below(lm, lane);
          if (at < P.dcap) DL[at] = fkey; else atomicOr(P.status, 2u);
        }
        lt += __popcll(lm);
        ndrop = min(ndrop + lt, P.dcap);
        if (lt) __builtin_amdgcn_s_waitcnt(0);
      }
      f = f_new;
      if (lane >= (int)f) { fkey = KEY_INF; fflag = 0u; }
      c = 0;                                                     // candidates.clear() (:182)
    }
    PANN_STAMP(4);
    first = false;
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pref_row));          // no speculative load may outlive its register

  const size_t qo = (size_t)qi * P.out.out_k;
  if (lane < (int)P.out.out_k) {                                 // out_k <= beam <= 64
    const bool ok = lane < (int)f;
    if (P.out.ids) P.out.ids[qo + lane] = ok ? key_id(fkey) : SENTINEL;
    if (P.out.dists) P.out.dists[qo + lane] = ok ? key_dist(fkey) : __builtin_inff();
  }
#ifdef PANN_STAMPS
  if (lane == 0 && P.stamps) for (int i = 0; i < 8; i++) P.stamps[(size_t)qi * 8 + i] = stamp_sum[i];
#endif
  if (lane == 0) {
    if (P.out.frontier_size) P.out.frontier_size[qi] = f;
    if (P.out.visited_count) P.out.visited_count[qi] = nvis;
    if (P.out.dist_cmps) P.out.dist_cmps[qi] = dcmps;
    if (P.out.degree_sum) P.out.degree_sum[qi] = degsum;
  }
}

// =============================================================================================
// 64 < beam <= 128 (the builder's L = 128): the same register-resident frontier with TWO entries
// per lane (entry e lives in lane e % 64, slot e / 64).  LDS per query: the 16 KB filter + a 128-entry
// scatter scratch + candidates = 18.2 KB -> 9 queries per CU (the LDS-frontier kernel needs 19.9 KB).
// =============================================================================================
// CODES (with HASH_LDS): the whole filter in LDS as 12-bit class codes (filter_update_codes), the codes of a row's neighbours
// read with the row from P.gcode -- the builder's searches when the index keeps the codes (filter_codes.hip).
template <int DT, int METRIC, int LPC, bool NCH1, bool HASH_LDS, bool CODES = false>
__global__ void __launch_bounds__(PANN_WAVE) beam_search_b128_kernel(BSParams P) {
  static_assert(!CODES || HASH_LDS, "the code table lives in LDS");
  constexpr int RB = 2;
  const int lane = threadIdx.x;
  extern __shared__ __align__(16) uint8_t smem[];
  uint64_t* S = reinterpret_cast<uint64_t*>(smem);          // [128] merge scatter scratch; first 256 B double as Pl
  uint64_t* C = S + 64 * RB;                                // [ccap] candidates (unsorted)
  uint8_t* Sv = reinterpret_cast<uint8_t*>(C + P.ccap);     // [128] flags of the scatter scratch
  uint4* qlds = reinterpret_cast<uint4*>(Sv + 64 * RB);     // [nch*LPC] query (generic variant)
  uint32_t* Hl = reinterpret_cast<uint32_t*>(qlds + (NCH1 ? 0 : P.nch * LPC));  // [1<<bits] if HASH_LDS, else the LDS part of a split table
  uint8_t* T = reinterpret_cast<uint8_t*>(S);               // 1 KB replay scratch of the HBM-table filter: S is idle outside the merge
  uint32_t* Pl = reinterpret_cast<uint32_t*>(S);

  const uint32_t hsize = 1u << P.bits, hmask = hsize - 1u;
  const uint32_t beam = P.beam;
  const uint32_t BIG_ORD = f2ord(2147483648.0f);
  const LdsPart Lp{reinterpret_cast<uint8_t*>(Hl), HASH_LDS ? 0u : (P.hsplit == 3 ? hsize : P.hsplit ? hsize - (hsize >> P.hsplit) : 0u), P.p24};
  // filter in LDS: one query per block.  Filter in HBM (16 KB of LDS would cap a CU at 8 queries): persistent
  // blocks, one table per block, queries pulled from a counter.
  uint32_t slot = blockIdx.x;
  if constexpr (!HASH_LDS) {
    slot = 0;
    if (lane == 0) slot = atomicAdd(P.work_counter, 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
  }
  while (slot < P.nq) {
  const uint32_t qi = P.order ? P.order[slot] : slot;
  uint32_t* H = HASH_LDS ? Hl : P.hash_global + ((size_t)blockIdx.x << P.bits);
  const uint32_t hb = HASH_LDS ? 0u : P.hsplit;
  uint8_t* const P8 = reinterpret_cast<uint8_t*>(Hl);                    // CODES: 8-bit plane [hsize], then
  uint32_t* const P4 = Hl + (hsize >> 2);                                 // ... the 4-bit plane, 8 slots per dword [hsize / 8]
  if constexpr (CODES) {
    for (uint32_t i = lane; i < (hsize >> 2) + (hsize >> 3); i += PANN_WAVE) Hl[i] = 0xFFFFFFFFu;
  } else {
    if (hb != 3) for (uint32_t i = lane; i < (hsize >> hb); i += PANN_WAVE) hstore<HASH_LDS>(H, i, SENTINEL);
    if (hb) lds_part_clear(Lp, lane);
  }
  const int64_t self = P.query_ids ? (int64_t)P.query_ids[qi] : -1;
  const uint8_t* qrow = P.query_ids ? P.points + (uint64_t)self * P.pstride : P.queries + (uint64_t)qi * P.qstride;
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(qrow, P.dbytes, P.nch, qreg, qlds, lane);
  hsync<HASH_LDS>();

  uint32_t f = 0, c = 0, nvis = 0, dcmps = P.nstarts, degsum = 0, ndrop = 0;
  uint64_t fkey[RB] = {KEY_INF, KEY_INF};
  uint32_t fflag[RB] = {0, 0};
  uint64_t* DL = P.dropped + (size_t)qi * P.dcap;
#ifdef PANN_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  auto entry_key = [&](uint32_t e) -> uint64_t {            // uniform e
    return (e < 64) ? readlane64(fkey[0], (int)e) : readlane64(fkey[1], (int)e - 64);
  };

  {   // start points (:66-70); nstarts <= 64 in this kernel
    const bool act = lane < (int)P.nstarts;
    const uint32_t a = act ? P.starts[(size_t)qi * P.starts_stride + lane] : 0u;
    if constexpr (CODES) (void)filter_update_codes(P8, P4, hmask, act, a, act ? (uint32_t)P.rank16[a] : 0u, lane, T);
    else (void)filter_update<HASH_LDS>(H, hmask, act, a, lane, T, hb, Lp);
    if (act) Pl[lane] = a;
    PANN_WSYNC();
    c = gather_distances<DT, METRIC, LPC, NCH1, 4, false>(P, qreg, qlds, Pl, P.nstarts, 0xFFFFFFFFu, C, c, lane);
    PANN_WSYNC();
  }

  // Speculative adjacency-row fetch: while the candidates of vertex v are gathered, the row of the frontier's NEXT
  // unvisited entry is already requested; it is the next vertex unless the merge puts something in front of it.
  uint32_t pref_id = SENTINEL, pref_row = SENTINEL, pref_code = 0;
  bool first = true;
  for (;;) {
    bool do_merge = first;
    if (!first) {
      const uint64_t um0 = __ballot(lane < (int)f && fflag[0] == 0);
      const uint64_t um1 = __ballot(lane + 64 < (int)f && fflag[1] == 0);
      if ((um0 | um1) == 0ull || nvis >= P.limit) break;
      const int cur_idx = um0 ? __ffsll((unsigned long long)um0) - 1 : 64 + __ffsll((unsigned long long)um1) - 1;
      const bool more_unvisited = (__popcll(um0) + __popcll(um1)) > 1;          // offset + 1 < remain (:165)
      PANN_STAMP(0);
      const uint64_t cur_key = entry_key((uint32_t)cur_idx);
      const uint32_t cur = key_id(cur_key);
      // ---- two vertices per memory round trip.  While merges are being skipped (:162-168: fewer than beam/8 candidates
      // and another unvisited entry) the frontier and the cutoff do not change, so the vertex after `cur` is already known:
      // the second unvisited entry.  83 % of the builder's iterations skip (66 % of beam-64 queries), so both adjacency
      // rows are fetched together, the filter is replayed row after row (row 2 sees row 1's updates, as in the
      // sequential loop) and ONE gather fetches the survivors of both -- twice the bytes in flight per wave in the phase
      // that is 52 % of this kernel's time.  If the candidates of row 1 alone end the skipping (or row 2 does not fit),
      // row 2 never happened: its candidates are cut off the list and its table writes are put back (FilterUndo).
      if ((CODES ? PANN_B128_PAIR_CODES : PANN_B128_PAIR) && P.skip_enabled && more_unvisited && P.gstride <= PANN_WAVE && nvis + 1 < P.limit) {
        uint64_t r0 = um0, r1 = um1;
        if (r0) r0 &= r0 - 1; else r1 &= r1 - 1;
        const int idx2 = r0 ? __ffsll((unsigned long long)r0) - 1 : 64 + __ffsll((unsigned long long)r1) - 1;
        if (r0) r0 &= r0 - 1; else r1 &= r1 - 1;
        const bool more_after2 = (r0 | r1) != 0ull;
        const uint64_t key2 = entry_key((uint32_t)idx2);
        const uint32_t cur2 = key_id(key2);
        uint32_t cutoff2 = BIG_ORD;
        if (f == beam) cutoff2 = (uint32_t)(entry_key(f - 1) >> 32);
        uint32_t a1 = SENTINEL, a2 = SENTINEL, code1 = 0, code2 = 0;
        if (lane < (int)P.gstride) {
          a1 = (pref_id == cur) ? pref_row : P.graph[(size_t)cur * P.gstride + lane];
          a2 = P.graph[(size_t)cur2 * P.gstride + lane];
          if constexpr (CODES) {
            code1 = (pref_id == cur) ? pref_code : (uint32_t)P.gcode[(size_t)cur * P.gstride + lane];
            code2 = P.gcode[(size_t)cur2 * P.gstride + lane];
          }
        }
        pref_id = SENTINEL;
        const bool act1 = (a1 != SENTINEL) && ((uint32_t)lane < P.degree_limit);
        const bool act2 = (a2 != SENTINEL) && ((uint32_t)lane < P.degree_limit);
        FilterUndo u2;
        bool seen1, seen2;
        if constexpr (CODES) {
          seen1 = filter_update_codes(P8, P4, hmask, act1, a1, code1, lane, T);
          seen2 = filter_update_codes(P8, P4, hmask, act2, a2, code2, lane, T, &u2);
        } else {
          seen1 = filter_update<HASH_LDS>(H, hmask, act1, a1, lane, T, hb, Lp);
          seen2 = filter_update<HASH_LDS>(H, hmask, act2, a2, lane, T, hb, Lp, &u2);
        }
        const bool keep1 = act1 && !seen1 && ((int64_t)a1 != self);
        const bool keep2 = act2 && !seen2 && ((int64_t)a2 != self);
        const uint64_t km1 = __ballot(keep1), km2 = __ballot(keep2);
        const uint32_t m1 = __popcll(km1), m2 = __popcll(km2);
        const bool both = (c + m1 + m2 <= P.ccap);                     // the candidate list is sized for beam/8 - 1 + ONE row
        if (keep1) Pl[lanes_below(km1, lane)] = a1;
        if (both && keep2) Pl[m1 + lanes_below(km2, lane)] = a2;
        PANN_WSYNC();
        uint32_t c_after1 = c;
        if (both) {
          if (m1 + m2) c = gather_distances_split<DT, METRIC, LPC, NCH1, PANN_GU_B128(LPC, NCH1)>(P, qreg, qlds, Pl, m1 + m2, m1, cutoff2, C, c, lane, &c_after1);
        } else {
          if (m1) c = gather_distances<DT, METRIC, LPC, NCH1, PANN_GU_B128(LPC, NCH1), false>(P, qreg, qlds, Pl, m1, cutoff2, C, c, lane);
          c_after1 = c;
        }
        PANN_WSYNC();
        // the reference's decision after vertex 1 (another unvisited entry exists): skip iff its list is still short
        const bool two = both && (c_after1 == 0 || c_after1 < beam / 8);
        if (!two) {
          c = c_after1;
          if constexpr (CODES) filter_undo_codes(P8, P4, u2);
          else filter_undo<HASH_LDS>(H, u2, hb, Lp);
        }
        // commit: visited.insert(current) (:112-114), counters
        if (cur_idx < 64) { if (lane == cur_idx) fflag[0] = 1; } else { if (lane == cur_idx - 64) fflag[1] = 1; }
        if (two) { if (idx2 < 64) { if (lane == idx2) fflag[0] = 1; } else { if (lane == idx2 - 64) fflag[1] = 1; } }
        if (lane == 0 && P.out.visited_cap) {
          const uint32_t nv = two ? 2u : 1u;
          if (nvis + nv <= P.out.visited_cap) {
            const size_t at = (size_t)qi * P.out.visited_cap + nvis;
            if (P.out.visited_ids) { P.out.visited_ids[at] = cur; if (two) P.out.visited_ids[at + 1] = cur2; }
            if (P.out.visited_dists) { P.out.visited_dists[at] = key_dist(cur_key); if (two) P.out.visited_dists[at + 1] = key_dist(key2); }
          } else {
            atomicOr(P.status, 1u);
          }
        }
        nvis += two ? 2u : 1u;
        degsum += __popcll(__ballot(act1)) + (two ? __popcll(__ballot(act2)) : 0u);
        dcmps += m1 + (two ? m2 : 0u);
        const bool more_left = two ? more_after2 : true;
        const bool skip = (c == 0) || (c < beam / 8 && more_left);
        do_merge = !skip;
      } else {
      if (cur_idx < 64) { if (lane == cur_idx) fflag[0] = 1; } else { if (lane == cur_idx - 64) fflag[1] = 1; }
      if (lane == 0 && P.out.visited_cap) {
        if (nvis < P.out.visited_cap) {
          if (P.out.visited_ids) P.out.visited_ids[(size_t)qi * P.out.visited_cap + nvis] = cur;
          if (P.out.visited_dists) P.out.visited_dists[(size_t)qi * P.out.visited_cap + nvis] = key_dist(cur_key);
        } else {
          atomicOr(P.status, 1u);
        }
      }
      nvis++;
      uint32_t cutoff_ord = BIG_ORD;
      if (f == beam) cutoff_ord = (uint32_t)(entry_key(f - 1) >> 32);

      const uint32_t* row = P.graph + (size_t)cur * P.gstride;
      const uint16_t* crow = CODES ? P.gcode + (size_t)cur * P.gstride : nullptr;
      const bool pref_hit = (pref_id == cur);
      const uint32_t pref_val = pref_row, pref_cval = pref_code;
      uint32_t next_id = SENTINEL;   // the entry after `cur` among the unvisited ones
      {
        uint64_t r0 = um0, r1 = um1;
        if (r0) r0 &= r0 - 1; else r1 &= r1 - 1;
        if (r0 | r1) {
          const int nx = r0 ? __ffsll((unsigned long long)r0) - 1 : 64 + __ffsll((unsigned long long)r1) - 1;
          next_id = key_id(entry_key((uint32_t)nx));
        }
      }
      pref_id = SENTINEL;
      for (uint32_t i0 = 0; i0 < P.gstride; i0 += PANN_WAVE) {
        const uint32_t i = i0 + lane;
        uint32_t a = SENTINEL, code = 0;
        if (i0 == 0 && pref_hit) { a = pref_val; code = pref_cval; }
        else if (i < P.gstride) { a = row[i]; if constexpr (CODES) code = crow[i]; }
        const bool act = (a != SENTINEL) && (i < P.degree_limit);
        const uint64_t am = __ballot(act);
        PANN_STAMP(1);
        if (am == 0ull) break;
        degsum += __popcll(am);
        bool seen;
        if constexpr (CODES) seen = filter_update_codes(P8, P4, hmask, act, a, code, lane, T);
        else seen = filter_update<HASH_LDS>(H, hmask, act, a, lane, T, hb, Lp);
        const bool keep = act && !seen && ((int64_t)a != self);
        const uint64_t km = __ballot(keep);
        const uint32_t m = __popcll(km);
        if (keep) Pl[lanes_below(km, lane)] = a;
        dcmps += m;
        PANN_WSYNC();
        PANN_STAMP(2);
        // requested after the filter's table loads, right before the gather's.  Only for base-point queries (the
        // builder's searches: f32 d=96 search phase -5.5 %); external beam-128 queries measured 2 % slower with it.
        if (i0 == 0 && next_id != SENTINEL && P.query_ids) {
          pref_id = next_id;
          pref_row = (lane < (int)P.gstride) ? P.graph[(size_t)next_id * P.gstride + lane] : SENTINEL;
          if constexpr (CODES) pref_code = (lane < (int)P.gstride) ? (uint32_t)P.gcode[(size_t)next_id * P.gstride + lane] : 0u;
        }
        if (m) c = gather_distances<DT, METRIC, LPC, NCH1, PANN_GU_B128(LPC, NCH1), false>(P, qreg, qlds, Pl, m, cutoff_ord, C, c, lane);
        PANN_WSYNC();
        PANN_STAMP(3);
      }
      const bool skip = (c == 0) || (P.skip_enabled && c < beam / 8 && more_unvisited);
      do_merge = !skip;
      }   // single-vertex path
    }
    if (do_merge) {
      const uint32_t f_old = f;
      const bool track = P.cut_enabled && !first;
      uint32_t lt = 0;
      for (uint32_t c0 = 0; c0 < c; c0 += PANN_WAVE) {
        const uint32_t cc = min(c - c0, (uint32_t)PANN_WAVE);
        const uint64_t ckey = (lane < (int)cc) ? C[c0 + lane] : KEY_INF;
        const bool fl[RB] = {lane < (int)f, lane + 64 < (int)f};
        uint32_t myp = 0, rank_c = 0, below_f[RB] = {0, 0};
        uint64_t live_mask = 0;
        // three 64-bit compares per candidate instead of eight (see the beam-64 kernel): ids decide equality, and the
        // entries below a candidate are those not above it
        const uint64_t vmask0 = __ballot(fl[0]), vmask1 = __ballot(fl[1]);
        const uint32_t cid = (uint32_t)ckey, fid0 = (uint32_t)fkey[0], fid1 = (uint32_t)fkey[1];
        for (uint32_t i = 0; i < cc; i++) {
          const uint64_t kk = readlane64(ckey, (int)i);
          const uint32_t kid = (uint32_t)kk;
          const bool g0 = kk < fkey[0], g1 = kk < fkey[1];
          const uint64_t gt0 = __ballot(g0), gt1 = __ballot(g1);
          const uint64_t eqm = __ballot(fid0 == kid || fid1 == kid);
          const uint64_t dupm = __ballot(cid == kid) & ((1ull << i) - 1ull);
          if (eqm == 0ull && dupm == 0ull) {
            live_mask |= 1ull << i;
            if (lane == (int)i) myp = f - (uint32_t)__popcll(gt0 & vmask0) - (uint32_t)__popcll(gt1 & vmask1);
            rank_c += (kk < ckey) ? 1u : 0u;
            below_f[0] += g0 ? 1u : 0u;
            below_f[1] += g1 ? 1u : 0u;
          }
        }
        const uint32_t nvalid = __popcll(live_mask);
        const bool clive = (lane < (int)cc) && ((live_mask >> lane) & 1ull);
        const uint32_t cpos = rank_c + myp;
        if (clive && cpos < beam) {
          uint32_t flag = 0u;
          for (uint32_t t = 0; t < ndrop; t++)
            flag |= (__hip_atomic_load(DL + t, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE) == ckey) ? 2u : 0u;
          S[cpos] = ckey; Sv[cpos] = (uint8_t)flag;
        }
#pragma unroll
        for (int r = 0; r < RB; r++) {
          const uint32_t fpos = (uint32_t)lane + 64 * r + below_f[r];
          if (fl[r] && fpos < beam) { S[fpos] = fkey[r]; Sv[fpos] = (uint8_t)fflag[r]; }
          if (track) {
            const bool lost = fl[r] && fflag[r] == 1u && fpos >= beam;
            const uint64_t lm = __ballot(lost);
            if (lost) {
              const uint32_t at = ndrop + lt + lanes_below(lm, lane);
              if (at < P.dcap) DL[at] = fkey[r]; else atomicOr(P.status, 2u);
            }
            lt += __popcll(lm);
          }
        }
        PANN_WSYNC();
        f = min(f + nvalid, beam);
#pragma unroll
        for (int r = 0; r < RB; r++) {
          const bool in = lane + 64 * r < (int)f;
          fkey[r] = in ? S[lane + 64 * r] : KEY_INF;
          fflag[r] = in ? (uint32_t)Sv[lane + 64 * r] : 0u;
        }
        PANN_WSYNC();
      }
      uint32_t f_new = f;
      if (!first && P.cut_enabled && f_new > P.k) {
        const float dk = key_dist(entry_key(P.k));
        const float thr = (float)(P.cut * (double)dk);
        const uint64_t thr_key = (uint64_t)f2ord(thr) << 32;
        const uint32_t ub = __popcll(__ballot(lane < (int)f_new && fkey[0] <= thr_key)) +
                            __popcll(__ballot(lane + 64 < (int)f_new && fkey[1] <= thr_key));
        f_new = max(ub, f_old);
      }
      if (f_new == beam) {
        ndrop = 0;
      } else if (track) {
#pragma unroll
        for (int r = 0; r < RB; r++) {
          const int e = lane + 64 * r;
          const bool lost = e >= (int)f_new && e < (int)f && fflag[r] == 1u;
          const uint64_t lm = __ballot(lost);
          if (lost) {
            const uint32_t at = ndrop + lt + lanes_below(lm, lane);
            if (at < P.dcap) DL[at] = fkey[r]; else atomicOr(P.status, 2u);
          }
          lt += __popcll(lm);
        }
        ndrop = min(ndrop + lt, P.dcap);
        if (lt) __builtin_amdgcn_s_waitcnt(0);
      }
      f = f_new;
#pragma unroll
      for (int r = 0; r < RB; r++) if (lane + 64 * r >= (int)f) { fkey[r] = KEY_INF; fflag[r] = 0u; }
      c = 0;
    }
    PANN_STAMP(4);
    first = false;
  }
#ifdef PANN_STAMPS
  if (lane == 0 && P.stamps) for (int i = 0; i < 8; i++) P.stamps[(size_t)qi * 8 + i] = stamp_sum[i];
#endif

  const size_t qo = (size_t)qi * P.out.out_k;
#pragma unroll
  for (int r = 0; r < RB; r++) {
    const uint32_t j = lane + 64 * r;
    if (j < P.out.out_k) {
      const bool ok = j < f;
      if (P.out.ids) P.out.ids[qo + j] = ok ? key_id(fkey[r]) : SENTINEL;
      if (P.out.dists) P.out.dists[qo + j] = ok ? key_dist(fkey[r]) : __builtin_inff();
    }
  }
  if (lane == 0) {
    if (P.out.frontier_size) P.out.frontier_size[qi] = f;
    if (P.out.visited_count) P.out.visited_count[qi] = nvis;
    if (P.out.dist_cmps) P.out.dist_cmps[qi] = dcmps;
    if (P.out.degree_sum) P.out.degree_sum[qi] = degsum;
  }
  if constexpr (HASH_LDS) {
    break;
  } else {
    PANN_WSYNC();
    if (lane == 0) slot = atomicAdd(P.work_counter, 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
  }
  }   // while (slot < P.nq)
}

// ---------------------------------------------------------------------------------------------
// host side: layout choice, LDS budget, dispatch
// ---------------------------------------------------------------------------------------------

// Lanes-per-candidate and chunks-per-lane for a row of dbytes.  One chunk (query in registers)
// whenever the row is 128/256/512 B after padding to 16*LPC; otherwise LPC=4 (64-B granules,
// the reference's own row alignment, point_range.h:94) or LPC=16 for long rows.
void choose_point_layout(uint32_t dbytes, uint32_t* lpc, uint32_t* nch) {
  const uint32_t r64 = (dbytes + 63) / 64 * 64;
  if (const char* f = ab_env("PANN_FORCE_LPC")) {          // diagnostic A/B switch: lanes per candidate 4 / 8 / 16
    const uint32_t l = (uint32_t)atoi(f);
    if ((l == 4 || l == 8 || l == 16) && r64 % (l * 16) == 0 && r64 / (l * 16) > 1) { *lpc = l; *nch = r64 / (l * 16); return; }
  }
  if (r64 == 128) { *lpc = 8; *nch = 1; return; }
  if (r64 == 256) { *lpc = 16; *nch = 1; return; }
  if (r64 == 512) { *lpc = 32; *nch = 1; return; }
  if (r64 % 256 == 0) { *lpc = 16; *nch = r64 / 256; return; }
  if (r64 % 128 == 0) { *lpc = 8; *nch = r64 / 128; return; }      // e.g. f32 d=96: 384 B = 3 chunks of 8 lanes x 16 B
  *lpc = 4; *nch = r64 / 64;
}

static uint32_t filter_bits(int64_t beam) {  // :52
  double l = std::ceil(std::log2((double)beam * (double)beam)) - 2.0;
  int b = (int)l;
  return (uint32_t)(b < 10 ? 10 : b);
}

struct Plan {
  uint32_t bits, bcap, ccap, deg_eff, dcap, lds_bytes; bool hash_lds; uint32_t slots; bool b64; bool b128;
  bool b128_codes; // beam 91..128 with the filter as 12-bit class codes in LDS (the index keeps the codes: filter_codes.hip)
  bool b128_hbm;   // beam 65..128 with the filter in HBM (persistent blocks)
  uint32_t hsplit; // ... of which the part with an LDS share (filter_update split mode)
  uint32_t p24;    // ... stored as planar 24-bit entries
};

static Plan make_plan(const DeviceIndex& ix, const SearchArgs& a) {
  Plan p;
  p.bits = filter_bits(a.beam);
  const uint32_t beam = (uint32_t)a.beam;
  p.bcap = (std::max<uint32_t>(beam, a.nstarts) + 63) / 64 * 64;
  int64_t dl = std::min<int64_t>(std::max<int64_t>(a.degree_limit, 0), (int64_t)ix.max_deg);
  p.deg_eff = (uint32_t)dl;
  // at merge time |C| <= (beam/8 - 1) + deg_eff (accumulation stops at beam/8); starts first
  p.ccap = (std::max<uint32_t>(beam / 8 + p.deg_eff, a.nstarts) + 63) / 64 * 64;
  p.dcap = std::max<uint32_t>(a.dcap, 64);
  const bool nch1 = layout_query_in_registers(ix);
  size_t fixed = (size_t)p.bcap * 8 * 2 + (size_t)p.ccap * 8 + (size_t)p.bcap * 2 +
                 (size_t)p.ccap * 2 + (nch1 ? 0 : (size_t)ix.nch * ix.lpc * 16);
  size_t hbytes = (size_t)4 << p.bits;
  p.hash_lds = (hbytes <= 16384) && (fixed + hbytes <= 64 * 1024);
  p.lds_bytes = (uint32_t)(fixed + (p.hash_lds ? hbytes : 1024));      // HBM filter: 1 KB replay scratch (filter_update)
  p.slots = 256 * 8;
  p.b64 = p.hash_lds && p.bcap == 64;
  if (p.b64) {   // register-frontier kernel: scratch[64] + candidates (exact, 8-entry granules) + flags + query + filter
    p.ccap = (std::max<uint32_t>(beam / 8 + p.deg_eff, a.nstarts) + 7) / 8 * 8;
    p.lds_bytes = (uint32_t)(64 * 8 + (size_t)p.ccap * 8 + 64 + (nch1 ? 0 : (size_t)ix.nch * ix.lpc * 16) + hbytes);
  }
  p.b128 = p.hash_lds && p.bcap == 128 && a.nstarts <= 64;
  p.b128_hbm = false;
  p.b128_codes = p.b128 && p.bits == FILTER_CODE_BITS && ix.codes_valid && ix.gcode && ix.rank16 && ix.gstride <= PANN_WAVE;
  if (p.b128_codes) {        // 6 KB of table + the merge scratch (which doubles as the replay scratch T) + candidates + flags + query
    p.ccap = (std::max<uint32_t>(beam / 8 + p.deg_eff, a.nstarts) + 7) / 8 * 8;
    p.hsplit = 0; p.p24 = 0;
    p.lds_bytes = (uint32_t)(128 * 8 + (size_t)p.ccap * 8 + 128 + (nch1 ? 0 : (size_t)ix.nch * ix.lpc * 16) + ((size_t)1 << p.bits) * 3 / 2);
  } else if (p.b128) {
    p.ccap = (std::max<uint32_t>(beam / 8 + p.deg_eff, a.nstarts) + 7) / 8 * 8;
    // 16 KB of filter per query leaves 8 queries per CU (2 048 on the chip).  Larger batches put the table in HBM
    // (Infinity-Cache resident): one dependent load per adjacency row, 3x the queries per CU; measured +17 % on the
    // C3 build's search phase and +10 % on beam-128 queries -- the 64 single-word table requests per row make the
    // L2 request rate the new limit (in-kernel stamps: every phase 2-3x longer at 3x the occupancy).
    static const bool force_lds = ab_env("PANN_B128_LDS") != nullptr;       // diagnostic A/B switch
    p.b128_hbm = (hbytes > 8192) && a.nq > 2048 && !force_lds;
    p.hsplit = 0;
    if (p.b128_hbm) {   // one table per block, >= the resident waves
      static const char* sp = ab_env("PANN_B128_SPLIT");
      p.hsplit = sp ? (uint32_t)atoi(sp) : 1u;     // measured: half in LDS 2.99 M q/s, none 2.46, three quarters 2.67, whole table in LDS 2.28
      if (p.hsplit > 3) p.hsplit = 0;
      static const bool no24 = ab_env("PANN_B128_NO24") != nullptr;              // diagnostic A/B switch
      if (p.hsplit == 3 && (no24 || !(ix.n < 0xFFFFFFull))) p.hsplit = 1;     // the all-LDS mode exists for 24-bit planes only
      p.hash_lds = false; p.slots = 256 * 32;
      p.p24 = (p.hsplit && ix.n < 0xFFFFFFull && !no24) ? 1u : 0u;
      // the 1 KB replay scratch of filter_update aliases the merge scratch S; the LDS share of the table follows the query
      hbytes = p.hsplit == 3 ? ((size_t)3 << p.bits)
             : p.hsplit ? (((size_t)1 << p.bits) - ((size_t)1 << (p.bits - p.hsplit))) * (p.p24 ? 3 : 4) : 0;
    }
    p.lds_bytes = (uint32_t)(128 * 8 + (size_t)p.ccap * 8 + 128 + (nch1 ? 0 : (size_t)ix.nch * ix.lpc * 16) + hbytes);
  }
  return p;
}

size_t search_workspace_bytes(const DeviceIndex& ix, const SearchArgs& a) {
  Plan p = make_plan(ix, a);
  size_t need = 256;                                   // counter + status
  need += (size_t)a.nq * p.dcap * 8;                   // dropped lists
  if (!p.hash_lds) need += ((size_t)std::min<uint64_t>(a.nq, p.slots) << p.bits) * 4;
  return need;
}

template <int DT, int METRIC, int LPC, bool NCH1>
static hipError_t launch_variant(const BSParams& P, const Plan& p, hipStream_t stream) {
  if (p.b64) {   // frontier in registers
    auto kern = beam_search_b64_kernel<DT, METRIC, LPC, NCH1>;
    hipLaunchKernelGGL(kern, dim3(P.nq), dim3(PANN_WAVE), p.lds_bytes, stream, P);
  } else if (p.b128_codes) {   // two frontier entries per lane, filter of class codes
    auto kern = beam_search_b128_kernel<DT, METRIC, LPC, NCH1, true, true>;
    hipLaunchKernelGGL(kern, dim3(P.nq), dim3(PANN_WAVE), p.lds_bytes, stream, P);
  } else if (p.b128) {   // two frontier entries per lane
    if (p.b128_hbm) {
      auto kern = beam_search_b128_kernel<DT, METRIC, LPC, NCH1, false>;
      const uint32_t grid = (uint32_t)std::min<uint64_t>(P.nq, p.slots);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(PANN_WAVE), p.lds_bytes, stream, P);
    } else {
      auto kern = beam_search_b128_kernel<DT, METRIC, LPC, NCH1, true>;
      hipLaunchKernelGGL(kern, dim3(P.nq), dim3(PANN_WAVE), p.lds_bytes, stream, P);
    }
  } else if (p.hash_lds) {
    auto kern = beam_search_kernel<DT, METRIC, LPC, NCH1, true>;
    if (p.lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3(P.nq), dim3(PANN_WAVE), p.lds_bytes, stream, P);
  } else {
    auto kern = beam_search_kernel<DT, METRIC, LPC, NCH1, false>;
    if (p.lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(P.nq, p.slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PANN_WAVE), p.lds_bytes, stream, P);
  }
  return hipGetLastError();
}

template <int DT, int METRIC>
static hipError_t launch_layout(const DeviceIndex& ix, const BSParams& P, const Plan& p, hipStream_t s) {
  if (layout_query_in_registers(ix)) {
    if (ix.lpc == 8) return launch_variant<DT, METRIC, 8, true>(P, p, s);
    if (ix.lpc == 16) return launch_variant<DT, METRIC, 16, true>(P, p, s);
    if (ix.lpc == 32) return launch_variant<DT, METRIC, 32, true>(P, p, s);
  }
  if (ix.lpc == 4) return launch_variant<DT, METRIC, 4, false>(P, p, s);
  if (ix.lpc == 8) return launch_variant<DT, METRIC, 8, false>(P, p, s);
  if (ix.lpc == 16) return launch_variant<DT, METRIC, 16, false>(P, p, s);
  return hipErrorInvalidValue;
}

int launch_beam_search(const DeviceIndex& ix, const SearchArgs& a, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
  if (a.nq == 0) return PANN_OK;
  if (a.nq > 0xFFFFFFF0ull) { set_error("pann_batch_search: nq too large"); return PANN_ERR_BAD_ARG; }
  if (a.beam <= 0 || a.beam > 65536) { set_error("pann_batch_search: beam out of range [1,65536]"); return PANN_ERR_BAD_ARG; }
  if (a.nstarts == 0) {  // beamSearch.h:38-41
    set_error("beam search expects at least one start point"); return PANN_ERR_BAD_ARG;
  }
  if ((int64_t)a.nstarts > a.beam) { set_error("pann_batch_search: more start points than beam"); return PANN_ERR_BAD_ARG; }
  if (a.out.out_k > (uint64_t)a.beam) {  // beamSearch.h:368-372
    set_error("Error: beam search parameter Q same size or smaller than k"); return PANN_ERR_BAD_ARG;
  }
  if ((a.queries == nullptr) == (a.query_ids == nullptr)) {
    set_error("pann_batch_search: exactly one of queries / query_ids must be given"); return PANN_ERR_BAD_ARG;
  }
  Plan p = make_plan(ix, a);
  if (p.lds_bytes > 160 * 1024) { set_error("pann_batch_search: beam/degree too large for LDS state"); return PANN_ERR_UNSUPPORTED; }
  if (search_workspace_bytes(ix, a) > ws_bytes) { set_error("pann_batch_search: workspace too small"); return PANN_ERR_BAD_ARG; }

  BSParams P;
  P.points = ix.points; P.pstride = ix.pstride; P.dbytes = ix.dbytes; P.nch = ix.nch; P.exact = ix.exact;
  P.graph = ix.graph; P.gstride = ix.gstride; P.max_deg = ix.max_deg;
  P.queries = a.queries; P.qstride = a.qstride; P.query_ids = a.query_ids;
  P.starts = a.starts; P.nstarts = a.nstarts; P.starts_stride = a.starts_per_query ? a.nstarts : 0u; P.nq = (uint32_t)a.nq;
  P.k = (uint32_t)std::max<int64_t>(a.k, 0); P.beam = (uint32_t)a.beam;
  P.limit = (uint32_t)std::min<int64_t>(std::max<int64_t>(a.limit, 0), 0xFFFFFFFFll);
  P.degree_limit = p.deg_eff; P.cut = a.cut;
  P.skip_enabled = (a.limit >= 2 * a.beam) ? 1u : 0u;
  P.cut_enabled = (a.k > 0 && ix.metric == PANN_L2) ? 1u : 0u;
  P.bits = p.bits; P.bcap = p.bcap; P.ccap = p.ccap;
  uint8_t* w = (uint8_t*)ws;
  P.work_counter = (uint32_t*)w; P.status = (uint32_t*)(w + 64);
  P.dropped = (uint64_t*)(w + 256); P.dcap = p.dcap;
  {
    static const char* pf = ab_env("PANN_PRIO_FROM");              // A/B switch: tenths of the grid without priority (default 8; 10 = off)
    const uint32_t tenths = pf ? (uint32_t)atoi(pf) : 8u;
    P.prio_start = tenths >= 10 ? 0xFFFFFFFFu : (uint32_t)((uint64_t)a.nq * tenths / 10);
  }
  P.hsplit = p.b128_hbm ? p.hsplit : 0u;
  P.p24 = p.b128_hbm ? p.p24 : 0u;
  P.gcode = ix.gcode; P.rank16 = ix.rank16;
  P.order = a.order;
  P.hash_global = p.hash_lds ? nullptr : (uint32_t*)(w + 256 + (size_t)a.nq * p.dcap * 8);
  P.out = a.out;
  P.stamps = nullptr;
#ifdef PANN_STAMPS
  static unsigned long long* d_stamps = nullptr; static size_t stamps_cap = 0;
  if (stamps_cap < a.nq) { if (d_stamps) (void)hipFree(d_stamps); (void)hipMalloc((void**)&d_stamps, a.nq * 64); stamps_cap = a.nq; }
  P.stamps = d_stamps;
#endif
  PANN_HIP(hipMemsetAsync(w, 0, 256, stream));

  hipError_t e = hipErrorInvalidValue;
#define PANN_DISPATCH(DT, MT) if (ix.dtype == DT && ix.metric == MT) e = launch_layout<DT, MT>(ix, P, p, stream);
  PANN_DISPATCH(PANN_U8, PANN_L2) PANN_DISPATCH(PANN_U8, PANN_MIPS)
  PANN_DISPATCH(PANN_I8, PANN_L2) PANN_DISPATCH(PANN_I8, PANN_MIPS)
  PANN_DISPATCH(PANN_F32, PANN_L2) PANN_DISPATCH(PANN_F32, PANN_MIPS)
  PANN_DISPATCH(PANN_F16, PANN_L2) PANN_DISPATCH(PANN_F16, PANN_MIPS)
  PANN_DISPATCH(PANN_BF16, PANN_L2) PANN_DISPATCH(PANN_BF16, PANN_MIPS)
#undef PANN_DISPATCH
  if (e != hipSuccess) return hip_fail(e, "beam_search_kernel launch");
  // the status word follows the results on the launch stream (pann_search_out::status, device pointer here)
  if (a.out.status) PANN_HIP(hipMemcpyAsync(a.out.status, P.status, 4, hipMemcpyDeviceToDevice, stream));
#ifdef PANN_STAMPS
  if (ab_env("PANN_STAMPS_PRINT") && a.nq >= 1000) {
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> h(a.nq * 8);
    (void)hipMemcpy(h.data(), P.stamps, a.nq * 64, hipMemcpyDeviceToHost);
    double sum[8] = {0}; for (size_t q = 0; q < a.nq; q++) for (int i = 0; i < 8; i++) sum[i] += (double)h[q * 8 + i];
    double tot = 0; for (int i = 0; i < 5; i++) tot += sum[i];
    fprintf(stderr, "[stamps] per-query cycles: head %.0f row %.0f filter %.0f gather %.0f merge %.0f total %.0f  (shares %.1f%% %.1f%% %.1f%% %.1f%% %.1f%%)\n",
            sum[0] / a.nq, sum[1] / a.nq, sum[2] / a.nq, sum[3] / a.nq, sum[4] / a.nq, tot / a.nq, 100 * sum[0] / tot, 100 * sum[1] / tot,
            100 * sum[2] / tot, 100 * sum[3] / tot, 100 * sum[4] / tot);
  }
#endif
  return PANN_OK;
}

}  // namespace pann
