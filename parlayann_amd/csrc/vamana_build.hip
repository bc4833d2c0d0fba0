// vamana_build.hip -- robustPrune and Vamana batch_insert on gfx950.
//
//   knn_index::robustPrune   vamana/index.h:63-137   -> prune_keys_kernel + segmented sort + prune_greedy_kernel
//   knn_index::batch_insert  vamana/index.h:188-316  -> insert_batch(): search (beam_search.hip) -> prune ->
//                                                       row write-back -> reverse edges (device radix sort
//                                                       instead of parlay::group_by_key) -> append / re-prune
//   knn_index::build_index   vamana/index.h:150-186  -> pann_vamana_build (batch schedule + final sort)
//
// One wavefront per owner vertex.  Candidate (dist,id) keys live in HBM scratch: a prune touches
// ~200 candidates and its working set stays in L2, the vectors it gathers are the HBM traffic.
// The alpha test `alpha * d(p*,p') <= d(p,p')` is evaluated in double exactly as the reference
// does (:111).
#include <cstring>
#include <chrono>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "pann_device.h"

#ifndef PANN_PRUNE_MINWAVES
#define PANN_PRUNE_MINWAVES 6   /* the greedy prune is latency-bound: the f16 / bf16 variants take 104 VGPRs (4 waves per SIMD) unless told otherwise, 77 with this */
#endif
#ifndef PANN_PRUNE_PB
#define PANN_PRUNE_PB 4   /* speculative picks per pass of the greedy prune */
#endif

namespace pann {

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

struct PruneArgs {
  PointsView pv; uint32_t dbytes;
  const uint32_t* graph; uint32_t gstride; uint32_t max_deg;
  const uint32_t* owners;        // [m]
  const uint32_t* cand_ids;      // candidate ids, addressed by cand_base[i] + j
  const float* cand_dists;       // or null: distances are computed (:124-137) and counted
  const uint64_t* cand_base;     // [m]
  const uint32_t* cand_cnt;      // [m]
  const uint32_t* seg_begin;     // [m] first key slot of owner i
  uint32_t* seg_end;             // [m] out: one past the last key written
  uint64_t* keys;                // key scratch
  uint32_t* dcmps;               // [m] distance_comps (accumulated)
  int add_out_nbrs;
  uint32_t m;
  const uint32_t* order;         // launch slot -> owner index (or null): owners that run side by side then share candidate rows
};

// candidates := given (id,dist) list  U  {(G[p][i], d(G[p][i], p))}   (:70-77)
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) prune_keys_kernel(PruneArgs A) {
  const int lane = threadIdx.x;
  const uint32_t oi = A.order ? A.order[blockIdx.x] : blockIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  const uint32_t p = A.owners[oi];
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(A.pv.points + (uint64_t)p * A.pv.pstride, A.dbytes, A.pv.nch, qreg, qlds, lane);
  __syncthreads();
  const uint64_t cb = A.cand_base[oi];
  const uint32_t cn = A.cand_cnt[oi];
  uint64_t* K = A.keys + A.seg_begin[oi];
  uint32_t w = 0, dc = 0;
  if (A.cand_dists) {
    for (uint32_t j = lane; j < cn; j += PANN_WAVE) K[j] = make_key(A.cand_dists[cb + j], A.cand_ids[cb + j]);
    w = cn;
  } else {
    for (uint32_t j0 = 0; j0 < cn; j0 += PANN_WAVE) {
      const uint32_t mm = min(cn - j0, (uint32_t)PANN_WAVE);
      if (lane < mm) Pl[lane] = A.cand_ids[cb + j0 + lane];
      __syncthreads();
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, mm, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) K[j0 + ci] = make_key(dist, id); });
      __syncthreads();
    }
    w = cn; dc = cn;
  }
  if (A.add_out_nbrs) {
    const uint32_t* row = A.graph + (size_t)p * A.gstride;
    for (uint32_t i0 = 0; i0 < A.gstride; i0 += PANN_WAVE) {
      const uint32_t i = i0 + lane;
      const uint32_t a = i < A.gstride ? row[i] : SENTINEL;
      const uint64_t am = __ballot(a != SENTINEL);
      if (am == 0ull) break;
      const uint32_t mm = __popcll(am);     // neighbours are packed at the front of the row
      if (a != SENTINEL) Pl[lane] = a;
      __syncthreads();
      gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, mm, lane,
        [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) K[w + ci] = make_key(dist, id); });
      __syncthreads();
      w += mm; dc += mm;
    }
  }
  if (lane == 0) { A.seg_end[oi] = A.seg_begin[oi] + w; A.dcmps[oi] += dc; }
}

struct GreedyArgs {
  PointsView pv; uint32_t dbytes;
  const uint32_t* owners;
  const uint32_t* seg_begin; const uint32_t* seg_end;
  uint64_t* keys;            // sorted by (dist,id) per segment
  double alpha; uint32_t R;
  uint32_t* rows_out; uint32_t rows_stride;  // [m x rows_stride] ids, SENTINEL padded (or null)
  uint32_t* cnt_out;                          // [m] (or null)
  uint32_t* graph; uint32_t gstride;          // direct write of the owner's row when rows_out == null
  uint16_t* gcode; const uint16_t* rank16;    // ... with the filter codes of the new row (filter_codes.hip), or null
  const uint32_t* order;                      // launch slot -> owner index (or null)
  uint32_t* dcmps;
  uint32_t m;
  uint32_t kcap;                              // lists up to this many keys are pruned in LDS
  uint32_t single_pick;                       // diagnostic (PANN_PRUNE_SINGLE): bit 0 one pick per pass, bit 1 lists in HBM
};

__device__ __forceinline__ uint64_t ld_key(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
}
__device__ __forceinline__ void st_key(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, PANN_PRIVATE_SCOPE);
}

// greedy alpha-prune over the sorted, de-duplicated candidate list (:90-116).  One wave per owner.  The list
// (<= A.kcap keys: visited + out-neighbours of one insert) lives in LDS for the whole loop: the sequential
// walk over picks and the kill flags then cost no HBM round trip; per pick only the pick's vector and the
// gathered vectors of the live candidates are fetched.  Longer lists (heavy re-prunes) walk the HBM copy.
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE, PANN_PRUNE_MINWAVES) prune_greedy_kernel(GreedyArgs A) {
  const int lane = threadIdx.x;
  const uint32_t oi = A.order ? A.order[blockIdx.x] : blockIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];    // live candidate ids of the current tile
  __shared__ uint32_t Pp[PANN_WAVE];    // their positions in the segment
  __shared__ float Pd[PANN_WAVE];       // their distance to p  (dist_pprime)
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  uint32_t* Out = reinterpret_cast<uint32_t*>(qlds + (NCH1 ? 0 : PANN_PRUNE_PB * A.pv.nch * LPC)); // [R rounded to 4] selected neighbours (after the PB query slots)
  uint64_t* Ks = reinterpret_cast<uint64_t*>(Out + ((A.R + 3) & ~3u));                   // [kcap] the list, when it fits
  const uint32_t p = A.owners[oi];
  uint64_t* K = A.keys + A.seg_begin[oi];
  const uint32_t n = A.seg_end[oi] - A.seg_begin[oi];
  const bool in_lds = n <= A.kcap && !(A.single_pick & 2u);
  auto ldk = [&](uint32_t i) -> uint64_t { return in_lds ? Ks[i] : ld_key(K + i); };
  auto stk = [&](uint32_t i, uint64_t v) { if (in_lds) Ks[i] = v; else st_key(K + i, v); };

  // std::unique by id (:86-88): after the sort equal ids are adjacent (same id => same key)
  uint32_t carry = SENTINEL;   // original id of the last entry of the previous tile
  for (uint32_t i0 = 0; i0 < n; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    uint64_t k = i < n ? K[i] : KEY_INF;
    const uint32_t myid = key_id(k);
    uint32_t previd = __shfl_up(myid, 1);
    if (lane == 0) previd = carry;
    carry = __shfl(myid, PANN_WAVE - 1);
    if (i < n && myid == previd) k = (k & 0xFFFFFFFF00000000ull) | SENTINEL;
    if (i < n) stk(i, k);
  }
  if (!in_lds) __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  uint32_t nsel = 0, dc = 0;
  // Several picks per pass.  The next PB live entries are taken as speculative picks: their vectors are loaded
  // together and every live candidate vector is fetched once and scored against all of them; the sequential
  // semantics (a pick killed by an earlier pick of the same pass is not a pick; a candidate is counted for and
  // killed by the selected picks in order) are then resolved from those distances -- a quarter of the dependent
  // memory round trips of one-pick-at-a-time.  (Lists longer than kcap are walked in HBM with the same logic; the
  // exact-float-order mode keeps the plain loop.)
  constexpr int PB = PANN_PRUNE_PB;
  __shared__ float Dd[PB][PANN_WAVE];
  const uint32_t qstride4 = NCH1 ? 0u : A.pv.nch * LPC;
  if (!A.pv.exact && !(A.single_pick & 1u)) {
    uint32_t idx = 0;
    while (idx < n && nsel < A.R) {
      const uint32_t wi = idx + lane;
      const uint32_t idw = wi < n ? key_id(ldk(wi)) : SENTINEL;
      uint64_t pm = __ballot(idw != SENTINEL && idw != p);                 // :99
      if (pm == 0ull) { idx += PANN_WAVE; continue; }
      uint32_t ppos[PB], pid[PB];
      int nb = 0;
      const int room = (int)min((uint32_t)PB, A.R - nsel);
#pragma unroll
      for (int b = 0; b < PB; b++) {
        ppos[b] = 0; pid[b] = 0;
        if (b < room && pm) {
          const int L = __ffsll((unsigned long long)pm) - 1;
          pm &= pm - 1;
          ppos[b] = idx + (uint32_t)L; pid[b] = (uint32_t)__builtin_amdgcn_readlane((int)idw, L); nb = b + 1;
        }
      }
      QReg<DT> qreg[PB];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < PB; b++) {
        const uint32_t src = pid[b < nb ? b : 0];                            // unused slots re-read pick 0 (uniform code)
        load_query<DT, LPC, NCH1>(A.pv.points + (uint64_t)src * A.pv.pstride, A.dbytes, A.pv.nch, qreg[b], qlds + b * qstride4, lane);
      }
      __syncthreads();
      bool sel[PB], kills[PB];
#pragma unroll
      for (int b = 0; b < PB; b++) { sel[b] = false; kills[b] = false; }
      bool first_tile = true;
      const uint32_t nsel0 = nsel;
      for (uint32_t t0 = ppos[0] + 1; t0 < n; t0 += PANN_WAVE) {             // :105-115
        const uint32_t i = t0 + lane;
        const uint64_t k = i < n ? ldk(i) : KEY_INF;
        const bool live = (i < n) && (key_id(k) != SENTINEL);
        const uint64_t lm = __ballot(live);
        const uint32_t mm = __popcll(lm);
        if (mm == 0) continue;
        if (live) { const uint32_t at = lanes_below(lm, lane); Pl[at] = key_id(k); Pp[at] = i; Pd[at] = key_dist(k); }
        __syncthreads();
        gather_tile_multi<DT, METRIC, LPC, NCH1, PB>(A.pv, qreg, qlds, qstride4, Pl, mm, lane,
          [&](bool has, uint32_t ci, uint32_t, const float (&d)[PB]) {
            if (has) {
#pragma unroll
              for (int b = 0; b < PB; b++) Dd[b][ci] = d[b];
            }
          });
        __syncthreads();
        const bool me = lane < (int)mm;
        const uint32_t mypos = me ? Pp[lane] : 0u;
        const float dpp = me ? Pd[lane] : 0.0f;
        float dl[PB];
#pragma unroll
        for (int b = 0; b < PB; b++) dl[b] = Dd[b][lane];
        if (first_tile) {   // which of the speculative picks survive the earlier ones; all of them sit in this tile
          first_tile = false;
          sel[0] = true; nsel++; kills[0] = nsel < A.R;                      // the R-th pick runs no inner loop
          bool stop = nsel == A.R;
#pragma unroll
          for (int j = 1; j < PB; j++) {
            if (j < nb && !stop) {
              const int Lj = __ffsll((unsigned long long)__ballot(me && mypos == ppos[j])) - 1;
              const float dppj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dpp), Lj));
              bool dead = false;
#pragma unroll
              for (int i2 = 0; i2 < j; i2++) {
                if (kills[i2]) {
                  const float dij = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dl[i2]), Lj));
                  dead = dead || (A.alpha * (double)dij <= (double)dppj);   // :111
                }
              }
              if (!dead) { sel[j] = true; nsel++; kills[j] = nsel < A.R; stop = nsel == A.R; }
            }
          }
        }
        bool alive = me;
#pragma unroll
        for (int j = 0; j < PB; j++) {
          if (kills[j]) {
            const bool elig = alive && mypos > ppos[j];
            dc += (uint32_t)__popcll(__ballot(elig));                        // distance_comps of pick j's inner loop
            if (elig && A.alpha * (double)dl[j] <= (double)dpp) alive = false;
          }
        }
        if (me && !alive) stk(mypos, ((uint64_t)f2ord(dpp) << 32) | SENTINEL);   // candidates[i].first = -1
        __syncthreads();
      }
      if (first_tile) { sel[0] = true; nsel++; }                             // nothing live after the pick
      if (lane == 0) {
        uint32_t at = nsel0;
#pragma unroll
        for (int j = 0; j < PB; j++) if (sel[j]) Out[at++] = pid[j];         // :103
      }
      idx = ppos[nb - 1] + 1;
      if (!in_lds) __builtin_amdgcn_s_waitcnt(0);                              // kill flags of this pass have landed
    }
  } else {
    for (uint32_t idx = 0; idx < n && nsel < A.R; idx++) {
      const uint32_t ps = key_id(ldk(idx));
      if (ps == p || ps == SENTINEL) continue;        // :99
      if (lane == 0) Out[nsel] = ps;                  // :103
      nsel++;
      if (nsel == A.R) break;                          // the inner loop's kills can no longer matter
      QReg<DT> qreg{};
      __syncthreads();
      load_query<DT, LPC, NCH1>(A.pv.points + (uint64_t)ps * A.pv.pstride, A.dbytes, A.pv.nch, qreg, qlds, lane);
      __syncthreads();
      for (uint32_t t0 = idx + 1; t0 < n; t0 += PANN_WAVE) {   // :105-115, 64 candidates at a time
        const uint32_t i = t0 + lane;
        uint64_t k = KEY_INF;
        if (i < n) k = ldk(i);
        const bool live = (i < n) && (key_id(k) != SENTINEL);
        const uint64_t lm = __ballot(live);
        const uint32_t mm = __popcll(lm);
        if (mm == 0) continue;
        if (live) { const uint32_t at = lanes_below(lm, lane); Pl[at] = key_id(k); Pp[at] = i; Pd[at] = key_dist(k); }
        dc += mm;
        __syncthreads();
        gather_tile<DT, METRIC, LPC, NCH1, 4>(A.pv, qreg, qlds, Pl, mm, lane,
          [&](bool has, uint32_t ci, uint32_t, float d_sp) {
            if (has) {
              const float d_pp = Pd[ci];
              if (A.alpha * (double)d_sp <= (double)d_pp)       // :111
                stk(Pp[ci], ((uint64_t)f2ord(d_pp) << 32) | SENTINEL);   // candidates[i].first = -1
            }
          });
        __syncthreads();
      }
      if (!in_lds) __builtin_amdgcn_s_waitcnt(0);
    }
  }
  __syncthreads();
  if (in_lds) for (uint32_t i = lane; i < n; i += PANN_WAVE) K[i] = Ks[i];     // the tail counter reads the kill flags
  if (A.rows_out) {
    for (uint32_t j = lane; j < A.rows_stride; j += PANN_WAVE)
      A.rows_out[(size_t)oi * A.rows_stride + j] = j < nsel ? Out[j] : SENTINEL;
    if (lane == 0 && A.cnt_out) A.cnt_out[oi] = nsel;
  } else {
    uint32_t* row = A.graph + (size_t)p * A.gstride;
    for (uint32_t j = lane; j < A.gstride; j += PANN_WAVE) row[j] = j < nsel ? Out[j] : SENTINEL;
    if (A.gcode) {
      uint16_t* crow = A.gcode + (size_t)p * A.gstride;
      for (uint32_t j = lane; j < A.gstride; j += PANN_WAVE) crow[j] = j < nsel ? A.rank16[Out[j]] : (uint16_t)0xFFFF;
    }
  }
  if (lane == 0) A.dcmps[oi] += dc;
}

// note on `if (nsel == R) break`: the reference runs the inner loop for the R-th pick too and counts
// those distance_comps; parity of the COUNTER is kept by the variant below (count_last = true).
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) prune_count_tail_kernel(GreedyArgs A) {
  // distance_comps of the last pick's inner loop == number of live candidates after it
  const int lane = threadIdx.x;
  const uint32_t oi = A.order ? A.order[blockIdx.x] : blockIdx.x;
  const uint32_t p = A.owners[oi];
  const uint64_t* K = A.keys + A.seg_begin[oi];
  const uint32_t n = A.seg_end[oi] - A.seg_begin[oi];
  // find the position of the R-th selected element: replay the selection order
  uint32_t nsel = 0, pos = n;
  for (uint32_t i0 = 0; i0 < n && pos == n; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    uint32_t id = SENTINEL;
    if (i < n) id = key_id(K[i]);
    const uint64_t sm = __ballot(i < n && id != SENTINEL && id != p);
    const uint32_t c = __popcll(sm);
    if (nsel + c >= A.R) {
      // the (R - nsel)-th set bit of sm
      uint64_t t = sm; uint32_t need = A.R - nsel;
      int b = -1;
      while (need) { b = __ffsll((unsigned long long)t) - 1; t &= t - 1; need--; }
      pos = i0 + (uint32_t)b;
    }
    nsel += c;
  }
  if (pos == n) return;     // fewer than R picks: the main kernel counted everything
  uint32_t live = 0;
  for (uint32_t i0 = pos + 1; i0 < n; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    live += __popcll(__ballot(i < n && key_id(K[i]) != SENTINEL));
  }
  if (lane == 0) A.dcmps[oi] += live;
}

// launch order of a batch's searches: slots sorted by (locality cell of the inserted point, position in the batch)
__global__ void order_keys_kernel(const uint32_t* batch, const uint32_t* cell, uint32_t m, uint64_t* keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) keys[i] = ((uint64_t)cell[batch[i]] << 32) | i;
}
__global__ void order_from_keys_kernel(const uint64_t* keys, uint32_t m, uint32_t* order) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) order[i] = (uint32_t)keys[i];
}

__global__ void fixed_stride_setup_kernel(uint64_t* cand_base, uint32_t* seg_begin, uint32_t m,
                                          uint32_t cand_stride, uint32_t seg_stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { cand_base[i] = (uint64_t)i * cand_stride; seg_begin[i] = i * seg_stride; }
}

// (gcode / rank16: the filter codes of the new rows are written with them, filter_codes.hip; null: not maintained)
__global__ void scatter_rows_kernel(uint32_t* graph, uint32_t gstride, const uint32_t* owners,
                                    const uint32_t* rows, uint32_t rows_stride, uint32_t m, uint16_t* gcode, const uint16_t* rank16) {
  const uint32_t oi = blockIdx.x;
  uint32_t* row = graph + (size_t)owners[oi] * gstride;
  uint16_t* crow = gcode ? gcode + (size_t)owners[oi] * gstride : nullptr;
  for (uint32_t j = threadIdx.x; j < gstride; j += blockDim.x) {
    const uint32_t a = j < rows_stride ? rows[(size_t)oi * rows_stride + j] : SENTINEL;
    row[j] = a;
    if (crow) crow[j] = a == SENTINEL ? (uint16_t)0xFFFF : rank16[a];
  }
}

// reverse edges (:278-281): key = (target << 32) | position of the source in the batch
__global__ void edge_keys_kernel(const uint32_t* rows, uint32_t rows_stride, uint32_t m, uint64_t* ekeys) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= (uint64_t)m * rows_stride) return;
  const uint32_t v = rows[t];
  ekeys[t] = (v == SENTINEL) ? KEY_INF : (((uint64_t)v << 32) | (uint32_t)(t / rows_stride));
}

__global__ void edge_heads_kernel(const uint64_t* ekeys, uint64_t total, uint32_t* heads, uint32_t* nvalid) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint64_t k = ekeys[t];
  const bool valid = k != KEY_INF;
  heads[t] = (valid && (t == 0 || (uint32_t)(ekeys[t - 1] >> 32) != (uint32_t)(k >> 32))) ? 1u : 0u;
  if (valid && (t + 1 == total || ekeys[t + 1] == KEY_INF)) *nvalid = (uint32_t)(t + 1);
}

__global__ void group_starts_kernel(const uint32_t* heads, const uint32_t* gidx_excl, uint64_t total,
                                    uint32_t* gstart) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= total) return;
  if (heads[t]) gstart[gidx_excl[t]] = (uint32_t)t;
}

struct ReverseArgs {
  uint32_t* graph; uint32_t gstride; uint32_t R;
  const uint64_t* ekeys; const uint32_t* gstart; const uint32_t* ngroups; const uint32_t* nvalid;
  const uint32_t* batch;        // position -> vertex id
  uint32_t* edge_src;           // [total] source vertex of each sorted edge
  uint32_t* heavy_owner; uint64_t* heavy_base; uint32_t* heavy_cnt; uint32_t* heavy_len;
  uint32_t* nheavy;
  uint16_t* gcode; const uint16_t* rank16;   // filter codes kept in step with the rows (filter_codes.hip), or null
};

// per target vertex: append-without-repeats when the row stays within R (:292-294), else queue the
// vertex for a re-prune (:296-298)
__global__ void __launch_bounds__(PANN_WAVE) reverse_light_kernel(ReverseArgs A) {
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  const uint32_t ng = *A.ngroups;
  if (g >= ng) return;
  __shared__ uint32_t Cn[1024];
  __shared__ uint16_t Cc[1024];                 // the codes of Cn (only when A.gcode)
  const uint32_t lo = A.gstart[g];
  const uint32_t hi = (g + 1 < ng) ? A.gstart[g + 1] : *A.nvalid;
  const uint32_t c = hi - lo;
  const uint32_t v = (uint32_t)(A.ekeys[lo] >> 32);
  uint32_t* row = A.graph + (size_t)v * A.gstride;
  uint16_t* crow = A.gcode ? A.gcode + (size_t)v * A.gstride : nullptr;
  uint32_t deg = 0;
  for (uint32_t i0 = 0; i0 < A.gstride; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    deg += __popcll(__ballot(i < A.gstride && row[i] != SENTINEL));
  }
  for (uint32_t j = lane; j < c; j += PANN_WAVE) {
    const uint32_t src = A.batch[(uint32_t)A.ekeys[lo + j]];
    A.edge_src[lo + j] = src;
    if (c + deg <= A.R) { Cn[j] = src; if (crow) Cc[j] = A.rank16[src]; }
  }
  if (c + deg > A.R) {
    if (lane == 0) {
      const uint32_t h = atomicAdd(A.nheavy, 1u);
      A.heavy_owner[h] = v; A.heavy_base[h] = lo; A.heavy_cnt[h] = c; A.heavy_len[h] = c + deg;
    }
    return;
  }
  __syncthreads();
  // new row = candidates ++ (old neighbours not among the candidates)
  uint32_t w = c;
  for (uint32_t i0 = 0; i0 < A.gstride; i0 += PANN_WAVE) {      // deg <= R <= gstride
    const uint32_t i = i0 + lane;
    const uint32_t a = (i < deg) ? row[i] : SENTINEL;
    const uint16_t ac = (crow && i < deg) ? crow[i] : (uint16_t)0xFFFF;      // a kept neighbour keeps its code
    bool keep = (i < deg);
    if (keep) for (uint32_t j = 0; j < c; j++) keep &= (Cn[j] != a);
    const uint64_t km = __ballot(keep);
    __syncthreads();
    if (keep) { const uint32_t at = w + lanes_below(km, lane); Cn[at] = a; Cc[at] = ac; }
    w += __popcll(km);
  }
  __syncthreads();
  for (uint32_t j = lane; j < A.gstride; j += PANN_WAVE) row[j] = j < w ? Cn[j] : SENTINEL;
  if (crow) for (uint32_t j = lane; j < A.gstride; j += PANN_WAVE) crow[j] = j < w ? Cc[j] : (uint16_t)0xFFFF;
}

// G[i].sort by distance to i (:180-185); ties by id
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) sort_rows_kernel(PointsView pv, uint32_t dbytes, uint32_t* graph,
                                                              uint32_t gstride, uint32_t n) {
  const int lane = threadIdx.x;
  const uint32_t v = blockIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  __shared__ uint64_t Kk[4096 + 64];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(pv.points + (uint64_t)v * pv.pstride, dbytes, pv.nch, qreg, qlds, lane);
  __syncthreads();
  uint32_t* row = graph + (size_t)v * gstride;
  uint32_t deg = 0;
  for (uint32_t i0 = 0; i0 < gstride; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    const uint32_t a = i < gstride ? row[i] : SENTINEL;
    const uint64_t am = __ballot(a != SENTINEL);
    if (am == 0ull) break;
    const uint32_t mm = __popcll(am);
    if (a != SENTINEL) Pl[lane] = a;
    __syncthreads();
    gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
      [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) Kk[deg + ci] = make_key(dist, id); });
    __syncthreads();
    deg += mm;
  }
  for (uint32_t j0 = 0; j0 < deg; j0 += PANN_WAVE) {   // rank sort; keys are distinct unless an id repeats
    const uint32_t j = j0 + lane;
    if (j < deg) {
      const uint64_t k = Kk[j];
      uint32_t r = 0;
      for (uint32_t i = 0; i < deg; i++) { const uint64_t o = Kk[i]; r += (o < k || (o == k && i < j)) ? 1u : 0u; }
      row[r] = key_id(k);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------

struct Bump {
  uint8_t* base; size_t cap; size_t off = 0; bool dry;
  Bump(void* b, size_t c) : base((uint8_t*)b), cap(c), dry(b == nullptr) {}
  template <typename T> T* take(size_t count) {
    off = (off + 255) / 256 * 256;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

static size_t seg_sort_temp_bytes(uint32_t size, uint32_t segs) {
  size_t t = 0;
  (void)rocprim::segmented_radix_sort_keys(nullptr, t, (uint64_t*)nullptr, (uint64_t*)nullptr, size, segs,
                                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, 64, (hipStream_t)0);
  return t;
}
static size_t sort_temp_bytes(uint32_t size) {
  size_t t = 0;
  (void)rocprim::radix_sort_keys(nullptr, t, (uint64_t*)nullptr, (uint64_t*)nullptr, size, 0, 64, (hipStream_t)0);
  return t;
}
static size_t scan_temp_bytes(uint32_t size) {
  size_t t = 0;
  (void)rocprim::exclusive_scan(nullptr, t, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, size,
                                rocprim::plus<uint32_t>(), (hipStream_t)0);
  return t;
}


// keys -> segmented sort -> greedy.  keys_a holds the unsorted keys, keys_b receives the sorted ones.
static int run_prune(const DeviceIndex& ix, PruneArgs pa, GreedyArgs ga, uint64_t* keys_a, uint64_t* keys_b,
                     uint32_t total_keys, void* sort_tmp, size_t sort_tmp_bytes, hipStream_t st, uint32_t max_seg_len) {
  const uint32_t m = pa.m;
  if (m == 0) return PANN_OK;
  const size_t qb = query_lds_bytes(ix);
  pa.keys = keys_a;
#define CALL_KEYS(DT, MT, L, N1) hipLaunchKernelGGL((prune_keys_kernel<DT, MT, L, N1>), dim3(m), dim3(PANN_WAVE), qb, st, pa)
  PANN_TYPE_SWITCH(ix, CALL_KEYS);
#undef CALL_KEYS
  PANN_HIP(hipGetLastError());
  PANN_HIP(rocprim::segmented_radix_sort_keys(sort_tmp, sort_tmp_bytes, keys_a, keys_b, total_keys, m,
                                              (const uint32_t*)pa.seg_begin, (const uint32_t*)pa.seg_end, 0, 64, st));
  ga.keys = keys_b; ga.seg_begin = pa.seg_begin; ga.seg_end = pa.seg_end;
  { static const char* sp = ab_env("PANN_PRUNE_SINGLE"); ga.single_pick = sp ? (uint32_t)atoi(sp) : 0u; }   // A/B switch: 1 one pick per pass, 2 lists in HBM
  ga.kcap = std::min<uint32_t>((max_seg_len + 63) / 64 * 64, 3072);                       // <= 24 KB of keys per wave
  const size_t gb_lds = qb * PANN_PRUNE_PB + (size_t)((ga.R + 3) & ~3u) * 4 + (size_t)ga.kcap * 8;      // PB query slots
#define CALL_GREEDY(DT, MT, L, N1) hipLaunchKernelGGL((prune_greedy_kernel<DT, MT, L, N1>), dim3(m), dim3(PANN_WAVE), gb_lds, st, ga)
  PANN_TYPE_SWITCH(ix, CALL_GREEDY);
#undef CALL_GREEDY
  PANN_HIP(hipGetLastError());
#define CALL_TAIL(DT, MT, L, N1) hipLaunchKernelGGL((prune_count_tail_kernel<DT, MT, L, N1>), dim3(m), dim3(PANN_WAVE), 0, st, ga)
  PANN_TYPE_SWITCH(ix, CALL_TAIL);
#undef CALL_TAIL
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

// Host-pointer robustPrune batch (the C-ABI entry): stage, run, fetch.
int robust_prune_batch_host(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint32_t* owners,
                            uint64_t m, const uint32_t* cand_ids, const float* cand_dists,
                            const uint64_t* cand_offsets, double alpha, uint32_t R, int add_out_nbrs,
                            uint32_t* out_rows, uint32_t* out_dist_cmps) {
  if (m == 0) return PANN_OK;
  if (R == 0 || R > 1024) { set_error("pann_robust_prune_batch: R out of range [1,1024]"); return PANN_ERR_BAD_ARG; }
  const uint64_t ncand = cand_offsets[m];
  std::vector<uint64_t> h_base(m); std::vector<uint32_t> h_cnt(m), h_seg(m);
  uint64_t total = 0, max_seg = 0;
  for (uint64_t i = 0; i < m; i++) {
    if (owners[i] >= ix.n) { set_error("pann_robust_prune_batch: owner out of range"); return PANN_ERR_BAD_ARG; }
    h_base[i] = cand_offsets[i]; h_cnt[i] = (uint32_t)(cand_offsets[i + 1] - cand_offsets[i]);
    h_seg[i] = (uint32_t)total;
    total += h_cnt[i] + (add_out_nbrs ? ix.gstride : 0);
    max_seg = std::max<uint64_t>(max_seg, h_cnt[i] + (add_out_nbrs ? ix.gstride : 0));
  }
  for (uint64_t j = 0; j < ncand; j++)
    if (cand_ids[j] >= ix.n) { set_error("pann_robust_prune_batch: candidate id out of range"); return PANN_ERR_BAD_ARG; }
  if (total >= 0xFFFFFFF0ull) { set_error("pann_robust_prune_batch: too many candidates in one call"); return PANN_ERR_BAD_ARG; }
  const size_t stmp = seg_sort_temp_bytes((uint32_t)total, (uint32_t)m);
  auto layout = [&](Bump& b, uint32_t*& d_own, uint32_t*& d_cid, float*& d_cd, uint64_t*& d_base, uint32_t*& d_cnt,
                    uint32_t*& d_seg, uint32_t*& d_send, uint64_t*& ka, uint64_t*& kb, uint32_t*& d_rows,
                    uint32_t*& d_rcnt, uint32_t*& d_dc, void*& d_tmp) {
    d_own = b.take<uint32_t>(m); d_cid = b.take<uint32_t>(ncand + 1); d_cd = b.take<float>(ncand + 1);
    d_base = b.take<uint64_t>(m); d_cnt = b.take<uint32_t>(m); d_seg = b.take<uint32_t>(m); d_send = b.take<uint32_t>(m);
    ka = b.take<uint64_t>(total + 1); kb = b.take<uint64_t>(total + 1);
    d_rows = b.take<uint32_t>(m * R); d_rcnt = b.take<uint32_t>(m); d_dc = b.take<uint32_t>(m);
    d_tmp = b.take<uint8_t>(stmp + 16);
  };
  uint32_t *d_own, *d_cid, *d_cnt, *d_seg, *d_send, *d_rows, *d_rcnt, *d_dc; float* d_cd; uint64_t *d_base, *ka, *kb; void* d_tmp;
  Bump dry(nullptr, 0);
  layout(dry, d_own, d_cid, d_cd, d_base, d_cnt, d_seg, d_send, ka, kb, d_rows, d_rcnt, d_dc, d_tmp);
  if (int rc = ws.ensure(dry.off + 4096)) return rc;
  Bump b(ws.buf, ws.bytes);
  layout(b, d_own, d_cid, d_cd, d_base, d_cnt, d_seg, d_send, ka, kb, d_rows, d_rcnt, d_dc, d_tmp);
  PANN_HIP(hipMemcpyAsync(d_own, owners, m * 4, hipMemcpyHostToDevice, st));
  if (ncand) PANN_HIP(hipMemcpyAsync(d_cid, cand_ids, ncand * 4, hipMemcpyHostToDevice, st));
  if (cand_dists && ncand) PANN_HIP(hipMemcpyAsync(d_cd, cand_dists, ncand * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(d_base, h_base.data(), m * 8, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(d_cnt, h_cnt.data(), m * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(d_seg, h_seg.data(), m * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemsetAsync(d_dc, 0, m * 4, st));
  PruneArgs pa{};
  pa.pv = PointsView{ix.points, ix.pstride, ix.nch, ix.exact}; pa.dbytes = ix.dbytes;
  pa.graph = ix.graph; pa.gstride = ix.gstride; pa.max_deg = ix.max_deg;
  pa.owners = d_own; pa.cand_ids = d_cid; pa.cand_dists = cand_dists ? d_cd : nullptr;
  pa.cand_base = d_base; pa.cand_cnt = d_cnt; pa.seg_begin = d_seg; pa.seg_end = d_send;
  pa.dcmps = d_dc; pa.add_out_nbrs = add_out_nbrs; pa.m = (uint32_t)m;
  GreedyArgs ga{};
  ga.pv = pa.pv; ga.dbytes = ix.dbytes; ga.owners = d_own; ga.alpha = alpha; ga.R = R;
  ga.rows_out = d_rows; ga.rows_stride = R; ga.cnt_out = d_rcnt; ga.graph = nullptr; ga.gstride = ix.gstride;
  ga.dcmps = d_dc; ga.m = (uint32_t)m;
  if (int rc = run_prune(ix, pa, ga, ka, kb, (uint32_t)total, d_tmp, stmp, st, (uint32_t)std::min<uint64_t>(max_seg, 1u << 30))) return rc;
  std::vector<uint32_t> h_rows(m * R), h_rcnt(m);
  PANN_HIP(hipMemcpyAsync(h_rows.data(), d_rows, m * R * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipMemcpyAsync(h_rcnt.data(), d_rcnt, m * 4, hipMemcpyDeviceToHost, st));
  if (out_dist_cmps) PANN_HIP(hipMemcpyAsync(out_dist_cmps, d_dc, m * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  for (uint64_t i = 0; i < m; i++) {   // reference row layout: slot 0 = count (graph.h:84-99)
    uint32_t* row = out_rows + i * (uint64_t)(R + 1);
    row[0] = h_rcnt[i];
    for (uint32_t j = 0; j < R; j++) row[1 + j] = j < h_rcnt[i] ? h_rows[i * R + j] : 0u;
  }
  return PANN_OK;
}

// One batch of batch_insert (:242-300) on the device, in two phases so that the batch can be split over GPUs
// (parlayann_amd/distributed.py):
//   phase A  vamana_search_prune_dev : beam search from `start` + robustPrune of the visited list for the batch points
//            handed in (:247-266) -- reads the graph only, so any subset of the batch can run anywhere; result = their new
//            out-neighbour rows, m x R, SENTINEL padded
//   phase B  vamana_apply_rows_dev   : write the rows of the WHOLE batch (:268-270), then the reverse edges grouped by
//            target, append-or-re-prune (:278-300) -- a deterministic function of (graph, batch ids, rows)
// insert_batch_dev = A then B on one device.
namespace {
auto now_ = [] { return std::chrono::steady_clock::now(); };
template <class A, class B> double secs_(A a, B b) { return std::chrono::duration<double>(b - a).count(); }
}  // namespace

int vamana_search_prune_dev(const DeviceIndex& ix, Workspace& ws, Workspace& search_ws, hipStream_t st, const uint32_t* d_batch,
                            uint32_t m, uint32_t start, uint32_t R, uint32_t L, double alpha, uint32_t* vcap_io,
                            uint32_t* d_rows, pann_build_stats* stats) {
  if (m == 0) return PANN_OK;
  if (R == 0 || R > ix.max_deg || R > 1024) { set_error("vamana insert: R must be in [1, min(max_deg,1024)]"); return PANN_ERR_BAD_ARG; }
  const auto t0 = now_();
  uint32_t vcap = *vcap_io;
  for (int attempt = 0;; attempt++) {
    const uint32_t seg_stride = vcap + ix.gstride;
    const uint64_t total_keys = (uint64_t)m * seg_stride;
    if (total_keys >= 0xFFFFFFF0ull) { set_error("vamana insert: batch too large"); return PANN_ERR_BAD_ARG; }
    const bool ordered = ix.cell != nullptr && m >= ix.cell_min_batch;          // launch the searches in locality order (api.hip: ensure_locality_cells)
    const size_t stmp = std::max(seg_sort_temp_bytes((uint32_t)total_keys, m), ordered ? sort_temp_bytes(m) : (size_t)0);
    uint32_t *d_start, *d_vis_ids, *d_vis_cnt, *d_dcs, *d_seg, *d_send, *d_rcnt, *d_dc, *d_order;
    float* d_vis_d; uint64_t *d_base, *ka, *kb; void* d_tmp;
    auto layout = [&](Bump& b) {
      d_start = b.take<uint32_t>(4);
      d_order = b.take<uint32_t>(ordered ? m : 1);
      d_vis_ids = b.take<uint32_t>((size_t)m * vcap); d_vis_d = b.take<float>((size_t)m * vcap);
      d_vis_cnt = b.take<uint32_t>(m); d_dcs = b.take<uint32_t>(m);
      d_base = b.take<uint64_t>(m); d_seg = b.take<uint32_t>(m); d_send = b.take<uint32_t>(m);
      ka = b.take<uint64_t>(total_keys + 1); kb = b.take<uint64_t>(total_keys + 1);
      d_rcnt = b.take<uint32_t>(m); d_dc = b.take<uint32_t>(m);
      d_tmp = b.take<uint8_t>(stmp + 16);
    };
    Bump dry(nullptr, 0); layout(dry);
    if (int rc = ws.ensure(dry.off + 4096)) return rc;
    Bump b(ws.buf, ws.bytes); layout(b);

    // ---- 1. beam search from `start` for every batch point (:247-259) ----
    PANN_HIP(hipMemcpyAsync(d_start, &start, 4, hipMemcpyHostToDevice, st));
    SearchArgs sa{};
    sa.queries = nullptr; sa.qstride = 0; sa.query_ids = d_batch; sa.nq = m; sa.starts = d_start; sa.nstarts = 1;
    sa.k = 0; sa.beam = L; sa.limit = (int64_t)ix.n; sa.degree_limit = ix.max_deg; sa.cut = 0.0;   // :250
    sa.out = pann_search_out{};
    sa.out.visited_ids = d_vis_ids; sa.out.visited_dists = d_vis_d; sa.out.visited_cap = vcap;
    sa.out.visited_count = d_vis_cnt; sa.out.dist_cmps = d_dcs;
    if (ordered) {      // (ka / kb / d_tmp are the prune's scratch: free until the searches are done)
      hipLaunchKernelGGL(order_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d_batch, ix.cell, m, ka);
      size_t tb = stmp + 16;
      PANN_HIP(rocprim::radix_sort_keys(d_tmp, tb, ka, kb, m, 0, 64, st));
      hipLaunchKernelGGL(order_from_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, st, kb, m, d_order);
      PANN_HIP(hipGetLastError());
      sa.order = d_order;
    }
    if (int rc = search_ws.ensure(search_workspace_bytes(ix, sa))) return rc;
    if (int rc = launch_beam_search(ix, sa, search_ws.buf, search_ws.bytes, st)) return rc;
    uint32_t status = 0;
    PANN_HIP(hipMemcpyAsync(&status, (uint8_t*)search_ws.buf + 64, 4, hipMemcpyDeviceToHost, st));
    PANN_HIP(hipStreamSynchronize(st));
    if (status & 1u) {   // a visited list did not fit: grow and redo the searches (nothing written so far)
      if (attempt >= 6) { set_error("vamana insert: visited lists keep overflowing"); return PANN_ERR_OVERFLOW; }
      vcap *= 2; *vcap_io = vcap; continue;
    }
    const auto t1 = now_();

    // ---- 2. robustPrune(index, visited) for every batch point (:264) ----
    hipLaunchKernelGGL(fixed_stride_setup_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d_base, d_seg, m, vcap, seg_stride);
    PANN_HIP(hipMemsetAsync(d_dc, 0, (size_t)m * 4, st));
    PruneArgs pa{};
    pa.pv = PointsView{ix.points, ix.pstride, ix.nch, ix.exact}; pa.dbytes = ix.dbytes;
    pa.graph = ix.graph; pa.gstride = ix.gstride; pa.max_deg = ix.max_deg;
    pa.owners = d_batch; pa.cand_ids = d_vis_ids; pa.cand_dists = d_vis_d; pa.cand_base = d_base; pa.cand_cnt = d_vis_cnt;
    pa.seg_begin = d_seg; pa.seg_end = d_send; pa.dcmps = d_dc; pa.add_out_nbrs = 1; pa.m = m;
    pa.order = ordered ? d_order : nullptr;
    GreedyArgs ga{};
    ga.pv = pa.pv; ga.dbytes = ix.dbytes; ga.owners = d_batch; ga.alpha = alpha; ga.R = R;
    ga.rows_out = d_rows; ga.rows_stride = R; ga.cnt_out = d_rcnt; ga.graph = nullptr; ga.gstride = ix.gstride;
    ga.dcmps = d_dc; ga.m = m; ga.order = pa.order;
    if (int rc = run_prune(ix, pa, ga, ka, kb, (uint32_t)total_keys, d_tmp, stmp, st, seg_stride)) return rc;
    PANN_HIP(hipStreamSynchronize(st));
    const auto t2 = now_();
    if (stats) {
      std::vector<uint32_t> hs(m), hp(m), hv(m);
      PANN_HIP(hipMemcpy(hs.data(), d_dcs, (size_t)m * 4, hipMemcpyDeviceToHost));
      PANN_HIP(hipMemcpy(hp.data(), d_dc, (size_t)m * 4, hipMemcpyDeviceToHost));
      PANN_HIP(hipMemcpy(hv.data(), d_vis_cnt, (size_t)m * 4, hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < m; i++) { stats->search_dist_cmps += hs[i]; stats->prune_dist_cmps += hp[i]; stats->visited_total += hv[i]; }
      if (stats->per_point_visited || stats->per_point_dist_cmps) {                                // vamana/index.h:261-266
        std::vector<uint32_t> hb(m);
        PANN_HIP(hipMemcpy(hb.data(), d_batch, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < m; i++) {
          if (stats->per_point_visited) stats->per_point_visited[hb[i]] += hv[i];
          if (stats->per_point_dist_cmps) stats->per_point_dist_cmps[hb[i]] += hs[i] + hp[i];
        }
      }
      stats->t_search_s += secs_(t0, t1); stats->t_prune_s += secs_(t1, t2);
    }
    return PANN_OK;
  }
}

int vamana_apply_rows_dev(const DeviceIndex& ix, Workspace& ws, Workspace& ws2, hipStream_t st, const uint32_t* d_batch, uint32_t m,
                          const uint32_t* d_rows, uint32_t R, double alpha, pann_build_stats* stats) {
  if (m == 0) return PANN_OK;
  if (R == 0 || R > ix.max_deg || R > 1024) { set_error("vamana insert: R must be in [1, min(max_deg,1024)]"); return PANN_ERR_BAD_ARG; }
  const auto t2 = now_();
  const uint64_t total_edges = (uint64_t)m * R;
  if (total_edges >= 0xFFFFFFF0ull) { set_error("vamana insert: batch too large"); return PANN_ERR_BAD_ARG; }
  const size_t stmp = std::max(sort_temp_bytes((uint32_t)total_edges), scan_temp_bytes((uint32_t)total_edges));
  uint32_t *d_heads, *d_gidx, *d_gstart, *d_src, *d_hown, *d_hcnt, *d_hlen, *d_scalars;
  uint64_t *ek_a, *ek_b, *d_hbase; void* d_tmp;
  auto layout = [&](Bump& b) {
    d_scalars = b.take<uint32_t>(64);   // [1] nvalid, [2] ngroups, [3] nheavy
    ek_a = b.take<uint64_t>(total_edges + 1); ek_b = b.take<uint64_t>(total_edges + 1);
    d_heads = b.take<uint32_t>(total_edges + 1); d_gidx = b.take<uint32_t>(total_edges + 1);
    d_gstart = b.take<uint32_t>(total_edges + 1); d_src = b.take<uint32_t>(total_edges + 1);
    d_hown = b.take<uint32_t>(total_edges + 1); d_hbase = b.take<uint64_t>(total_edges + 1);
    d_hcnt = b.take<uint32_t>(total_edges + 1); d_hlen = b.take<uint32_t>(total_edges + 1);
    d_tmp = b.take<uint8_t>(stmp + 16);
  };
  Bump dry(nullptr, 0); layout(dry);
  if (int rc = ws.ensure(dry.off + 4096)) return rc;
  Bump b(ws.buf, ws.bytes); layout(b);

  // ---- :268-270 write the new out-neighbourhoods (only now: every search and prune of the batch saw the old graph)
  uint16_t* const gcode = ix.codes_valid ? ix.gcode : nullptr;      // filter codes maintained with the rows (filter_codes.hip)
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(m), dim3(PANN_WAVE), 0, st, ix.graph, ix.gstride, d_batch, d_rows, R, m, gcode, ix.rank16);
  PANN_HIP(hipGetLastError());

  // ---- 3. reverse edges grouped by target (:278-282) ----
  const uint32_t te = (uint32_t)total_edges;
  hipLaunchKernelGGL(edge_keys_kernel, dim3((te + 255) / 256), dim3(256), 0, st, d_rows, R, m, ek_a);
  size_t tb = stmp;
  PANN_HIP(rocprim::radix_sort_keys(d_tmp, tb, ek_a, ek_b, te, 0, 64, st));
  PANN_HIP(hipMemsetAsync(d_scalars, 0, 256, st));
  hipLaunchKernelGGL(edge_heads_kernel, dim3((te + 255) / 256), dim3(256), 0, st, ek_b, (uint64_t)te, d_heads, d_scalars + 1);
  tb = stmp;
  PANN_HIP(rocprim::exclusive_scan(d_tmp, tb, d_heads, d_gidx, 0u, te, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(group_starts_kernel, dim3((te + 255) / 256), dim3(256), 0, st, d_heads, d_gidx, (uint64_t)te, d_gstart);
  uint32_t h_last[2];
  PANN_HIP(hipMemcpyAsync(&h_last[0], d_gidx + (te - 1), 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipMemcpyAsync(&h_last[1], d_heads + (te - 1), 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  const uint32_t ngroups = h_last[0] + h_last[1];
  PANN_HIP(hipMemcpyAsync(d_scalars + 2, &ngroups, 4, hipMemcpyHostToDevice, st));
  const auto t3 = now_();

  // ---- 4. append or re-prune per target (:289-300) ----
  uint32_t nheavy = 0;
  if (ngroups) {
    ReverseArgs ra{};
    ra.graph = ix.graph; ra.gstride = ix.gstride; ra.R = R; ra.ekeys = ek_b; ra.gstart = d_gstart;
    ra.ngroups = d_scalars + 2; ra.nvalid = d_scalars + 1; ra.batch = d_batch; ra.edge_src = d_src;
    ra.heavy_owner = d_hown; ra.heavy_base = d_hbase; ra.heavy_cnt = d_hcnt; ra.heavy_len = d_hlen; ra.nheavy = d_scalars + 3;
    ra.gcode = gcode; ra.rank16 = ix.rank16;
    hipLaunchKernelGGL(reverse_light_kernel, dim3(ngroups), dim3(PANN_WAVE), 0, st, ra);
    PANN_HIP(hipGetLastError());
    PANN_HIP(hipMemcpyAsync(&nheavy, d_scalars + 3, 4, hipMemcpyDeviceToHost, st));
    PANN_HIP(hipStreamSynchronize(st));
  }
  uint64_t reprune_dc = 0;
  if (nheavy) {
    // the heavy vertices were queued through an atomic counter, i.e. in no fixed order; every re-prune reads only its own
    // vertex's row and candidates, so the graph does not depend on that order
    std::vector<uint32_t> h_len(nheavy), h_seg(nheavy);
    PANN_HIP(hipMemcpy(h_len.data(), d_hlen, (size_t)nheavy * 4, hipMemcpyDeviceToHost));
    uint64_t tk = 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < nheavy; i++) { h_seg[i] = (uint32_t)tk; tk += h_len[i]; max_len = std::max(max_len, h_len[i]); }
    if (tk >= 0xFFFFFFF0ull) { set_error("vamana insert: re-prune too large"); return PANN_ERR_BAD_ARG; }
    const size_t stmp2 = seg_sort_temp_bytes((uint32_t)tk, nheavy);
    uint32_t *h_dseg, *h_dsend, *h_ddc; uint64_t *hk_a, *hk_b; void* h_tmp;
    auto layout2 = [&](Bump& bb) {
      h_dseg = bb.take<uint32_t>(nheavy); h_dsend = bb.take<uint32_t>(nheavy); h_ddc = bb.take<uint32_t>(nheavy);
      hk_a = bb.take<uint64_t>(tk + 1); hk_b = bb.take<uint64_t>(tk + 1); h_tmp = bb.take<uint8_t>(stmp2 + 16);
    };
    Bump dry2(nullptr, 0); layout2(dry2);
    if (int rc = ws2.ensure(dry2.off + 4096)) return rc;
    Bump b2(ws2.buf, ws2.bytes); layout2(b2);
    PANN_HIP(hipMemcpyAsync(h_dseg, h_seg.data(), (size_t)nheavy * 4, hipMemcpyHostToDevice, st));
    PANN_HIP(hipMemsetAsync(h_ddc, 0, (size_t)nheavy * 4, st));
    PruneArgs pb{};
    pb.pv = PointsView{ix.points, ix.pstride, ix.nch, ix.exact}; pb.dbytes = ix.dbytes; pb.graph = ix.graph; pb.gstride = ix.gstride; pb.max_deg = ix.max_deg;
    pb.owners = d_hown; pb.cand_ids = d_src; pb.cand_dists = nullptr; pb.cand_base = d_hbase; pb.cand_cnt = d_hcnt;
    pb.seg_begin = h_dseg; pb.seg_end = h_dsend; pb.dcmps = h_ddc; pb.add_out_nbrs = 1; pb.m = nheavy;
    GreedyArgs gb{};
    gb.pv = pb.pv; gb.dbytes = ix.dbytes; gb.owners = d_hown; gb.alpha = alpha; gb.R = R;
    gb.rows_out = nullptr; gb.rows_stride = 0; gb.cnt_out = nullptr; gb.graph = ix.graph; gb.gstride = ix.gstride;
    gb.gcode = gcode; gb.rank16 = ix.rank16;
    gb.dcmps = h_ddc; gb.m = nheavy;
    if (int rc = run_prune(ix, pb, gb, hk_a, hk_b, (uint32_t)tk, h_tmp, stmp2, st, max_len)) return rc;
    if (stats) {
      std::vector<uint32_t> hd(nheavy), ho(stats->per_point_dist_cmps ? nheavy : 0);
      PANN_HIP(hipMemcpyAsync(hd.data(), h_ddc, (size_t)nheavy * 4, hipMemcpyDeviceToHost, st));
      if (!ho.empty()) PANN_HIP(hipMemcpyAsync(ho.data(), d_hown, (size_t)nheavy * 4, hipMemcpyDeviceToHost, st));
      PANN_HIP(hipStreamSynchronize(st));
      for (uint32_t v : hd) reprune_dc += v;
      for (size_t i = 0; i < ho.size(); i++) stats->per_point_dist_cmps[ho[i]] += hd[i];      // vamana/index.h:298
    }
  }
  PANN_HIP(hipStreamSynchronize(st));
  const auto t4 = now_();
  if (stats) {
    stats->prune_dist_cmps += reprune_dc;
    stats->t_bidirect_s += secs_(t2, t3); stats->t_reprune_s += secs_(t3, t4);
  }
  return PANN_OK;
}

int insert_batch_dev(const DeviceIndex& ix, Workspace& ws, Workspace& ws2, Workspace& search_ws, Workspace& rows_ws, hipStream_t st,
                     const uint32_t* d_batch, uint32_t m, uint32_t start, uint32_t R, uint32_t L, double alpha,
                     uint32_t* vcap_io, pann_build_stats* stats) {
  if (m == 0) return PANN_OK;
  if (int rc = rows_ws.ensure((size_t)m * R * 4 + 256)) return rc;
  uint32_t* d_rows = (uint32_t*)rows_ws.buf;
  if (int rc = vamana_search_prune_dev(ix, ws, search_ws, st, d_batch, m, start, R, L, alpha, vcap_io, d_rows, stats)) return rc;
  return vamana_apply_rows_dev(ix, ws, ws2, st, d_batch, m, d_rows, R, alpha, stats);
}

int sort_neighbors_dev(const DeviceIndex& ix, hipStream_t st) {
  if (ix.max_deg > 4096) { set_error("sort_neighbors: max_deg > 4096"); return PANN_ERR_UNSUPPORTED; }
  const PointsView pv{ix.points, ix.pstride, ix.nch, ix.exact};
  const size_t qb = query_lds_bytes(ix);
#define CALL_SORT(DT, MT, L, N1) hipLaunchKernelGGL((sort_rows_kernel<DT, MT, L, N1>), dim3((uint32_t)ix.n), dim3(PANN_WAVE), qb, st, pv, ix.dbytes, ix.graph, ix.gstride, (uint32_t)ix.n)
  PANN_TYPE_SWITCH(ix, CALL_SORT);
#undef CALL_SORT
  PANN_HIP(hipGetLastError());
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

}  // namespace pann
