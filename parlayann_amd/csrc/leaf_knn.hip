// leaf_knn.hip -- all-pairs + per-row m nearest inside many small segments (HCNNG leaves, hcnng_index.h:145-181) for the
// ONE-BYTE element types (uint8 / int8; north_star: packed integer dot products, no MFMA), rows of at most 256 bytes, m <= 16.
//
// Every LANE owns whole A rows: lane l of a wave keeps two rows of the leaf (l and l + 64 of the wave's 128) entirely in
// registers (<= 16 chunks of 16 bytes each) together with their sorted best-m lists (64-bit (dist,id) keys) -- the m-nearest
// of a row never leaves its lane, nothing is transposed through LDS.  The leaf's rows stream past as B rows: a workgroup (two
// waves = 256 A rows) stages 64 of them in LDS per step; a B chunk is ONE broadcast ds_read_b128 that feeds 8 v_dot4 (2 rows x
// 4 dwords), so the loop is bound by the dot-product issue rate, not by LDS (the round-1 kernel read 18 chunks per 128 v_dot4
// and re-staged per 64 A rows behind two barriers per tile: 5-8 % of the v_dot4 peak).
//
// Top-m: a distance that beats its row's current m-th best is queued (8-entry per-row queue in LDS, private to the lane); the
// wave empties all queues together -- one compare-exchange chain pass per queue slot -- when some queue is full, so the
// divergent insert is paid once per ~100 accepted candidates of the wave instead of once per candidate.
// Integer arithmetic: sums are exact in int32 (euclidian_point.h:54-62,74-81; mips_point.h:43-57), one cast to float at the
// end, ties broken by id through the key -- results are bit-identical to dense_topk_kernel's.
#include <vector>

#include "pann_device.h"

namespace pann {

constexpr int LK_ROWS = 256;    // A rows per workgroup (2 waves x 2 rows per lane)
constexpr int LK_TB = 64;       // B rows staged per step
constexpr int LK_PD = 8;        // pending-queue depth per row

struct LeafArgs {
  const uint8_t* points; uint32_t pstride;
  const uint32_t* ids;                 // members of all segments, concatenated (A rows == B rows)
  const uint64_t* off;                 // [nseg + 1] segment bounds into ids
  const uint32_t* tile_seg;            // [grid] segment of each workgroup
  const uint32_t* tile_a0;             // [grid] first A row (position in ids) of each workgroup
  uint32_t m; int exclude_same_id;
  uint32_t* out_ids; float* out_dists; // [total][m]
};

template <int MC>
__device__ __forceinline__ void lk_chain_insert(uint64_t (&L)[MC], uint64_t x) {
#pragma unroll
  for (int i = 0; i < MC; i++) {
    const bool lt = x < L[i];
    const uint64_t lo = lt ? x : L[i], hi = lt ? L[i] : x;
    L[i] = lo; x = hi;
  }
}

typedef uint32_t lk_u32x4 __attribute__((ext_vector_type(4)));
// LDS byte address of a __shared__ object (the low 32 bits of its flat address are its LDS offset)
__device__ __forceinline__ uint32_t lk_lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }

template <int DT>
__device__ __forceinline__ int lk_dot4(uint32_t a, uint32_t b, int acc) {
  if constexpr (DT == PANN_U8) return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)acc, false);
  else return __builtin_amdgcn_sdot4((int)a, (int)b, acc, false);
}
template <int DT>
__device__ __forceinline__ int lk_dot16(const uint4& a, const uint4& b, int acc) {
  acc = lk_dot4<DT>(a.x, b.x, acc); acc = lk_dot4<DT>(a.y, b.y, acc);
  acc = lk_dot4<DT>(a.z, b.z, acc); acc = lk_dot4<DT>(a.w, b.w, acc);
  return acc;
}

template <int DT, int METRIC, int NCH, int MC>
__global__ void __launch_bounds__(128) leaf_knn_kernel(LeafArgs A) {
  __shared__ uint4 Bt[LK_TB * NCH];
  __shared__ uint32_t Bid[LK_TB];
  __shared__ int Bnn[LK_TB];
  __shared__ uint64_t Pq[128 * 2 * LK_PD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t seg = A.tile_seg[blockIdx.x];
  const uint64_t lo = A.off[seg], hi = A.off[seg + 1];
  const uint64_t a0 = A.tile_a0[blockIdx.x];

  // ---- this lane's two A rows: registers for the whole kernel ----
  uint4 a[2][NCH];
  uint32_t aid[2]; bool valid[2]; int aa[2];
  uint64_t L[2][MC], tau[2]; uint32_t npend[2];
  float taud[2];                       // key_dist(tau): the m-th best distance so far (+inf until the list is full)
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint64_t ar = a0 + (uint64_t)wave * 128 + r * 64 + lane;
    valid[r] = ar < hi;
    aid[r] = valid[r] ? A.ids[ar] : SENTINEL;
    const uint8_t* rp = A.points + (uint64_t)(valid[r] ? aid[r] : 0u) * A.pstride;
    aa[r] = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      a[r][c] = valid[r] ? *reinterpret_cast<const uint4*>(rp + c * 16) : make_uint4(0, 0, 0, 0);
      if constexpr (METRIC == PANN_L2) aa[r] = lk_dot16<DT>(a[r][c], a[r][c], aa[r]);
    }
#pragma unroll
    for (int i = 0; i < MC; i++) L[r][i] = (i < MC - (int)A.m) ? 0ull : KEY_INF;     // leading key-0 entries are never displaced
    tau[r] = KEY_INF; npend[r] = 0; taud[r] = __builtin_inff();
  }
  uint64_t* myq = Pq + (size_t)tid * 2 * LK_PD;

  auto flush = [&]() {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      for (uint32_t s = 0; s < LK_PD; s++) {
        if (!__any(s < npend[r])) break;
        if (s < npend[r]) {
          const uint64_t x = myq[r * LK_PD + s];
          if (x < tau[r]) { lk_chain_insert<MC>(L[r], x); tau[r] = L[r][MC - 1]; }
        }
      }
      taud[r] = (tau[r] == KEY_INF) ? __builtin_inff() : key_dist(tau[r]);
      npend[r] = 0;
    }
  };

  for (uint64_t bt = lo; bt < hi; bt += LK_TB) {
    const uint32_t nb = (uint32_t)min((uint64_t)LK_TB, hi - bt);
    __syncthreads();                                   // the previous tile's readers are done
    for (int i = tid; i < LK_TB * NCH; i += 128) {     // consecutive threads read consecutive chunks of a row
      const int row = i / NCH, c = i % NCH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < (int)nb) v = *reinterpret_cast<const uint4*>(A.points + (uint64_t)A.ids[bt + row] * A.pstride + c * 16);
      Bt[row * NCH + c] = v;
    }
    if (tid < LK_TB) Bid[tid] = tid < (int)nb ? A.ids[bt + tid] : SENTINEL;
    __syncthreads();
    if constexpr (METRIC == PANN_L2) {
      if (tid < LK_TB) {
        int bb = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++) { const uint4 v = Bt[tid * NCH + c]; bb = lk_dot16<DT>(v, v, bb); }
        Bnn[tid] = bb;
      }
      __syncthreads();
    }
    // B row j's chunks sit in b[]; each register is refilled with the NEXT row's chunk right after its last use, so the
    // broadcast reads of row j + 1 are in flight during the dot products of row j (one row of software pipelining in the
    // registers the row occupies anyway).  The reads and their waits are issued by hand: LDS returns in order, and the
    // compiler's own s_waitcnt placement for such loop-carried reads drains all but the two youngest before every chunk
    // (measured: 42 % of the wave cycles waiting).  Here every use waits with lgkmcnt(LK_WAIT = LK_INFLIGHT - 1): the wanted read is
    // the oldest of the LK_INFLIGHT that are outstanding (NCH chunks + the id word [+ the norm word]), a full row old.
    constexpr int LK_INFLIGHT = NCH + (METRIC == PANN_L2 ? 2 : 1);
    constexpr int LK_WAIT = LK_INFLIGHT - 1 <= 15 ? LK_INFLIGHT - 1 : 15;      // lgkmcnt is a 4-bit counter; a smaller count only waits for more
    lk_u32x4 b[NCH];
    uint32_t bid_next, bnn_next = 0;
    {
      const uint32_t base = lk_lds_addr(Bt);
#pragma unroll
      for (int c = 0; c < NCH; c++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[c]) : "v"(base), "n"(c * 16));
      asm volatile("ds_read_b32 %0, %1" : "=v"(bid_next) : "v"(lk_lds_addr(Bid)));
      if constexpr (METRIC == PANN_L2) asm volatile("ds_read_b32 %0, %1" : "=v"(bnn_next) : "v"(lk_lds_addr(Bnn)));
    }
    for (uint32_t j = 0; j < nb; j++) {
      // two accumulators per A row (even / odd dwords): four independent v_dot4 chains per wave
      int acc0[2] = {0, 0}, acc1[2] = {0, 0};
      const uint32_t jn = min(j + 1, (uint32_t)LK_TB - 1);
      const uint32_t nxt = lk_lds_addr(Bt) + jn * (NCH * 16);
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[c]) : "n"(LK_WAIT));
        acc0[0] = lk_dot4<DT>(a[0][c].x, b[c].x, acc0[0]); acc1[0] = lk_dot4<DT>(a[1][c].x, b[c].x, acc1[0]);
        acc0[1] = lk_dot4<DT>(a[0][c].y, b[c].y, acc0[1]); acc1[1] = lk_dot4<DT>(a[1][c].y, b[c].y, acc1[1]);
        acc0[0] = lk_dot4<DT>(a[0][c].z, b[c].z, acc0[0]); acc1[0] = lk_dot4<DT>(a[1][c].z, b[c].z, acc1[0]);
        acc0[1] = lk_dot4<DT>(a[0][c].w, b[c].w, acc0[1]); acc1[1] = lk_dot4<DT>(a[1][c].w, b[c].w, acc1[1]);
        // the refill is ordered behind the sums just updated (they are inputs of the asm), so it reuses the registers of b[c]
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[c]) : "v"(nxt), "n"(c * 16), "v"(acc0[0]), "v"(acc0[1]), "v"(acc1[0]), "v"(acc1[1]));
      }
      uint32_t bid = bid_next; int bnn = (int)bnn_next;
      if constexpr (METRIC == PANN_L2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bid), "+v"(bnn) : "n"(LK_WAIT));
      else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bid) : "n"(LK_WAIT));
      asm volatile("ds_read_b32 %0, %1" : "=v"(bid_next) : "v"(lk_lds_addr(Bid) + jn * 4), "v"(bid));
      if constexpr (METRIC == PANN_L2) asm volatile("ds_read_b32 %0, %1" : "=v"(bnn_next) : "v"(lk_lds_addr(Bnn) + jn * 4), "v"(bnn));
      const int acc[2] = {acc0[0] + acc0[1], acc1[0] + acc1[1]};
#pragma unroll
      for (int r = 0; r < 2; r++) {
        // cheap test first: one conversion and one float compare against the row's current m-th best distance (a tie
        // still passes: the id decides below); the key is only built for the few that get through
        float dist;
        if constexpr (METRIC == PANN_L2) dist = (float)(aa[r] + bnn - 2 * acc[r]);
        else dist = -(float)acc[r];
        if (valid[r] && dist <= taud[r]) {
          const uint64_t key = make_key(dist, bid);
          if (key < tau[r] && !(A.exclude_same_id && bid == aid[r])) { myq[r * LK_PD + npend[r]] = key; npend[r]++; }
        }
      }
      if (__any(npend[0] == LK_PD || npend[1] == LK_PD)) flush();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the clamped reads past the tile's last row have landed
  }
  flush();
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if (!valid[r]) continue;
    const uint64_t ar = a0 + (uint64_t)wave * 128 + r * 64 + lane;
#pragma unroll
    for (int i = 0; i < MC; i++) {
      const int o = i - (MC - (int)A.m);
      if (o < 0) continue;
      const uint64_t k = L[r][i];
      A.out_ids[ar * A.m + o] = (k == KEY_INF) ? SENTINEL : key_id(k);
      A.out_dists[ar * A.m + o] = (k == KEY_INF) ? __builtin_inff() : key_dist(k);
    }
  }
}

bool leaf_knn_rows_eligible(const DeviceIndex& ix, uint32_t m) {
  static const bool off = ab_env("PANN_LEAF_OLD") != nullptr;                 // diagnostic A/B switch: the round-1 kernel
  return !off && (ix.dtype == PANN_U8 || ix.dtype == PANN_I8) && ix.pstride <= 256 && m >= 1 && m <= 16;
}

// d_ids / d_off: device; h_off: the same bounds on the host (nseg + 1) for the tile list
int leaf_knn_rows_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint32_t* d_ids, const uint64_t* d_off,
                      const uint64_t* h_off, uint64_t nseg, uint32_t m, int exclude_same, uint32_t* d_out_ids, float* d_out_dists) {
  std::vector<uint32_t> tseg, ta0;
  for (uint64_t s = 0; s < nseg; s++)
    for (uint64_t a = h_off[s]; a < h_off[s + 1]; a += LK_ROWS) { tseg.push_back((uint32_t)s); ta0.push_back((uint32_t)a); }
  if (tseg.empty()) return PANN_OK;
  const size_t nt = tseg.size();
  if (int rc = ws.ensure(nt * 8 + 256)) return rc;
  uint32_t* d_tseg = (uint32_t*)ws.buf; uint32_t* d_ta0 = d_tseg + nt;
  PANN_HIP(hipMemcpyAsync(d_tseg, tseg.data(), nt * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(d_ta0, ta0.data(), nt * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipStreamSynchronize(st));                                          // the host vectors go out of scope
  LeafArgs A{};
  A.points = ix.points; A.pstride = ix.pstride; A.ids = d_ids; A.off = d_off; A.tile_seg = d_tseg; A.tile_a0 = d_ta0;
  A.m = m; A.exclude_same_id = exclude_same; A.out_ids = d_out_ids; A.out_dists = d_out_dists;
  const uint32_t nch = (ix.dbytes + 15) / 16;                                  // chunks that hold data (the rest is zero padding)
  // every instantiation reads NCH * 16 bytes of a row: the NCH chosen below must not exceed the row stride (64-byte granules:
  // 4 -> 64, 8 -> 128, 12 -> 192, 13 and 16 -> 256), or the sums would take in the head of the next row
  if (ix.pstride % 64 != 0 || nch * 16 > ix.pstride) return PANN_ERR_BAD_ARG;
#define LK_LAUNCH(DT, MT, NCH, MC) hipLaunchKernelGGL((leaf_knn_kernel<DT, MT, NCH, MC>), dim3((uint32_t)nt), dim3(128), 0, st, A)
#define LK_NCH(DT, MT, MC)                                  \
  do {                                                      \
    if (nch <= 4) LK_LAUNCH(DT, MT, 4, MC);                 \
    else if (nch <= 8) LK_LAUNCH(DT, MT, 8, MC);            \
    else if (nch <= 12) LK_LAUNCH(DT, MT, 12, MC);          \
    else if (nch <= 13) LK_LAUNCH(DT, MT, 13, MC);          \
    else LK_LAUNCH(DT, MT, 16, MC);                         \
  } while (0)
#define LK_MC(DT, MT) do { if (m <= 10) LK_NCH(DT, MT, 10); else LK_NCH(DT, MT, 16); } while (0)
  if (ix.dtype == PANN_U8 && ix.metric == PANN_L2) LK_MC(PANN_U8, PANN_L2);
  else if (ix.dtype == PANN_U8) LK_MC(PANN_U8, PANN_MIPS);
  else if (ix.metric == PANN_L2) LK_MC(PANN_I8, PANN_L2);
  else LK_MC(PANN_I8, PANN_MIPS);
#undef LK_MC
#undef LK_NCH
#undef LK_LAUNCH
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

}  // namespace pann
