// dense.hip -- dense all-pairs distances with a per-row top-m, on gfx950.
//
//   hcnng_index::MSTk  HCNNG/hcnng_index.h:145-181  all N*(N-1) leaf distances, 10 smallest per row
//   compute_groundtruth  data_tools/compute_groundtruth.cpp:22-59  exact kNN of every query
//
// Tiling: a 256-thread workgroup owns 64 A-rows (16 per wave) and streams B in tiles of 128 rows (64 in
// the MFMA variant);
// both tiles are staged in LDS in 256-byte dimension segments (B padded by 16 B per row so the
// lane-per-row ds_read_b128 is conflict free, A read as broadcasts).  Lane l of a wave holds the
// running distances of B-row l to the wave's 16 A-rows in registers, so a B tile is read from HBM
// once per 64 A-rows.  The per-row top-m lists live in LDS and are touched only when a distance
// beats the row's current m-th best (rare after the first tiles).
//
// Arithmetic is the same dist_accum as the gather kernels (exact for integer types; for float
// types one lane sums a whole row left to right in 16-byte chunks).
#include <vector>

#include "pann_device.h"

namespace pann {

constexpr int DT_A = 64;        // A rows per workgroup
constexpr int DT_AW = 16;       // A rows per wave
constexpr int DT_B = 64;        // B rows per lane group (one per lane)
constexpr int DT_RB = 2;        // VALU kernel: B rows per lane (tile = 128 rows); halves the LDS broadcasts per FMA
constexpr int DT_SEG = 256;     // bytes of the dimension staged per step
constexpr int DT_BSTRIDE = DT_SEG + 16;

struct DenseArgs {
  const uint8_t* points; uint32_t pstride; uint32_t dbytes;
  // A rows: external vectors (a_ext, stride) or point ids (a_ids) -- per segment ranges below
  const uint8_t* a_ext; uint64_t a_stride; const uint32_t* a_ids;
  // B rows: point ids (b_ids) or the contiguous range of all points
  const uint32_t* b_ids;
  // segments (leaves): A rows [a_off[s], a_off[s+1]) against B rows [b_off[s], b_off[s+1]);
  const uint64_t* a_off; const uint64_t* b_off;   // device arrays, nseg+1 (null: one segment 0..na / 0..nb)
  uint64_t na, nb;
  uint32_t nsplit;            // B range of a segment is cut into nsplit pieces (grid.y)
  uint32_t m, mcap;           // top-m; mcap = m rounded up to 16
  int exclude_same_id;        // leaf mode: skip j == i (hcnng_index.h:153)
  uint32_t exact;             // exact float order: one sequential accumulator per (A row, B row)
  uint64_t* partial;          // [total A rows][nsplit][m] keys
  const uint32_t* tile_seg;   // [grid.x] segment of each A tile
  const uint32_t* tile_a0;    // [grid.x] first A row (global index) of each tile
};


// sorted insert of key x into list[0..mcap) (ascending, KEY_INF padded), dropping the last entry
__device__ __forceinline__ void list_insert(uint64_t* list, uint32_t mcap, uint64_t x, int lane) {
  uint64_t nv[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t i = lane + r * 64;
    nv[r] = KEY_INF;
    if (i < mcap) {
      const uint64_t cur = list[i];
      const bool prev_lt = (i == 0) ? true : (list[i - 1] < x);
      nv[r] = (cur < x) ? cur : (prev_lt ? x : list[i - 1]);
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t i = lane + r * 64;
    if (i < mcap) list[i] = nv[r];
  }
  wave_lds_sync();
}

// ---- lane-parallel top-m for m <= 16 (the HCNNG leaf case, m = 10) ----
// The one-insert-at-a-time list above costs ~40 instructions per improving candidate with the whole wave
// working on ONE row; at m = 10, N = 1000 that was 65 % of the leaf kernel.  Here the wave's 16 A rows are
// updated together: lane (row = lane & 15, quarter = lane >> 4) keeps its own sorted list of the best keys
// among the columns {quarter*32 .. quarter*32+31} of every B tile in REGISTERS (16 entries; the unused
// leading 16-m hold key 0, which no real key displaces, so the m-th best is always entry 15).  Per tile the
// distances go through LDS (transposed: K[row][column]), each lane marks the entries below its threshold in
// a 32-bit mask and the marked ones are inserted by an unrolled compare-exchange chain; the four quarter
// lists of a row are merged by rank once, after the last tile.
constexpr int TM_KSTRIDE = 133;   // words per K row: 4 quarters x 33 (bank spread)

__device__ __forceinline__ void tm_chain_insert(uint64_t (&L)[16], uint64_t x) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const bool lt = x < L[i];
    const uint64_t lo = lt ? x : L[i], hi = lt ? L[i] : x;
    L[i] = lo; x = hi;
  }
}

// first index in sorted list[0..16) with list[i] >= x
__device__ __forceinline__ uint32_t tm_lower_bound16(const uint64_t* list, uint64_t x) {
  uint32_t lo = 0;
#pragma unroll
  for (int step = 8; step >= 1; step >>= 1) lo += (list[lo + step - 1] < x) ? step : 0;
  lo += (list[lo] < x) ? 1 : 0;      // lo <= 15 here
  return lo;
}

template <int DT, int METRIC>
__global__ void __launch_bounds__(256) dense_topk_kernel(DenseArgs A) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint8_t* At = smem;                                   // [64][DT_SEG]
  constexpr int TB = DT_B * DT_RB;                      // B rows per tile: DT_RB per lane
  uint8_t* Bt = At + DT_A * DT_SEG;                     // [TB][DT_BSTRIDE]
  uint32_t* Bid = reinterpret_cast<uint32_t*>(Bt + TB * DT_BSTRIDE);     // [TB] ids of the B tile
  uint32_t* Aid = Bid + TB;                             // [64] ids of the A tile (SENTINEL: none)
  uint64_t* lists = reinterpret_cast<uint64_t*>(Aid + DT_A);            // [64][mcap]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t seg = A.tile_seg ? A.tile_seg[blockIdx.x] : 0u;
  const uint64_t a_lo = A.a_off ? A.a_off[seg] : 0ull, a_hi = A.a_off ? A.a_off[seg + 1] : A.na;
  const uint64_t b_lo = A.b_off ? A.b_off[seg] : 0ull, b_hi = A.b_off ? A.b_off[seg + 1] : A.nb;
  const uint64_t a0 = A.tile_a0 ? (uint64_t)A.tile_a0[blockIdx.x] : (uint64_t)blockIdx.x * DT_A;
  const uint32_t na_tile = (uint32_t)min((uint64_t)DT_A, a_hi - a0);
  (void)a_lo;
  // this block's share of the B range
  const uint64_t nb_seg = b_hi - b_lo;
  const uint64_t per = ((nb_seg + A.nsplit - 1) / A.nsplit + TB - 1) / TB * TB;
  const uint64_t bs = b_lo + min(nb_seg, (uint64_t)blockIdx.y * per);
  const uint64_t be = b_lo + min(nb_seg, (uint64_t)(blockIdx.y + 1) * per);

  for (uint32_t i = tid; i < DT_A * A.mcap; i += 256) lists[i] = KEY_INF;
  if (tid < DT_A) Aid[tid] = (tid < (int)na_tile && A.a_ids) ? A.a_ids[a0 + tid] : SENTINEL;
  __syncthreads();
  const bool lane_lists = (A.mcap == 16);                 // m <= 16: lane-parallel top-m (uniform)
  uint64_t TL[16];
#pragma unroll
  for (int i = 0; i < 16; i++) TL[i] = (i < 16 - (int)A.m) ? 0ull : KEY_INF;
  uint32_t* Kw = reinterpret_cast<uint32_t*>(Bt) + wave * (DT_AW * TM_KSTRIDE);   // aliases the B tile between tiles

  const uint32_t nseg = (A.pstride + DT_SEG - 1) / DT_SEG;
  // stage one 256-byte segment of 64 rows: 16 lanes x 16 B per row, 16 rows per pass of 256 threads
  auto stage_rows = [&](uint8_t* dst, uint32_t dstride, uint32_t sg, auto rowptr, auto rowvalid, uint32_t nrows, int rows_total) {
    const int r0 = tid >> 4, c = tid & 15;
    for (int r = r0; r < rows_total; r += 16) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint32_t off = sg * DT_SEG + c * 16;
      if (r < (int)nrows) {
        const uint8_t* rp = rowptr(r);
        const uint32_t valid = rowvalid();
        if ((reinterpret_cast<uintptr_t>(rp) & 15) == 0) v = load16_guarded(rp, off, valid);
        else {
          uint8_t tmp[16];
#pragma unroll
          for (int i = 0; i < 16; i++) tmp[i] = (off + i < valid) ? rp[off + i] : (uint8_t)0;
          __builtin_memcpy(&v, tmp, 16);
        }
      }
      *reinterpret_cast<uint4*>(dst + (size_t)r * dstride + c * 16) = v;
    }
  };
  auto a_rowptr = [&](int r) -> const uint8_t* {
    return A.a_ids ? A.points + (uint64_t)A.a_ids[a0 + r] * A.pstride : A.a_ext + (a0 + r) * A.a_stride;
  };
  auto a_valid = [&]() -> uint32_t { return A.a_ids ? A.pstride : A.dbytes; };

  if (nseg == 1) { stage_rows(At, DT_SEG, 0, a_rowptr, a_valid, na_tile, DT_A); }

  for (uint64_t bt = bs; bt < be; bt += TB) {
    const uint32_t nb_tile = (uint32_t)min((uint64_t)TB, be - bt);
    auto b_rowptr = [&](int r) -> const uint8_t* {
      const uint64_t id = A.b_ids ? (uint64_t)A.b_ids[bt + r] : (bt + r);
      return A.points + id * A.pstride;
    };
    auto b_valid = [&]() -> uint32_t { return A.pstride; };
    Acc<DT> acc[DT_AW][DT_RB];
#pragma unroll
    for (int a = 0; a < DT_AW; a++)
#pragma unroll
      for (int rb = 0; rb < DT_RB; rb++) acc[a][rb].clear();
    __syncthreads();          // previous tile's readers are done with Bt / Bid
    if (tid < TB) Bid[tid] = tid < (int)nb_tile ? (A.b_ids ? A.b_ids[bt + tid] : (uint32_t)(bt + tid)) : SENTINEL;
    for (uint32_t sg = 0; sg < nseg; sg++) {
      if (sg > 0) __syncthreads();
      stage_rows(Bt, DT_BSTRIDE, sg, b_rowptr, b_valid, nb_tile, TB);
      if (nseg > 1) stage_rows(At, DT_SEG, sg, a_rowptr, a_valid, na_tile, DT_A);
      __syncthreads();
      // chunks that hold only the rows' zero padding (d*esize .. pstride) add nothing: skip them
      const uint32_t seg_bytes = min((uint32_t)DT_SEG, A.pstride - sg * DT_SEG);
      const uint32_t seg_valid = A.dbytes > sg * DT_SEG ? min(seg_bytes, A.dbytes - sg * DT_SEG) : 0u;
      const uint32_t nchunk = (seg_valid + 15) / 16;
      for (uint32_t c = 0; c < nchunk; c++) {
        // the lane's DT_RB B chunks are digested once (QReg); each of the 16 A chunks is ONE LDS broadcast
        // that feeds DT_RB accumulators (the LDS read rate, not the VALU, bounds this loop at DT_RB = 1)
        QReg<DT> b[DT_RB];
#pragma unroll
        for (int rb = 0; rb < DT_RB; rb++)
          b[rb] = make_qreg<DT>(*reinterpret_cast<const uint4*>(Bt + (size_t)(lane + 64 * rb) * DT_BSTRIDE + c * 16));
        if constexpr (is_float_dt<DT>()) {
          if (A.exact) {      // validation mode: strictly left-to-right, unfused (s.y stays 0)
            uint4 braw[DT_RB];
#pragma unroll
            for (int rb = 0; rb < DT_RB; rb++) braw[rb] = *reinterpret_cast<const uint4*>(Bt + (size_t)(lane + 64 * rb) * DT_BSTRIDE + c * 16);
#pragma unroll
            for (int a = 0; a < DT_AW; a++) {
              const uint4 q = *reinterpret_cast<const uint4*>(At + (size_t)(wave * DT_AW + a) * DT_SEG + c * 16);
#pragma unroll
              for (int rb = 0; rb < DT_RB; rb++) { float t = acc[a][rb].s.x; dist_accum_exact<DT, METRIC>(t, q, braw[rb]); acc[a][rb].s.x = t; }
            }
            continue;
          }
        }
#pragma unroll
        for (int a = 0; a < DT_AW; a++) {
          const uint4 q = *reinterpret_cast<const uint4*>(At + (size_t)(wave * DT_AW + a) * DT_SEG + c * 16);
#pragma unroll
          for (int rb = 0; rb < DT_RB; rb++) dist_accum<DT, METRIC>(acc[a][rb], q, b[rb]);
        }
      }
    }
    if (lane_lists) {
      __syncthreads();                                    // every wave is done reading Bt: reuse it as K
#pragma unroll
      for (int rb = 0; rb < DT_RB; rb++) {
        const uint32_t brow = lane + 64 * rb;
        const uint32_t bid = Bid[brow];
#pragma unroll
        for (int a = 0; a < DT_AW; a++) {
          const uint32_t ar = wave * DT_AW + a;
          const float dist = dist_finish<DT, METRIC>(acc_lane_value<DT, METRIC>(acc[a][rb]));
          bool ok = (brow < nb_tile) && (ar < na_tile);
          if (A.exclude_same_id) ok = ok && (bid != Aid[ar]);
          Kw[a * TM_KSTRIDE + (brow >> 5) * 33 + (brow & 31)] = ok ? f2ord(dist) : 0xFFFFFFFFu;
        }
      }
      wave_lds_sync();
      {
        const uint32_t* krow = Kw + (lane & 15) * TM_KSTRIDE + (lane >> 4) * 33;
        const uint32_t* idq = Bid + (lane >> 4) * 32;
        uint64_t tau = TL[15];
        uint32_t bits = 0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) bits |= (krow[j] <= (uint32_t)(tau >> 32) && krow[j] != 0xFFFFFFFFu) ? (1u << j) : 0u;
        while (bits) {                                    // per-lane walk over its marked entries
          const int j = __ffs(bits) - 1;
          bits &= bits - 1;
          const uint64_t x = ((uint64_t)krow[j] << 32) | idq[j];
          if (x < tau) { tm_chain_insert(TL, x); tau = TL[15]; }
        }
      }
      continue;                                           // the next tile's barrier protects K
    }
    // ---- top-m update: wave-private lists of its 16 A rows ----
#pragma unroll
    for (int rb = 0; rb < DT_RB; rb++) {
      const uint32_t brow = lane + 64 * rb;
      const uint32_t bid = Bid[brow];
#pragma unroll
      for (int a = 0; a < DT_AW; a++) {
        const uint32_t ar = wave * DT_AW + a;
        if (ar >= na_tile) break;               // uniform per wave
        uint64_t* list = lists + (size_t)ar * A.mcap;
        const float dist = dist_finish<DT, METRIC>(acc_lane_value<DT, METRIC>(acc[a][rb]));
        const uint64_t key = make_key(dist, bid);
        bool ok = (brow < nb_tile);
        if (A.exclude_same_id) ok = ok && (bid != Aid[ar]);
        uint64_t tau = list[A.m - 1];
        uint64_t mask = __ballot(ok && key < tau);
        while (mask) {
          const int L = __ffsll((unsigned long long)mask) - 1;
          const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, L);
          const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), L);
          const uint64_t x = ((uint64_t)khi << 32) | klo;
          mask &= mask - 1;
          if (x < tau) {
            list_insert(list, A.mcap, x, lane);
            tau = list[A.m - 1];
          }
        }
      }
    }
  }
  __syncthreads();
  if (lane_lists) {
    // merge the four quarter lists of every row by rank: position in the own list + entries of the other three
    // that are smaller (keys are unique: a column belongs to one quarter)
    uint64_t* Lw = reinterpret_cast<uint64_t*>(Bt) + (size_t)wave * (DT_AW * 4 * 16);       // [16 rows][4][16]
    const int row = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; i++) Lw[(row * 4 + qd) * 16 + i] = TL[i];
    const uint32_t ar = wave * DT_AW + row;
    uint64_t* out = A.partial + ((a0 + ar) * A.nsplit + blockIdx.y) * A.m;
    __syncthreads();                                      // lists visible
    const int lead = 16 - (int)A.m;                       // leading key-0 entries of every list
    if (ar < na_tile && qd == 0) {                        // slots beyond the number of real keys stay empty
      int real = 0;
#pragma unroll
      for (int o = 0; o < 4; o++) real += (int)tm_lower_bound16(Lw + (row * 4 + o) * 16, KEY_INF) - lead;
      for (int j = real; j < (int)A.m; j++) out[j] = KEY_INF;
    }
    if (ar < na_tile) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint64_t x = TL[e];
        if (e >= lead && x != KEY_INF) {
          int rank = e - lead;
#pragma unroll
          for (int o = 1; o < 4; o++) rank += (int)tm_lower_bound16(Lw + (row * 4 + ((qd + o) & 3)) * 16, x) - lead;
          if (rank < (int)A.m) out[rank] = x;
        }
      }
    }
    return;
  }
  // ---- emit this block's partial lists ----
  for (uint32_t i = tid; i < na_tile * A.m; i += 256) {
    const uint32_t ar = i / A.m, j = i % A.m;
    A.partial[((a0 + ar) * A.nsplit + blockIdx.y) * A.m + j] = lists[(size_t)ar * A.mcap + j];
  }
}

// ---- fp16 variant on the matrix cores (north_star: MFMA only for the dense fp16 contraction) ----
// Same tiling as dense_topk_kernel.  Each wave owns 16 A rows and computes the 16 x 64 block of dot
// products against the B tile with v_mfma_f32_16x16x32_f16 (A and B fragments are both "row with k
// contiguous", i.e. plain ds_read_b128 from the padded LDS tiles); distances are
//   L2 : |a|^2 + |b|^2 - 2 a.b      MIPS : -a.b
// with the row norms summed in f32 while the tiles are staged.  On integer-valued data every
// product and partial sum is exact, so the result is bit-identical to the VALU path; on real-valued
// data the norm form differs from sum((a-b)^2) by cancellation error (DESIGN.md "float order").
typedef _Float16 mf_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 mf_bf8 __attribute__((ext_vector_type(8)));
typedef float mf_float4 __attribute__((ext_vector_type(4)));

template <bool BF>
__device__ __forceinline__ float sumsq16(uint4 v) {
  float ss = 0.f;
  if constexpr (BF) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { const float lo = bf16_lo(w[i]), hi = bf16_hi(w[i]); ss = fmaf(lo, lo, ss); ss = fmaf(hi, hi, ss); }
  } else {
    mf_half8 h; __builtin_memcpy(&h, &v, 16);
#pragma unroll
    for (int i = 0; i < 8; i++) { const float f = (float)h[i]; ss = fmaf(f, f, ss); }
  }
  return ss;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector memory counter, which would wait
// for the tile requested two iterations ahead at every tile (measured: 2.6 us per tile, the loaded HBM latency)
__device__ __forceinline__ void gt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BF == false: IEEE binary16 operands (v_mfma_f32_16x16x32_f16); BF == true: bfloat16 (v_mfma_f32_16x16x32_bf16).
// Both accumulate in f32; products of two-byte values are exact in f32, so integer-valued data gives exact sums.
template <int METRIC, bool BF>
__global__ void __launch_bounds__(256) dense_topk_mfma_f16_kernel(DenseArgs A) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint8_t* At = smem;                                   // [64][DT_BSTRIDE]
  uint8_t* Bt = At + DT_A * DT_BSTRIDE;                 // [64][DT_BSTRIDE]
  uint32_t* Bid = reinterpret_cast<uint32_t*>(Bt + DT_B * DT_BSTRIDE);
  uint32_t* Aid = Bid + DT_B;
  float* An = reinterpret_cast<float*>(Aid + DT_A);     // [64] |a|^2
  float* Bn = An + DT_A;                                // [64] |b|^2
  uint64_t* lists = reinterpret_cast<uint64_t*>(Bn + DT_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t seg = A.tile_seg ? A.tile_seg[blockIdx.x] : 0u;
  const uint64_t a_hi = A.a_off ? A.a_off[seg + 1] : A.na;
  const uint64_t b_lo = A.b_off ? A.b_off[seg] : 0ull, b_hi = A.b_off ? A.b_off[seg + 1] : A.nb;
  const uint64_t a0 = A.tile_a0 ? (uint64_t)A.tile_a0[blockIdx.x] : (uint64_t)blockIdx.x * DT_A;
  const uint32_t na_tile = (uint32_t)min((uint64_t)DT_A, a_hi - a0);
  const uint64_t nb_seg = b_hi - b_lo;
  const uint64_t per = ((nb_seg + A.nsplit - 1) / A.nsplit + DT_B - 1) / DT_B * DT_B;
  const uint64_t bs = b_lo + min(nb_seg, (uint64_t)blockIdx.y * per);
  const uint64_t be = b_lo + min(nb_seg, (uint64_t)(blockIdx.y + 1) * per);

  for (uint32_t i = tid; i < DT_A * A.mcap; i += 256) lists[i] = KEY_INF;
  if (tid < DT_A) { Aid[tid] = (tid < (int)na_tile && A.a_ids) ? A.a_ids[a0 + tid] : SENTINEL; An[tid] = 0.f; }
  __syncthreads();
  // m <= 16: lane-parallel top-m as in dense_topk_kernel -- lane (row = lane & 15, quarter = lane >> 4) owns the 16
  // columns {quarter*16 ..} of every 64-row B tile; K (16 x 4 x 17 words per wave) aliases the B tile between tiles
  const bool lane_lists = (A.mcap == 16);
  constexpr int MF_KS = 68;
  uint64_t TL[16];
#pragma unroll
  for (int i = 0; i < 16; i++) TL[i] = (i < 16 - (int)A.m) ? 0ull : KEY_INF;
  uint32_t* Kw = reinterpret_cast<uint32_t*>(Bt) + wave * (DT_AW * MF_KS);
  const uint32_t nseg = (A.pstride + DT_SEG - 1) / DT_SEG;

  // stage one 256-byte segment of 64 rows (16 threads x 16 B per row) and add its squared norm
  auto stage_rows = [&](uint8_t* dst, float* norms, uint32_t sg, auto rowptr, uint32_t valid, uint32_t nrows) {
    const int r0 = tid >> 4, c = tid & 15;
    for (int r = r0; r < 64; r += 16) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint32_t off = sg * DT_SEG + c * 16;
      if (r < (int)nrows) {
        const uint8_t* rp = rowptr(r);
        if ((reinterpret_cast<uintptr_t>(rp) & 15) == 0) v = load16_guarded(rp, off, valid);
        else {
          uint8_t tmp[16];
#pragma unroll
          for (int i = 0; i < 16; i++) tmp[i] = (off + i < valid) ? rp[off + i] : (uint8_t)0;
          __builtin_memcpy(&v, tmp, 16);
        }
      }
      *reinterpret_cast<uint4*>(dst + (size_t)r * DT_BSTRIDE + c * 16) = v;
      float ss = 0.f;
      if constexpr (BF) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { const float lo = bf16_lo(w[i]), hi = bf16_hi(w[i]); ss = fmaf(lo, lo, ss); ss = fmaf(hi, hi, ss); }
      } else {
        mf_half8 h; __builtin_memcpy(&h, &v, 16);
#pragma unroll
        for (int i = 0; i < 8; i++) { const float f = (float)h[i]; ss = fmaf(f, f, ss); }
      }
      ss = group_sum<16>(ss);                       // the 16 threads of a row are 16 consecutive lanes
      if (c == 0) norms[r] += ss;
    }
  };
  auto a_rowptr = [&](int r) -> const uint8_t* {
    return A.a_ids ? A.points + (uint64_t)A.a_ids[a0 + r] * A.pstride : A.a_ext + (a0 + r) * A.a_stride;
  };
  const uint32_t a_valid = A.a_ids ? A.pstride : A.dbytes;
  if (nseg == 1) stage_rows(At, An, 0, a_rowptr, a_valid, na_tile);

  // epilogue of one tile: acc[t][r] = a(row wave*16 + 4*(lane>>4) + r) . b(column t*16 + (lane & 15)) -> distance, top-m update
  auto epilogue = [&](mf_float4 (&acc)[4], uint32_t nb_tile, uint32_t* Kw, bool k_aliases_bt) {
  // ---- epilogue: C[row = 4*(lane>>4) + r][col = lane&15] of tile t -> distance, top-m update ----
  const int q = lane >> 4;
  if (lane_lists) {
    if (k_aliases_bt) __syncthreads();                    // every wave is done reading Bt: reuse it as K
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t row = q * 4 + r, ar = wave * DT_AW + row;
      const float an = An[ar];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t bc = t * 16 + (lane & 15);
        const uint32_t bid = Bid[bc];
        float dist;
        if constexpr (METRIC == PANN_L2) dist = (an + Bn[bc]) - 2.0f * acc[t][r];
        else dist = -acc[t][r];
        bool ok = (bc < nb_tile) && (ar < na_tile);
        if (A.exclude_same_id) ok = ok && (bid != Aid[ar]);
        Kw[row * MF_KS + t * 17 + (lane & 15)] = ok ? f2ord(dist) : 0xFFFFFFFFu;
      }
    }
    wave_lds_sync();
    {
      const uint32_t* krow = Kw + (lane & 15) * MF_KS + (lane >> 4) * 17;
      const uint32_t* idq = Bid + (lane >> 4) * 16;
      uint64_t tau = TL[15];
      uint32_t bits = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) bits |= (krow[j] <= (uint32_t)(tau >> 32) && krow[j] != 0xFFFFFFFFu) ? (1u << j) : 0u;
      while (bits) {
        const int j = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint64_t x = ((uint64_t)krow[j] << 32) | idq[j];
        if (x < tau) { tm_chain_insert(TL, x); tau = TL[15]; }
      }
    }
    return;                                               // (K aliasing Bt: the next tile's barrier protects it)
  }
  // m > 16 (ground truth): the 16 distances of a lane (4 rows x 4 columns) are tested against register copies of their rows'
  // m-th best FIRST -- after the lists have warmed up nearly every tile ends here with one ballot; only a tile with a
  // survivor walks the (row, column) pairs and inserts (the lists live in LDS, one wave-wide shift per insert)
  {
    uint64_t key[4][4];
    bool pass[4][4];
    bool any = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t ar = wave * DT_AW + q * 4 + r;          // this lane's A row for register r
      const float an = An[ar];
      const uint64_t tau = (lists + (size_t)ar * A.mcap)[A.m - 1];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t bc = t * 16 + (lane & 15);
        const uint32_t bid = Bid[bc];
        float dist;
        if constexpr (METRIC == PANN_L2) dist = (an + Bn[bc]) - 2.0f * acc[t][r];
        else dist = -acc[t][r];
        key[r][t] = make_key(dist, bid);
        bool ok = (bc < nb_tile) && (ar < na_tile);
        if (A.exclude_same_id) ok = ok && (bid != Aid[ar]);
        pass[r][t] = ok && key[r][t] < tau;
        any = any || pass[r][t];
      }
    }
    if (__any(any)) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          uint64_t mask = __ballot(pass[r][t]);
          while (mask) {
            const int L = __ffsll((unsigned long long)mask) - 1;
            const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key[r][t], L);
            const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key[r][t] >> 32), L);
            const uint64_t x = ((uint64_t)khi << 32) | klo;
            mask &= mask - 1;
            uint64_t* list = lists + (size_t)(wave * DT_AW + (L >> 4) * 4 + r) * A.mcap;
            if (x < list[A.m - 1]) list_insert(list, A.mcap, x, lane);
          }
        }
      }
    }
  }
    };
  if (nseg == 1) {
    // ---- single-segment rows (every HCNNG leaf build on two-byte points): the staging of dense_gt_mfma_kernel (round 3) ----
    // unconditional clamped loads (a position beyond the piece re-reads its last row and is masked by nb_tile), the ids ONE tile
    // ahead of the rows and the rows one tile ahead of the multiply, barriers that order LDS only (__syncthreads() also drains
    // the vector memory counter: it waited for the tile it had just requested), K in its own region so that no barrier stands
    // between the multiply and the epilogue: two LDS barriers per tile instead of four full ones.
    uint32_t* Ks = reinterpret_cast<uint32_t*>(lists + (size_t)DT_A * A.mcap) + wave * (DT_AW * MF_KS);
    const int r0 = tid >> 4, c = tid & 15;
    const uint32_t nchunk = A.pstride >> 4;
    const bool cvalid = (uint32_t)c < nchunk;
    const uint8_t* cbase = A.points + min((uint32_t)c, nchunk - 1u) * 16u;
    const uint64_t last = be - 1;                                       // (bs < be inside)
    uint32_t id_cur[4], id_nxt[4];
    uint4 prer[4];
    auto load_ids = [&](uint64_t bt, uint32_t (&ids)[4]) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint64_t pos = min(bt + (uint64_t)(r0 + 16 * k), last);
        ids[k] = A.b_ids ? A.b_ids[pos] : (uint32_t)pos;
      }
    };
    auto load_rows = [&](const uint32_t (&ids)[4]) {
#pragma unroll
      for (int k = 0; k < 4; k++) prer[k] = *reinterpret_cast<const uint4*>(cbase + (uint64_t)ids[k] * A.pstride);
    };
    auto store_rows = [&](const uint32_t (&ids)[4], uint64_t bt) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int r = r0 + 16 * k;
        const uint4 v = cvalid ? prer[k] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Bt + (size_t)r * DT_BSTRIDE + c * 16) = v;
        const float ss = group_sum<16>(sumsq16<BF>(v));
        if (c == 0) { Bn[r] = ss; Bid[r] = bt + (uint64_t)r < be ? ids[k] : SENTINEL; }
      }
    };
    if (bs < be) {
      load_ids(bs, id_cur);
      load_rows(id_cur);
      load_ids(bs + DT_B, id_nxt);
    }
    for (uint64_t bt = bs; bt < be; bt += DT_B) {
      const uint32_t nb_tile = (uint32_t)min((uint64_t)DT_B, be - bt);
      gt_lds_barrier();                                                 // the previous tile's multiply has read Bt
      store_rows(id_cur, bt);
      load_rows(id_nxt);                                                // tile bt + 64 (its ids arrived a tile ago)
#pragma unroll
      for (int k = 0; k < 4; k++) id_cur[k] = id_nxt[k];
      load_ids(bt + 2 * DT_B, id_nxt);                                  // ids of tile bt + 128
      gt_lds_barrier();
      mf_float4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = mf_float4{0.f, 0.f, 0.f, 0.f};
      const uint32_t ksteps = A.pstride / 64;
      for (uint32_t ks = 0; ks < ksteps; ks++) {
        const uint32_t koff = ks * 64 + (lane >> 4) * 16;
        if constexpr (BF) {
          const mf_bf8 af = *reinterpret_cast<const mf_bf8*>(At + (size_t)(wave * DT_AW + (lane & 15)) * DT_BSTRIDE + koff);
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const mf_bf8 bf = *reinterpret_cast<const mf_bf8*>(Bt + (size_t)(t * 16 + (lane & 15)) * DT_BSTRIDE + koff);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
          }
        } else {
          const mf_half8 af = *reinterpret_cast<const mf_half8*>(At + (size_t)(wave * DT_AW + (lane & 15)) * DT_BSTRIDE + koff);
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const mf_half8 bf = *reinterpret_cast<const mf_half8*>(Bt + (size_t)(t * 16 + (lane & 15)) * DT_BSTRIDE + koff);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[t], 0, 0, 0);
          }
        }
      }
      epilogue(acc, nb_tile, Ks, false);
    }
  } else
  for (uint64_t bt = bs; bt < be; bt += DT_B) {
    const uint32_t nb_tile = (uint32_t)min((uint64_t)DT_B, be - bt);
    auto b_rowptr = [&](int r) -> const uint8_t* {
      const uint64_t id = A.b_ids ? (uint64_t)A.b_ids[bt + r] : (bt + r);
      return A.points + id * A.pstride;
    };
    mf_float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = mf_float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (tid < DT_B) { Bid[tid] = tid < (int)nb_tile ? (A.b_ids ? A.b_ids[bt + tid] : (uint32_t)(bt + tid)) : SENTINEL; Bn[tid] = 0.f; }
    if (tid < DT_A) An[tid] = 0.f;
    __syncthreads();
    for (uint32_t sg = 0; sg < nseg; sg++) {
      if (sg > 0) __syncthreads();
      stage_rows(Bt, Bn, sg, b_rowptr, A.pstride, nb_tile);
      stage_rows(At, An, sg, a_rowptr, a_valid, na_tile);
      __syncthreads();
      const uint32_t ksteps = min((uint32_t)DT_SEG, A.pstride - sg * DT_SEG) / 64;   // 32 halves per MFMA
      for (uint32_t ks = 0; ks < ksteps; ks++) {
        const uint32_t koff = ks * 64 + (lane >> 4) * 16;                              // k = 8*(lane>>4) + j
        if constexpr (BF) {
          const mf_bf8 af = *reinterpret_cast<const mf_bf8*>(At + (size_t)(wave * DT_AW + (lane & 15)) * DT_BSTRIDE + koff);
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const mf_bf8 bf = *reinterpret_cast<const mf_bf8*>(Bt + (size_t)(t * 16 + (lane & 15)) * DT_BSTRIDE + koff);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
          }
        } else {
          const mf_half8 af = *reinterpret_cast<const mf_half8*>(At + (size_t)(wave * DT_AW + (lane & 15)) * DT_BSTRIDE + koff);
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const mf_half8 bf = *reinterpret_cast<const mf_half8*>(Bt + (size_t)(t * 16 + (lane & 15)) * DT_BSTRIDE + koff);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[t], 0, 0, 0);
          }
        }
      }
    }
    epilogue(acc, nb_tile, Kw, true);
  }
  __syncthreads();
  if (lane_lists) {   // rank merge of the four quarter lists of every row (as in dense_topk_kernel); At + Bt hold the lists
    uint64_t* Lw = reinterpret_cast<uint64_t*>(At) + (size_t)wave * (DT_AW * 4 * 16);
    const int row = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; i++) Lw[(row * 4 + qd) * 16 + i] = TL[i];
    const uint32_t ar = wave * DT_AW + row;
    uint64_t* out = A.partial + ((a0 + ar) * A.nsplit + blockIdx.y) * A.m;
    __syncthreads();
    const int lead = 16 - (int)A.m;
    if (ar < na_tile && qd == 0) {
      int real = 0;
#pragma unroll
      for (int o = 0; o < 4; o++) real += (int)tm_lower_bound16(Lw + (row * 4 + o) * 16, KEY_INF) - lead;
      for (int j = real; j < (int)A.m; j++) out[j] = KEY_INF;
    }
    if (ar < na_tile) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const uint64_t x = TL[e];
        if (e >= lead && x != KEY_INF) {
          int rank = e - lead;
#pragma unroll
          for (int o = 1; o < 4; o++) rank += (int)tm_lower_bound16(Lw + (row * 4 + ((qd + o) & 3)) * 16, x) - lead;
          if (rank < (int)A.m) out[rank] = x;
        }
      }
    }
    return;
  }
  for (uint32_t i = tid; i < na_tile * A.m; i += 256) {
    const uint32_t ar = i / A.m, j = i % A.m;
    A.partial[((a0 + ar) * A.nsplit + blockIdx.y) * A.m + j] = lists[(size_t)ar * A.mcap + j];
  }
}

// ---- ground-truth variant (m in 17..128, two-byte floats, rows of at most 256 bytes, B = all points) ----
// The kernel above keeps the per-row lists in LDS: at k = 100 that is 57 KB per workgroup (ONE workgroup per CU, every
// barrier and LDS round trip exposed) and an insert is a read-shift-write chain through LDS.  Here
//   * the lists live in REGISTERS: a wave owns 16 A rows, the list of a row is 128 keys = 2 x uint64 per lane (key p of the
//     sorted list in lane p & 63 of register p >> 6); an insert is a wave-wide compare + shift by one lane (DPP wave_shr:1)
//     + max -- about 25 VALU instructions and no memory;
//   * a lane tests its 16 distances of a tile against FLOAT copies of its rows' m-th best (4 registers); only a tile with
//     a survivor enters the insert loop, whose exact 64-bit compare decides ties;
//   * the A fragments of the wave stay in registers for the whole launch (the A tile is fixed), only B goes through LDS,
//     double-buffered: tile i+1 is written to the other buffer while tile i is multiplied and tile i+2 is in flight from
//     HBM in registers -- one barrier per tile;
//   * |b|^2 comes from a prepass over the points (row_norms_kernel) instead of being re-summed by each of the
//     na/64 workgroups that stream the same rows.
// LDS is 36 KB per workgroup, so 2-3 workgroups share a CU (register-bound).  Results are the same sets as the
// kernel above: a row's list ends as the m smallest (distance, id) keys of its piece, whatever the insert order.
// B tile of the ground-truth kernels: 64 rows of 256 bytes, NO padding, the 16-byte slot of a row XOR-swizzled with the row
// number: byte (row, off) lives at row * 256 + (off ^ ((row & 15) << 4)).  ds_read_b128 serves a wave in four groups of 16
// lanes that mix two quarters ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS): with the padded rows of the kernels above
// (272-byte stride) the MFMA fragment read -- lane = row (lane & 15), slot 4 ks + (lane >> 4) -- put two lanes of a group on one
// bank (35 % of the LDS cycles of round 2 were conflict cycles); with the swizzle every group reads 16 different slots.
constexpr int GT_BSTRIDE = DT_SEG;
constexpr int GT_BT_BYTES = DT_B * GT_BSTRIDE;
__device__ __forceinline__ uint32_t gt_swz(uint32_t row, uint32_t off) { return row * GT_BSTRIDE + (off ^ ((row & 15u) << 4)); }
#ifndef PANN_GT_TAU_PERIOD
#define PANN_GT_TAU_PERIOD 16
#endif
constexpr uint32_t GT_TAU_PERIOD = PANN_GT_TAU_PERIOD;      // tiles between two looks at what the other pieces of a row have published (4 / 8 / 16 / 32: see DESIGN.md K4c)

// |row|^2 of every point, summed like the staging code of the kernel above (16 lanes x 16 bytes, f32 fma chain per lane,
// butterfly over the 16 lanes)
template <bool BF>
__global__ void __launch_bounds__(256) row_norms_kernel(const uint8_t* points, uint32_t pstride, uint64_t n, float* out) {
  const uint64_t row = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < n) v = load16_guarded(points + row * pstride, c * 16, pstride);
  const float ss = group_sum<16>(sumsq16<BF>(v));
  if (row < n && c == 0) out[row] = ss;
}

// One insert step on the 128-key list of a row, held by the 16 lanes of a quarter in 8 registers (key p in register p >> 4,
// lane p & 15 of the quarter): every key above x moves one place up (row_shr:1 inside a register, row_ror:1 carries lane 15
// of register j-1 into lane 0 of register j), x lands in the gap, the largest key drops out.  The four quarters of the wave
// run this together, each on its own row with its own x (KEY_INF: nothing to insert, the list is unchanged).
template <int CTRL>
__device__ __forceinline__ uint64_t gt_dpp64(uint64_t old, uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
  return ((uint64_t)hi << 32) | lo;
}
template <int NR>
__device__ __forceinline__ void gt_quarter_insert(uint64_t (&R)[NR], uint64_t x) {
#pragma unroll
  for (int j = NR - 1; j >= 0; j--) {
    uint64_t prev = 0ull;
    if (j > 0) prev = gt_dpp64<0x121>(0ull, R[j - 1]);       // row_ror:1 -- lane 0 of the quarter <- lane 15 of register j-1
    prev = gt_dpp64<0x111>(prev, R[j]);                       // row_shr:1 -- lanes 1..15 <- lane-1 of register j; lane 0 keeps the carry
    const uint64_t in = prev > x ? prev : x;                  // the key below also moves: take it; else x lands here
    R[j] = R[j] > x ? in : R[j];
  }
}

#ifdef PANN_GT_COUNTERS
// diagnostic build only (make alt ALTFLAGS=-DPANN_GT_COUNTERS): [0] tiles x waves, [2] insert rounds
__device__ unsigned long long gt_counters[8];
#define GT_COUNT(i, v) do { if (lane == 0) atomicAdd(&gt_counters[i], (unsigned long long)(v)); } while (0)
#else
#define GT_COUNT(i, v) do { } while (0)
#endif

// selection state of a wave in the ground-truth / leaf kernels: lists, float thresholds, what was published.
// NR registers of 16 places per row: 8 (m <= 128, ground truth) or 1 (m <= 16: HCNNG leaves, small-k ground truth).
// survivor queues of the queued selection (gt_select_queued below): 16 rows per wave, GT_QCAP keys each
constexpr uint32_t GT_QCAP = 64;       // >= 4 x 16: what one tile can offer a row
constexpr size_t GT_QUEUE_BYTES = 4 * (size_t)DT_AW * GT_QCAP * 8;      // per workgroup: 32 KB
#ifndef PANN_GT_FLUSH_EVERY
#define PANN_GT_FLUSH_EVERY 32
#endif
constexpr uint32_t GT_FLUSH_EVERY = PANN_GT_FLUSH_EVERY;               // tiles between two flushes of all waves of a workgroup (a power of two)
template <int NR>
struct GtSel {
  uint64_t R[4][NR];     // R[r] = row 4q + r, right-aligned in 16*NR places (the leading ones hold key 0, which nothing displaces):
                         // the m-th best of a row is always the last place = register NR-1, lane 15 of the quarter
  float tauf[4];         // float copy of the row's threshold (min of its own m-th best and the bound shared by the pieces)
  bool rowok[4];
  uint32_t pub[4];       // last value published per row (lane 0 of the quarter)
  uint32_t cnt[4];       // keys waiting in the row's LDS queue (gt_select_queued)
};
template <int NR>
__device__ __forceinline__ void gt_sel_init(GtSel<NR>& S, uint32_t m, uint32_t row0, uint32_t na_tile, int lane) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int j = 0; j < NR; j++) S.R[r][j] = (uint32_t)(j * 16 + (lane & 15)) < 16u * NR - m ? 0ull : KEY_INF;
    S.rowok[r] = row0 + r < na_tile;
    S.tauf[r] = S.rowok[r] ? __builtin_inff() : -__builtin_inff();        // a padding row accepts nothing
    S.pub[r] = 0xFFFFFFFFu;
    S.cnt[r] = 0;
  }
}
// Every piece of a row publishes its ceil(m / nsplit)-th best.  The largest of those bounds the final m-th best from above
// (the pieces together hold at least m keys at or below it), so it is a valid threshold for every piece -- and much tighter
// than a piece's own m-th best, which has seen only 1/nsplit of the points.  A stale or missing value only lets more
// candidates through.
// max over the 16 lanes of a quarter, in every lane (the butterfly of group_sum)
__device__ __forceinline__ uint32_t gt_quarter_max(uint32_t v) {
  v = max(v, (uint32_t)dpp_mov<0xB1>((int)v));     // lane ^ 1
  v = max(v, (uint32_t)dpp_mov<0x4E>((int)v));     // lane ^ 2
  v = max(v, (uint32_t)dpp_mov<0x141>((int)v));    // row_half_mirror
  v = max(v, (uint32_t)dpp_mov<0x140>((int)v));    // row_mirror
  return v;
}
// (Lane l of a quarter reads piece l, l + 16, ... of the quarter's row 4q + r: the loads of the four row sets are issued together
//  and waited for once, then a butterfly takes the maximum.  One lane reading piece after piece in a loop -- a dependent,
//  individually awaited sc1 load per piece and row -- made a refresh every 8 tiles cost 0.4 ms of 8.1 at 10K x 1M, k = 100.)
template <int NR>
__device__ __forceinline__ void gt_sel_refresh(GtSel<NR>& S, const uint32_t* gtau_rows, uint32_t row0, uint32_t na_tile, uint32_t nsplit, int lane) {
  uint32_t g[4] = {0u, 0u, 0u, 0u};
  for (uint32_t c0 = 0; c0 < nsplit; c0 += 16) {
    const uint32_t cpiece = c0 + (uint32_t)(lane & 15);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t* gp = gtau_rows + (size_t)min(row0 + r, na_tile - 1u) * nsplit;
      if (cpiece < nsplit) g[r] = max(g[r], __hip_atomic_load(gp + cpiece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t gm = gt_quarter_max(g[r]);
    const float gf = gm == 0xFFFFFFFFu ? __builtin_inff() : ord2f(gm);
    S.tauf[r] = S.rowok[r] ? fminf(S.tauf[r], gf) : S.tauf[r];
  }
}
// dist[r][t]: distance of row 4q + r to the column this lane holds in column block t (id bid[t], SENTINEL: no column;
// a column whose id equals skip[r] is not a candidate of row r -- hcnng_index.h:153).
// Rows one r at a time; every lane offers its first still-pending column, each quarter takes the offer of its first lane.
template <int NR>
__device__ __forceinline__ void gt_select(GtSel<NR>& S, const float (&dist)[4][4], const uint32_t (&bid)[4], const uint32_t (&skip)[4],
                                          uint32_t pplace, uint32_t* gtau_mine /* [row 4q of this wave][this piece] */, uint32_t nsplit, int lane) {
  const int q = lane >> 4;
  bool any = false;
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int t = 0; t < 4; t++) any = any || (dist[r][t] <= S.tauf[r]);
  if (!__any(any)) return;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    uint32_t pend = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) pend |= (dist[r][t] <= S.tauf[r] && bid[t] != SENTINEL && bid[t] != skip[r]) ? (1u << t) : 0u;
    uint64_t mask = __ballot(pend != 0);
    bool touched = false;
    while (mask) {
      const int t0 = __ffs(pend) - 1;
      const float dsel = t0 == 0 ? dist[r][0] : t0 == 1 ? dist[r][1] : t0 == 2 ? dist[r][2] : dist[r][3];
      const uint32_t isel = t0 == 0 ? bid[0] : t0 == 1 ? bid[1] : t0 == 2 ? bid[2] : bid[3];
      const uint32_t osel = f2ord(dsel);
      uint32_t xh = 0xFFFFFFFFu, xl = 0xFFFFFFFFu;
      int mine = -1;
#pragma unroll
      for (int C = 0; C < 4; C++) {
        const uint32_t field = (uint32_t)(mask >> (16 * C)) & 0xFFFFu;
        const int L = 16 * C + (field ? __builtin_ctz(field) : 0);
        const uint32_t h = field ? __builtin_amdgcn_readlane(osel, L) : 0xFFFFFFFFu;
        const uint32_t l = field ? __builtin_amdgcn_readlane(isel, L) : 0xFFFFFFFFu;
        if (q == C) { xh = h; xl = l; mine = field ? L : -1; }
      }
      if (lane == mine) pend &= pend - 1;
      GT_COUNT(2, 1);
      gt_quarter_insert<NR>(S.R[r], ((uint64_t)xh << 32) | xl);
      // the row's m-th best may have tightened: refresh the float threshold, drop what no longer passes
      uint32_t th = 0;
#pragma unroll
      for (int C = 0; C < 4; C++) {
        const uint32_t h = __builtin_amdgcn_readlane((uint32_t)(S.R[r][NR - 1] >> 32), 16 * C + 15);
        if (q == C) th = h;
      }
      const float nt = th == 0xFFFFFFFFu ? __builtin_inff() : ord2f(th);
      S.tauf[r] = !S.rowok[r] ? -__builtin_inff() : fminf(S.tauf[r], nt);
#pragma unroll
      for (int t = 0; t < 4; t++) pend &= (dist[r][t] <= S.tauf[r]) ? ~0u : ~(1u << t);
      mask = __ballot(pend != 0);
      touched = true;
    }
    if (touched) {        // (also with a single piece: a null check here made hipcc keep the lists in scratch memory)
      // this piece's share-th best (place pplace of the list): publish it when it improved
      uint32_t ph = (uint32_t)(S.R[r][0] >> 32);
#pragma unroll
      for (int j = 1; j < NR; j++) ph = (pplace >> 4) == (uint32_t)j ? (uint32_t)(S.R[r][j] >> 32) : ph;
      uint32_t pv = 0;
#pragma unroll
      for (int C = 0; C < 4; C++) {
        const uint32_t h = __builtin_amdgcn_readlane(ph, 16 * C + (int)(pplace & 15));
        if (q == C) pv = h;
      }
      if ((lane & 15) == 0 && S.rowok[r] && pv < S.pub[r]) {
        S.pub[r] = pv;
        __hip_atomic_store(gtau_mine + (size_t)r * nsplit, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// ---- queued selection (round 3).  gt_select above inserts a survivor the moment it appears: 0.84 insert rounds per wave-tile
// at k = 100, each a ~180-instruction round that serves 1.3 of the 4 quarters on average, and the wave that inserts makes the
// other three wait at the tile barrier.  Here a survivor of the float test is only APPENDED to its row's queue in LDS (GT_QCAP
// keys per row, 16 rows per wave); when an append would not fit, the wave empties all its queues: for r = 0..3 the four
// quarters insert key i of "their" row r together (a quarter that has run out inserts KEY_INF, a no-op), so a round is one
// broadcast LDS read + the shift/max chain and serves up to four rows, and a wave-tile without a survivor costs the compares and
// nothing else.  The thresholds move only at a flush, so more candidates are queued than the immediate form inserts; a queued
// key that no longer beats its row's m-th best falls out of the chain unchanged.  Same result: every candidate at or below the
// (stale, hence larger) threshold is offered to the exact 64-bit insert.

template <int NR>
__device__ __forceinline__ void gt_flush(GtSel<NR>& S, const uint64_t* Qw /* this wave's [16][GT_QCAP] */, uint32_t pplace,
                                         uint32_t* gtau_mine, uint32_t nsplit, int lane) {
  const int q = lane >> 4;
  GT_COUNT(4, 1);                                                // flushes
  wave_lds_sync();                                               // the appends of this wave are visible to its reads
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t c0 = __builtin_amdgcn_readlane(S.cnt[r], 0), c1 = __builtin_amdgcn_readlane(S.cnt[r], 16);
    const uint32_t c2 = __builtin_amdgcn_readlane(S.cnt[r], 32), c3 = __builtin_amdgcn_readlane(S.cnt[r], 48);
    const uint32_t nmax = max(max(c0, c1), max(c2, c3));
    if (nmax == 0) continue;
    const uint64_t* Qr = Qw + (size_t)(q * 4 + r) * GT_QCAP;
    for (uint32_t i = 0; i < nmax; i++) {
      const uint64_t x = i < S.cnt[r] ? Qr[i] : KEY_INF;
      GT_COUNT(2, 1);
      gt_quarter_insert<NR>(S.R[r], x);
    }
    S.cnt[r] = 0;
    // the row's m-th best has tightened: refresh the float threshold
    uint32_t th = 0;
#pragma unroll
    for (int C = 0; C < 4; C++) {
      const uint32_t h = __builtin_amdgcn_readlane((uint32_t)(S.R[r][NR - 1] >> 32), 16 * C + 15);
      if (q == C) th = h;
    }
    const float nt = th == 0xFFFFFFFFu ? __builtin_inff() : ord2f(th);
    S.tauf[r] = !S.rowok[r] ? -__builtin_inff() : fminf(S.tauf[r], nt);
    // this piece's share-th best (place pplace of the list): publish it when it improved
    uint32_t ph = (uint32_t)(S.R[r][0] >> 32);
#pragma unroll
    for (int j = 1; j < NR; j++) ph = (pplace >> 4) == (uint32_t)j ? (uint32_t)(S.R[r][j] >> 32) : ph;
    uint32_t pv = 0;
#pragma unroll
    for (int C = 0; C < 4; C++) {
      const uint32_t h = __builtin_amdgcn_readlane(ph, 16 * C + (int)(pplace & 15));
      if (q == C) pv = h;
    }
    if ((lane & 15) == 0 && S.rowok[r] && pv < S.pub[r]) {
      S.pub[r] = pv;
      __hip_atomic_store(gtau_mine + (size_t)r * nsplit, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  wave_lds_sync();                                               // the queue may be written again
}

// Pass 1 appends every (row r, column block t) group of offers that fits; a group that does not is left for pass 2, after ONE
// flush (a single call site: inlined into each of the 16 groups the flush made the tile loop a 13 000-instruction body; a real
// loop around one copy of the append code made hipcc copy the lists at every back edge and cost 2 ms).  After a flush the queues
// are empty and a row receives at most 4 x 16 keys from one tile, so pass 2 always fits (GT_QCAP >= 64).
// NEG: d holds NEGATED distances (the matrix-core kernel computes 2 a.b - (|a|^2 + |b|^2) = -dist with one add and one fma per
// element and no sign flip of the accumulators; the test is d >= -tau, the sign modifier of the compare is free).
template <int NR, bool NEG = false>
__device__ __forceinline__ void gt_select_queued(GtSel<NR>& S, const float (&d)[4][4], const uint32_t (&bid)[4], const uint32_t (&skip)[4],
                                                 uint32_t pplace, uint32_t* gtau_mine, uint32_t nsplit, uint64_t* Qw, int lane) {
  const int q = lane >> 4;
  auto pass = [&](int r, int t) -> bool { return NEG ? (d[r][t] >= -S.tauf[r]) : (d[r][t] <= S.tauf[r]); };
  auto dist = [&](int r, int t) -> float { return NEG ? -d[r][t] : d[r][t]; };
  // the common case -- no survivor in the wave's 16 x 64 distances -- is 16 compares whose masks are OR-ed on the scalar side
  // (per row set r first: a wave-tile with a survivor has one in 1.2 of its 4 row sets on average -- 2.4 of 16 groups -- and the
  //  other row sets are skipped on their mask, a scalar test, instead of 4 x (6 vector instructions + a branch) each)
  //  the row set's test is ONE compare of its best of the four column blocks: max3 + max, then a compare, instead of four compares)
  uint64_t rowm[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float best = NEG ? fmaxf(fmaxf(fmaxf(d[r][0], d[r][1]), d[r][2]), d[r][3]) : fminf(fminf(fminf(d[r][0], d[r][1]), d[r][2]), d[r][3]);
    rowm[r] = __ballot(NEG ? (best >= -S.tauf[r]) : (best <= S.tauf[r]));
  }
  if ((rowm[0] | rowm[1] | rowm[2] | rowm[3]) == 0ull) return;
  GT_COUNT(1, 1);                                                // wave-tiles with a survivor of the float test
  // from here on this wave keeps its three siblings at the tile barrier: it goes ahead of the other workgroup's wave on its SIMD
  // (7.45 -> 7.3 ms at 10K x 1M, k = 100)
  __builtin_amdgcn_s_setprio(3);
  struct PrioReset { __device__ ~PrioReset() { __builtin_amdgcn_s_setprio(0); } } prio_reset;
  const uint32_t below = (1u << (lane & 15)) - 1u;
  uint32_t left = 0;                                             // wave-uniform: bit 4r + t
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (rowm[r] == 0ull) continue;                               // wave-uniform, scalar
    // the four column blocks of the row set together: ONE wave-uniform decision (does every quarter's queue take all its
    // offers?) instead of two per block -- every such decision is a vector compare the scalar unit has to wait for
    bool p[4]; uint32_t f[4]; uint32_t n = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      p[t] = pass(r, t) && bid[t] != SENTINEL && bid[t] != skip[r];
      f[t] = (uint32_t)(__ballot(p[t]) >> (16 * q)) & 0xFFFFu;   // the offers of my quarter
      n += __popc(f[t]);
    }
    GT_COUNT(5, 1); GT_COUNT(3, n);                              // row sets with an offer; keys offered (lane 0's quarter)
    if (__any(S.cnt[r] + n > GT_QCAP)) { left |= 0xFu << (4 * r); continue; }
    uint32_t pos = S.cnt[r];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (p[t]) Qw[(size_t)(q * 4 + r) * GT_QCAP + pos + __popc(f[t] & below)] = make_key(dist(r, t), bid[t]);
      pos += __popc(f[t]);
    }
    S.cnt[r] = pos;
  }
  if (left == 0) return;
  gt_flush<NR>(S, Qw, pplace, gtau_mine, nsplit, lane);
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!((left >> (4 * r + t)) & 1u)) continue;
      const bool p = pass(r, t) && bid[t] != SENTINEL && bid[t] != skip[r];      // (the threshold has just tightened)
      const uint32_t f = (uint32_t)(__ballot(p) >> (16 * q)) & 0xFFFFu;
      if (p) Qw[(size_t)(q * 4 + r) * GT_QCAP + S.cnt[r] + __popc(f & below)] = make_key(dist(r, t), bid[t]);
      S.cnt[r] += __popc(f);
    }
  }
}

// this piece's lists -> partial[row][piece][0..m): place p of a list is entry p - (16*NR - m)
template <int NR>
__device__ __forceinline__ void gt_sel_write(const GtSel<NR>& S, uint64_t* partial_row0 /* [row 4q of this wave][this piece][0] */,
                                             uint32_t m, uint32_t nsplit, int lane) {
  const uint32_t lead = 16u * NR - m;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (S.rowok[r]) {
      uint64_t* out = partial_row0 + (size_t)r * nsplit * m;
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const uint32_t p = (uint32_t)(j * 16 + (lane & 15));
        if (p >= lead) out[p - lead] = S.R[r][j];
      }
    }
  }
}

template <int METRIC, bool BF, int NR>
#ifndef PANN_GT_WGS
#define PANN_GT_WGS 2     /* workgroups per CU the register budget allows (3: measurement builds; the k > 16 lists then spill) */
#endif
__global__ void __launch_bounds__(256, PANN_GT_WGS) dense_gt_mfma_kernel(DenseArgs A, const float* __restrict__ bnorm, uint32_t* gtau) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint8_t* Bt0 = smem;                                                    // [2][64][GT_BSTRIDE], slots swizzled (gt_swz)
  float2* Bm = reinterpret_cast<float2*>(smem + 2 * GT_BT_BYTES);         // [2][64] (|b|^2, id bits)
  float* An = reinterpret_cast<float*>(Bm + 2 * DT_B);                    // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
  uint64_t* Qw = reinterpret_cast<uint64_t*>(An + DT_A) + (size_t)wave * DT_AW * GT_QCAP;   // this wave's survivor queues [16][GT_QCAP]
  const uint64_t a0 = (uint64_t)blockIdx.x * DT_A;
  const uint32_t na_tile = (uint32_t)min((uint64_t)DT_A, A.na - a0);
  const uint64_t per = ((A.nb + A.nsplit - 1) / A.nsplit + DT_B - 1) / DT_B * DT_B;
  const uint64_t bs = min(A.nb, (uint64_t)blockIdx.y * per), be = min(A.nb, (uint64_t)(blockIdx.y + 1) * per);
  const uint32_t ntile = (uint32_t)((be - bs + DT_B - 1) / DT_B);
  const uint32_t a_valid = A.a_ids ? A.pstride : A.dbytes;
  auto a_load = [&](uint32_t r, uint32_t off) -> uint4 {
    if (r >= na_tile) return make_uint4(0, 0, 0, 0);
    const uint8_t* rp = A.a_ids ? A.points + (uint64_t)A.a_ids[a0 + r] * A.pstride : A.a_ext + (a0 + r) * A.a_stride;
    if ((reinterpret_cast<uintptr_t>(rp) & 15) == 0) return load16_guarded(rp, off, a_valid);
    uint8_t tmp[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tmp[i] = (off + i < a_valid) ? rp[off + i] : (uint8_t)0;
    uint4 v; __builtin_memcpy(&v, tmp, 16);
    return v;
  };
  const int r0 = tid >> 4, c = tid & 15;
  // |a|^2 of the 64 A rows (same summation as the staging code above)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float ss = group_sum<16>(sumsq16<BF>(a_load(r0 + 16 * k, c * 16)));
    if (c == 0) An[r0 + 16 * k] = ss;
  }
  // A fragments of this wave: row wave*16 + (lane & 15), bytes ks*64 + q*16 .. +16 of it, for the 4 k-steps of a 256-byte row
  uint4 af[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) af[ks] = a_load(wave * DT_AW + (lane & 15), ks * 64 + q * 16);

  // The queue pays where an insert is long: k > 16 (NR = 8, an 80-instruction shift chain: 12.9 -> 11.0 ms at k = 100).  At
  // k <= 16 (NR = 1) an insert is one shift + max step, cheaper than a trip through the queue (6.8 vs 7.5 ms at k = 10).
#ifdef PANN_GT_IMMEDIATE
  constexpr bool QUEUED = false;
#else
  constexpr bool QUEUED = NR > 1;
#endif
  GtSel<NR> S;
  gt_sel_init<NR>(S, A.m, (uint32_t)(wave * DT_AW + q * 4), na_tile, lane);
  const uint32_t skip[4] = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
  __syncthreads();
  float an[4];
#pragma unroll
  for (int r = 0; r < 4; r++) an[r] = An[wave * DT_AW + q * 4 + r];
  float nan[4];
#pragma unroll
  for (int r = 0; r < 4; r++) nan[r] = -an[r];
  const uint32_t pplace = 16u * NR - A.m + (A.m + A.nsplit - 1) / A.nsplit - 1;   // list place of the ceil(m/nsplit)-th best
  const uint32_t tau_period = GT_TAU_PERIOD * ((A.nsplit + 7) / 8);
  uint32_t tau_wait = tau_period;
  uint32_t* gtau_mine = gtau + (a0 + wave * DT_AW + q * 4) * A.nsplit + blockIdx.y;

  // Two tiles are in flight in registers (pa*, pb*), each requested TWO iterations before it is written to LDS: with one set
  // an iteration could not be shorter than the latency of the L2 miss behind it (round 3: 1.9 us per tile, whatever the
  // instruction stream did -- one 16 KB tile outstanding per workgroup = 2.9 TB/s at that latency).
  uint4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
  float pan = 0.f, pbn = 0.f;
  // Branch-free, mask-free staging: rows are 16-byte aligned, a multiple of 16 bytes long and zero padded
  // (pann_index_create), so every lane reads a clamped row unconditionally; a row index beyond the piece re-reads the
  // piece's last row and is labelled SENTINEL (never offered to a list).  Lanes whose 16-byte column lies beyond the row
  // length do not write at all: their LDS columns are zeroed once, below.  (With the guarded loader's byte-wise tail
  // path in the loop the compiler drained the vector memory counter before the first MFMA of every tile.)
  const uint32_t nchunk = A.pstride >> 4;
  const bool cvalid = (uint32_t)c < nchunk;
  const uint8_t* cbase = A.points + min((uint32_t)c, nchunk - 1u) * 16u;
  const uint32_t last_row = (uint32_t)(be - 1);                    // be > bs wherever a load is issued; positions are 32-bit
  // (named registers, not arrays or a struct passed by reference: those the compiler kept in scratch memory)
#define GT_LOAD_PRE(v0, v1, v2, v3, vn, bt_)          /* requests only: nothing here may depend on the loaded values */              \
  do {                                                                                                                              \
    const uint32_t b_ = (uint32_t)(bt_) + (uint32_t)r0;                                                                             \
    v0 = *reinterpret_cast<const uint4*>(cbase + (uint64_t)min(b_, last_row) * A.pstride);                                          \
    v1 = *reinterpret_cast<const uint4*>(cbase + (uint64_t)min(b_ + 16u, last_row) * A.pstride);                                    \
    v2 = *reinterpret_cast<const uint4*>(cbase + (uint64_t)min(b_ + 32u, last_row) * A.pstride);                                    \
    v3 = *reinterpret_cast<const uint4*>(cbase + (uint64_t)min(b_ + 48u, last_row) * A.pstride);                                    \
    if (METRIC == PANN_L2) vn = bnorm[min((uint32_t)(bt_) + (uint32_t)lane, last_row)];    /* every wave: no branch around a load */ \
  } while (0)
  auto store_tile = [&](const uint4& v0, const uint4& v1, const uint4& v2, const uint4& v3, float vn, int buf, uint64_t bt) {
    if (cvalid) {
      uint8_t* dst = Bt0 + buf * GT_BT_BYTES + gt_swz((uint32_t)r0, (uint32_t)c * 16u);      // rows r0 + 16 k share r0's swizzle
      *reinterpret_cast<uint4*>(dst) = v0;
      *reinterpret_cast<uint4*>(dst + 16 * GT_BSTRIDE) = v1;
      *reinterpret_cast<uint4*>(dst + 32 * GT_BSTRIDE) = v2;
      *reinterpret_cast<uint4*>(dst + 48 * GT_BSTRIDE) = v3;
    }
    if (tid < DT_B)
      Bm[buf * DT_B + tid] = make_float2(vn, __uint_as_float(bt + tid < be ? (uint32_t)(bt + tid) : SENTINEL));
  };
  // tile bt -> LDS buffer buf from its register set, then the set is refilled with tile bt + 2 tiles
  auto stage_a = [&](int buf, uint64_t bt) { store_tile(pa0, pa1, pa2, pa3, pan, buf, bt); GT_LOAD_PRE(pa0, pa1, pa2, pa3, pan, bt + 2 * DT_B); };
  auto stage_b = [&](int buf, uint64_t bt) { store_tile(pb0, pb1, pb2, pb3, pbn, buf, bt); GT_LOAD_PRE(pb0, pb1, pb2, pb3, pbn, bt + 2 * DT_B); };
  if (!cvalid) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      *reinterpret_cast<uint4*>(Bt0 + (k >> 2) * GT_BT_BYTES + gt_swz((uint32_t)(r0 + 16 * (k & 3)), (uint32_t)c * 16u)) = make_uint4(0, 0, 0, 0);
  }
  if (ntile > 0) {           // (an empty piece issues no loads: last_row is meaningless there)
    GT_LOAD_PRE(pa0, pa1, pa2, pa3, pan, bs); store_tile(pa0, pa1, pa2, pa3, pan, 0, bs);
    GT_LOAD_PRE(pa0, pa1, pa2, pa3, pan, bs + DT_B);
    GT_LOAD_PRE(pb0, pb1, pb2, pb3, pbn, bs + 2 * DT_B);
  }
  gt_lds_barrier();

  // one tile: `buf` holds it, `stage` writes tile i+1 (requested two iterations ago) and requests tile i+3 into the same registers
  auto tile_step = [&](uint32_t i, const int buf, auto&& stage) {
    const uint64_t bt = bs + (uint64_t)i * DT_B;
    if (A.nsplit > 1 && --tau_wait == 0) {
      tau_wait = tau_period;
      gt_sel_refresh<NR>(S, gtau + a0 * A.nsplit, (uint32_t)(wave * DT_AW + q * 4), na_tile, A.nsplit, lane);
    }
    // No branches around these: beyond the last tile they re-stage the piece's last row (clamped loads, SENTINEL labels)
    // into the buffer nobody reads again -- with conditional staging the compiler waited for the requests just made
    // before the first MFMA.
    stage(buf ^ 1, bt + DT_B);                                 // tile i+1 -> the other buffer; tile i+3 -> registers, in flight during two tiles' math
    mf_float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = mf_float4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* Bt = Bt0 + buf * GT_BT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {      // always the 4 k-steps of a 256-byte row: bytes beyond the row are zero in both tiles
      const uint32_t koff = ks * 64 + q * 16;
      if constexpr (BF) {
        mf_bf8 a8; __builtin_memcpy(&a8, &af[ks], 16);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const mf_bf8 b8 = *reinterpret_cast<const mf_bf8*>(Bt + gt_swz((uint32_t)(t * 16 + (lane & 15)), koff));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[t], 0, 0, 0);
        }
      } else {
        mf_half8 a8; __builtin_memcpy(&a8, &af[ks], 16);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const mf_half8 b8 = *reinterpret_cast<const mf_half8*>(Bt + gt_swz((uint32_t)(t * 16 + (lane & 15)), koff));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[t], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: acc[t][r] = a(row wave*16 + 4q + r) . b(column t*16 + (lane & 15)) ----
    float bn[4]; uint32_t bid[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { const float2 m2 = Bm[buf * DT_B + t * 16 + (lane & 15)]; bn[t] = m2.x; bid[t] = __float_as_uint(m2.y); }
    GT_COUNT(0, 1);
    auto distf = [&](int r, int t) -> float {
      if constexpr (METRIC == PANN_L2) return (an[r] + bn[t]) - 2.0f * acc[t][r];
      else return -acc[t][r];
    };
    if constexpr (QUEUED) {
      // negated distances: -(|a|^2 + |b|^2) + 2 a.b -- the same value as the formula above with the sign flipped (negation is
      // exact and rounding is symmetric), without a sign flip per accumulator
      float nd[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if constexpr (METRIC == PANN_L2) nd[r][t] = __builtin_fmaf(2.0f, acc[t][r], nan[r] - bn[t]);
          else nd[r][t] = acc[t][r];
        }
#ifdef PANN_GT_DIAG_NOSELECT      /* timing diagnostic: no selection at all (results are wrong) */
      { float mx = nd[0][0];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int t = 0; t < 4; t++) mx = fmaxf(mx, nd[r][t]);
        S.R[0][0] += (uint64_t)__float_as_uint(mx + __uint_as_float(bid[0])); }
#else
      gt_select_queued<NR, true>(S, nd, bid, skip, pplace, gtau_mine, A.nsplit, Qw, lane);
#endif
      // all four waves of the workgroup also empty their queues at the SAME tiles, every GT_FLUSH_EVERY-th: a wave that flushes
      // alone keeps its three siblings at the tile barrier for the whole flush (8 / 32 / 64 / 128 tiles: 10.3 / 10.1 / 10.15 /
      // 10.25 ms, never: 11.5 ms at 10K x 1M, k = 100)
      if ((i & (GT_FLUSH_EVERY - 1)) == GT_FLUSH_EVERY - 1) gt_flush<NR>(S, Qw, pplace, gtau_mine, A.nsplit, lane);
    } else {
      float dist[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < 4; t++) dist[r][t] = distf(r, t);
      gt_select<NR>(S, dist, bid, skip, pplace, gtau_mine, A.nsplit, lane);
    }
#ifdef PANN_GT_DIAG_NOBARRIER     /* timing diagnostic: the waves of a workgroup run free (results are wrong) */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    gt_lds_barrier();
#endif
  };
  // (an odd tile count runs one more step on a tile of SENTINEL labels: no branch around the second half's loads)
  for (uint32_t i = 0; i < ntile; i += 2) {
    tile_step(i, 0, stage_a);
    tile_step(i + 1, 1, stage_b);
  }
  if constexpr (QUEUED) gt_flush<NR>(S, Qw, pplace, gtau_mine, A.nsplit, lane);
  gt_sel_write<NR>(S, A.partial + ((a0 + wave * DT_AW + q * 4) * A.nsplit + blockIdx.y) * A.m, A.m, A.nsplit, lane);
#undef GT_LOAD_PRE
}

// ---- the same kernel for the types whose contraction stays on the VALU (north_star: no MFMA for int8/uint8; f32 has none
// that is exact): a 4 x 4 register tile per lane in the MFMA's own output layout (rows 4q + r, columns t*16 + (lane & 15)), so
// the selection above is shared.  Per 16-byte chunk of the dimension a lane reads 4 A chunks (one address per quarter: LDS
// broadcasts) and 4 B chunks and issues 64 v_dot4 (one-byte types) or 64 (sub +) fma (f32).  A sits in LDS (up to two
// 256-byte segments of the row), B streams through the same double buffer, one segment per pipeline step.
template <int DT, int METRIC>
__device__ __forceinline__ void gt_chunk_accum(typename AccT<DT>::type (&acc)[4][4], const uint4 (&av)[4], const uint4 (&bv)[4]) {
#pragma unroll
  for (int t = 0; t < 4; t++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if constexpr (DT == PANN_U8) {
        uint32_t x = (uint32_t)acc[t][r];
        x = __builtin_amdgcn_udot4(av[r].x, bv[t].x, x, false); x = __builtin_amdgcn_udot4(av[r].y, bv[t].y, x, false);
        x = __builtin_amdgcn_udot4(av[r].z, bv[t].z, x, false); x = __builtin_amdgcn_udot4(av[r].w, bv[t].w, x, false);
        acc[t][r] = (int)x;
      } else if constexpr (DT == PANN_I8) {
        int x = acc[t][r];
        x = __builtin_amdgcn_sdot4((int)av[r].x, (int)bv[t].x, x, false); x = __builtin_amdgcn_sdot4((int)av[r].y, (int)bv[t].y, x, false);
        x = __builtin_amdgcn_sdot4((int)av[r].z, (int)bv[t].z, x, false); x = __builtin_amdgcn_sdot4((int)av[r].w, (int)bv[t].w, x, false);
        acc[t][r] = x;
      } else {
        const float a4[4] = {__uint_as_float(av[r].x), __uint_as_float(av[r].y), __uint_as_float(av[r].z), __uint_as_float(av[r].w)};
        const float b4[4] = {__uint_as_float(bv[t].x), __uint_as_float(bv[t].y), __uint_as_float(bv[t].z), __uint_as_float(bv[t].w)};
        float x = acc[t][r];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if constexpr (METRIC == PANN_L2) { const float d = a4[e] - b4[e]; x = fmaf(d, d, x); }
          else x = fmaf(a4[e], b4[e], x);
        }
        acc[t][r] = x;
      }
    }
  }
}
// sum of squares of 16 bytes of a one-byte type
template <int DT>
__device__ __forceinline__ int sumsq16_i8(uint4 v) {
  if constexpr (DT == PANN_U8) {
    uint32_t x = __builtin_amdgcn_udot4(v.x, v.x, 0u, false); x = __builtin_amdgcn_udot4(v.y, v.y, x, false);
    x = __builtin_amdgcn_udot4(v.z, v.z, x, false); x = __builtin_amdgcn_udot4(v.w, v.w, x, false);
    return (int)x;
  } else {
    int x = __builtin_amdgcn_sdot4((int)v.x, (int)v.x, 0, false); x = __builtin_amdgcn_sdot4((int)v.y, (int)v.y, x, false);
    x = __builtin_amdgcn_sdot4((int)v.z, (int)v.z, x, false); x = __builtin_amdgcn_sdot4((int)v.w, (int)v.w, x, false);
    return x;
  }
}
// |row|^2 of every point of a one-byte type as an int32 (exact), stored through the float pointer
template <int DT>
__global__ void __launch_bounds__(256) row_norms_i8_kernel(const uint8_t* points, uint32_t pstride, uint64_t n, int* out) {
  const uint64_t row = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  int ss = 0;
  if (row < n) for (uint32_t off = c * 16; off < pstride; off += 256) ss += sumsq16_i8<DT>(*reinterpret_cast<const uint4*>(points + row * pstride + off));
  ss = group_sum<16>(ss);
  if (row < n && c == 0) out[row] = ss;
}

template <int DT, int METRIC, int NR>
__global__ void __launch_bounds__(256, 2) dense_gt_valu_kernel(DenseArgs A, const int* __restrict__ bnorm, uint32_t* gtau) {
  using acc_t = typename AccT<DT>::type;
  constexpr bool INTS = DT == PANN_U8 || DT == PANN_I8;
  extern __shared__ __align__(16) uint8_t smem[];
  const uint32_t nseg = (A.pstride + DT_SEG - 1) / DT_SEG;              // 1 or 2 (the launcher checks)
  const uint32_t astride = nseg * DT_SEG + 16;
  uint8_t* Bt0 = smem;                                                    // [2][64][GT_BSTRIDE], slots swizzled (gt_swz)
  float2* Bm = reinterpret_cast<float2*>(smem + 2 * GT_BT_BYTES);         // [2][64] (|b|^2 bits, id bits)
  int* An = reinterpret_cast<int*>(Bm + 2 * DT_B);                        // [64] |a|^2 (one-byte types, L2)
  uint8_t* At = reinterpret_cast<uint8_t*>(An + DT_A);                    // [64][astride]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
  const uint64_t a0 = (uint64_t)blockIdx.x * DT_A;
  const uint32_t na_tile = (uint32_t)min((uint64_t)DT_A, A.na - a0);
  const uint64_t per = ((A.nb + A.nsplit - 1) / A.nsplit + DT_B - 1) / DT_B * DT_B;
  const uint64_t bs = min(A.nb, (uint64_t)blockIdx.y * per), be = min(A.nb, (uint64_t)(blockIdx.y + 1) * per);
  const uint32_t ntile = (uint32_t)((be - bs + DT_B - 1) / DT_B);
  const uint32_t a_valid = A.a_ids ? A.pstride : A.dbytes;
  auto a_load = [&](uint32_t r, uint32_t off) -> uint4 {
    if (r >= na_tile) return make_uint4(0, 0, 0, 0);
    const uint8_t* rp = A.a_ids ? A.points + (uint64_t)A.a_ids[a0 + r] * A.pstride : A.a_ext + (a0 + r) * A.a_stride;
    if ((reinterpret_cast<uintptr_t>(rp) & 15) == 0) return load16_guarded(rp, off, a_valid);
    uint8_t tmp[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tmp[i] = (off + i < a_valid) ? rp[off + i] : (uint8_t)0;
    uint4 v; __builtin_memcpy(&v, tmp, 16);
    return v;
  };
  const int r0 = tid >> 4, c = tid & 15;
  // the A tile -> LDS, with |a|^2 for the one-byte L2 form |a|^2 + |b|^2 - 2 a.b (all int32, exact)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int ss = 0;
    for (uint32_t sg = 0; sg < nseg; sg++) {
      const uint4 v = a_load(r0 + 16 * k, sg * DT_SEG + c * 16);
      *reinterpret_cast<uint4*>(At + (size_t)(r0 + 16 * k) * astride + sg * DT_SEG + c * 16) = v;
      if constexpr (INTS) ss += sumsq16_i8<DT>(v);
    }
    if constexpr (INTS) { ss = group_sum<16>(ss); if (c == 0) An[r0 + 16 * k] = ss; }
  }
  GtSel<NR> S;
  gt_sel_init<NR>(S, A.m, (uint32_t)(wave * DT_AW + q * 4), na_tile, lane);
  __syncthreads();
  int an[4] = {0, 0, 0, 0};
  if constexpr (INTS && METRIC == PANN_L2) {
#pragma unroll
    for (int r = 0; r < 4; r++) an[r] = An[wave * DT_AW + q * 4 + r];
  }
  const uint32_t pplace = 16u * NR - A.m + (A.m + A.nsplit - 1) / A.nsplit - 1;
  const uint32_t skip[4] = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
  const uint32_t tau_period = GT_TAU_PERIOD * ((A.nsplit + 7) / 8);
  uint32_t tau_wait = tau_period;
  uint32_t* gtau_mine = gtau + (a0 + wave * DT_AW + q * 4) * A.nsplit + blockIdx.y;

  // pipeline step u = (tile u / nseg, segment u % nseg); buffer u & 1 therefore always holds the same segment.  Only the
  // chunks that hold data are staged and multiplied (the zero padding of a row adds nothing): chunk j of segment sg is
  // valid when sg*16 + j < nch.
  const uint32_t nch = (A.dbytes + 15) / 16;
  const uint32_t nunits = ntile * nseg;
  const uint32_t last_row = (uint32_t)(be - 1);
  const bool cv0 = (uint32_t)c < nch, cv1 = nseg == 2 && 16u + (uint32_t)c < nch;
  const uint8_t* cb0 = A.points + min((uint32_t)c, nch - 1u) * 16u;
  const uint8_t* cb1 = A.points + min(16u + (uint32_t)c, nch - 1u) * 16u;
  uint4 pre0, pre1, pre2, pre3;
  int pn = 0;
  auto unit_bt = [&](uint32_t u) -> uint32_t { return (uint32_t)bs + (nseg == 2 ? (u >> 1) : u) * DT_B; };
  auto load_pre = [&](uint32_t u) {          // requests only
    const uint32_t bt = unit_bt(u);
    const uint8_t* cb = (nseg == 2 && (u & 1)) ? cb1 : cb0;
    pre0 = *reinterpret_cast<const uint4*>(cb + (uint64_t)min(bt + (uint32_t)r0, last_row) * A.pstride);
    pre1 = *reinterpret_cast<const uint4*>(cb + (uint64_t)min(bt + (uint32_t)r0 + 16u, last_row) * A.pstride);
    pre2 = *reinterpret_cast<const uint4*>(cb + (uint64_t)min(bt + (uint32_t)r0 + 32u, last_row) * A.pstride);
    pre3 = *reinterpret_cast<const uint4*>(cb + (uint64_t)min(bt + (uint32_t)r0 + 48u, last_row) * A.pstride);
    if (INTS && METRIC == PANN_L2) pn = bnorm[min(bt + (uint32_t)lane, last_row)];
  };
  auto store_pre = [&](uint32_t u) {
    const uint32_t bt = unit_bt(u);
    const int buf = (int)(u & 1);
    if ((nseg == 2 && (u & 1)) ? cv1 : cv0) {
      uint8_t* dst = Bt0 + buf * GT_BT_BYTES + gt_swz((uint32_t)r0, (uint32_t)c * 16u);      // rows r0 + 16 k share r0's swizzle
      *reinterpret_cast<uint4*>(dst) = pre0;
      *reinterpret_cast<uint4*>(dst + 16 * GT_BSTRIDE) = pre1;
      *reinterpret_cast<uint4*>(dst + 32 * GT_BSTRIDE) = pre2;
      *reinterpret_cast<uint4*>(dst + 48 * GT_BSTRIDE) = pre3;
    }
    if (tid < DT_B)
      Bm[buf * DT_B + tid] = make_float2(__int_as_float(pn), __uint_as_float((uint64_t)bt + tid < be ? bt + (uint32_t)tid : SENTINEL));
  };
  if (nunits > 0) {
    load_pre(0); store_pre(0);
    load_pre(1);
  }
  gt_lds_barrier();

  acc_t acc[4][4];
  for (uint32_t u = 0; u < nunits; u++) {
    const int buf = (int)(u & 1);
    const uint32_t sg = nseg == 2 ? (u & 1) : 0u;
    if (sg == 0 && A.nsplit > 1 && --tau_wait == 0) { tau_wait = tau_period; gt_sel_refresh<NR>(S, gtau + a0 * A.nsplit, (uint32_t)(wave * DT_AW + q * 4), na_tile, A.nsplit, lane); }
    store_pre(u + 1);                                          // (beyond the last step: the piece's last row again, into a dead buffer)
    load_pre(u + 2);
    if (sg == 0) {
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[t][r] = (acc_t)0;
    }
    const uint8_t* Bp = Bt0 + buf * GT_BT_BYTES + (size_t)(lane & 15) * GT_BSTRIDE;
    const uint32_t bsw = (uint32_t)(lane & 15) << 4;           // the row's swizzle (rows (lane & 15) + 16 t share it)
    const uint8_t* Ap = At + (size_t)(wave * DT_AW + q * 4) * astride + sg * DT_SEG;
    const uint32_t nc = nch > sg * 16u ? min(16u, nch - sg * 16u) : 0u;
    for (uint32_t j = 0; j < nc; j++) {
      uint4 av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) av[r] = *reinterpret_cast<const uint4*>(Ap + (size_t)r * astride + j * 16);
#pragma unroll
      for (int t = 0; t < 4; t++) bv[t] = *reinterpret_cast<const uint4*>(Bp + (size_t)t * 16 * GT_BSTRIDE + ((j * 16) ^ bsw));
      gt_chunk_accum<DT, METRIC>(acc, av, bv);
    }
    if (sg == nseg - 1) {
      int bn[4]; uint32_t bid[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { const float2 m2 = Bm[buf * DT_B + t * 16 + (lane & 15)]; bn[t] = __float_as_int(m2.x); bid[t] = __float_as_uint(m2.y); }
      float dist[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if constexpr (INTS) {
            if constexpr (METRIC == PANN_L2) dist[r][t] = (float)(an[r] + bn[t] - 2 * (int)acc[t][r]);
            else dist[r][t] = -(float)(int)acc[t][r];
          } else {
            if constexpr (METRIC == PANN_L2) dist[r][t] = (float)acc[t][r];
            else dist[r][t] = -(float)acc[t][r];
          }
        }
      }
      GT_COUNT(0, 1);
      // (immediate inserts here: the contraction, not the selection, bounds these kernels -- u8 24.8 ms queued vs 25.2 -- and the
      // queues would take the LDS of the second workgroup per CU)
      gt_select<NR>(S, dist, bid, skip, pplace, gtau_mine, A.nsplit, lane);
    }
    gt_lds_barrier();
  }
  gt_sel_write<NR>(S, A.partial + ((a0 + wave * DT_AW + q * 4) * A.nsplit + blockIdx.y) * A.m, A.m, A.nsplit, lane);
}

// merge the nsplit partial lists of each A row (one wave per row) and write ids / dists
__global__ void __launch_bounds__(64) dense_merge_kernel(const uint64_t* partial, uint64_t na, uint32_t nsplit,
                                                         uint32_t m, uint32_t mcap, uint32_t* out_ids, float* out_dists) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint64_t* list = reinterpret_cast<uint64_t*>(smem);
  const int lane = threadIdx.x;
  const uint64_t ar = blockIdx.x;
  for (uint32_t i = lane; i < mcap; i += 64) list[i] = KEY_INF;
  wave_lds_sync();
  const uint64_t* src = partial + ar * nsplit * m;
  const uint32_t total = nsplit * m;
  for (uint32_t i0 = 0; i0 < total; i0 += 64) {
    const uint32_t i = i0 + lane;
    const uint64_t key = i < total ? src[i] : KEY_INF;
    uint64_t tau = list[m - 1];
    uint64_t mask = __ballot(key < tau);
    while (mask) {
      const int L = __ffsll((unsigned long long)mask) - 1;
      const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, L);
      const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), L);
      const uint64_t x = ((uint64_t)khi << 32) | klo;
      mask &= mask - 1;
      if (x < tau) { list_insert(list, mcap, x, lane); tau = list[m - 1]; }
    }
  }
  for (uint32_t j = lane; j < m; j += 64) {
    const uint64_t k = list[j];
    out_ids[ar * m + j] = (k == KEY_INF) ? SENTINEL : key_id(k);
    out_dists[ar * m + j] = (k == KEY_INF) ? __builtin_inff() : key_dist(k);
  }
}

// ---------------------------------------------------------------------------------------------

constexpr size_t GT_LDS_BYTES = 2 * (size_t)GT_BT_BYTES + 2 * DT_B * sizeof(float2) + DT_A * sizeof(float);

// external / id'd A rows against the contiguous range of all points (ground truth), any supported type, m <= 128.
// (The HCNNG leaf form -- rows gathered by id, ~1000 columns per row, m = 10 -- was tried on these kernels and lost to the
// lane-list kernels above, 0.167 s vs 0.121 s for 30 trees over 1M x 128 fp16: a row sees so few columns that 5.6 % of them
// enter its list, and 64 lane-owned lists take that in parallel where the quarter lists take four at a time.)
bool dense_gt_eligible(const DeviceIndex& ix, uint32_t m, bool b_ids, bool segmented, int exclude_same) {
  static const bool off = ab_env("PANN_GT_OLD") != nullptr;      // diagnostic A/B switch
  if (off || m == 0 || m > 128 || ix.exact || b_ids || segmented || exclude_same) return false;
  const bool twobyte = ix.dtype == PANN_F16 || ix.dtype == PANN_BF16;
  return twobyte ? ix.pstride <= 256 : ix.pstride <= 512;        // matrix cores: one 256-byte segment; VALU register tile: two
}
static size_t dense_gt_lds(const DeviceIndex& ix, uint32_t m) {
  if (ix.dtype == PANN_F16 || ix.dtype == PANN_BF16)       // matrix-core kernel: + the survivor queues (k > 16: one list register per row inserts at once)
    return GT_LDS_BYTES + (m > 16 ? GT_QUEUE_BYTES : 0);
  const size_t nseg = (ix.pstride + DT_SEG - 1) / DT_SEG;
  return GT_LDS_BYTES + (size_t)DT_A * (nseg * DT_SEG + 16);
}
template <typename F>
static int dense_gt_pick(const DeviceIndex& ix, uint32_t m, F&& f) {
  // list registers per row: 1 (k <= 16), 7 (k <= 112: the usual ground truth, k = 100 -- an insert step is one register shorter
  // than with 8 and the kernel holds 8 registers less) or 8
  const bool l2 = ix.metric == PANN_L2, small = m <= 16, mid = m <= 112;
#define GT_MF(BF)                                                                                                                   \
  (small ? (l2 ? f(dense_gt_mfma_kernel<PANN_L2, BF, 1>) : f(dense_gt_mfma_kernel<PANN_MIPS, BF, 1>))                                \
   : mid ? (l2 ? f(dense_gt_mfma_kernel<PANN_L2, BF, 7>) : f(dense_gt_mfma_kernel<PANN_MIPS, BF, 7>))                                \
         : (l2 ? f(dense_gt_mfma_kernel<PANN_L2, BF, 8>) : f(dense_gt_mfma_kernel<PANN_MIPS, BF, 8>)))
#define GT_VA(DT)                                                                                                                   \
  (small ? (l2 ? f(dense_gt_valu_kernel<DT, PANN_L2, 1>) : f(dense_gt_valu_kernel<DT, PANN_MIPS, 1>))                                \
         : (l2 ? f(dense_gt_valu_kernel<DT, PANN_L2, 8>) : f(dense_gt_valu_kernel<DT, PANN_MIPS, 8>)))
  switch (ix.dtype) {
    case PANN_F16: return GT_MF(false);
    case PANN_BF16: return GT_MF(true);
    case PANN_U8: return GT_VA(PANN_U8);
    case PANN_I8: return GT_VA(PANN_I8);
    default: return GT_VA(PANN_F32);
  }
#undef GT_MF
#undef GT_VA
}

// workgroups of the ground-truth launch that are resident at once (for the caller's choice of nsplit)
uint32_t dense_gt_slots(const DeviceIndex& ix, uint32_t m) {
  if (!dense_gt_eligible(ix, m, false, false, 0)) return 256;
  const size_t lds = dense_gt_lds(ix, m);
  const int nb = dense_gt_pick(ix, m, [&](auto kern) -> int {
    int v = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || v < 1) v = 1;
    return v;
  });
  return 256u * (uint32_t)nb;
}

static size_t dense_lds_bytes(uint32_t mcap) {
  return (size_t)DT_A * DT_SEG + (size_t)DT_B * DT_RB * DT_BSTRIDE + (DT_A + DT_B * DT_RB) * 4 + (size_t)DT_A * mcap * 8;
}

// Runs the dense top-m.  All pointers device.  tile arrays may be null for a single segment.
int dense_topk_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint8_t* d_a_ext, uint64_t a_stride,
                   const uint32_t* d_a_ids, const uint32_t* d_b_ids, const uint64_t* d_a_off, const uint64_t* d_b_off,
                   const uint32_t* d_tile_seg, const uint32_t* d_tile_a0, uint32_t ntiles, uint64_t na, uint64_t nb,
                   uint32_t nsplit, uint32_t m, int exclude_same, uint32_t* d_out_ids, float* d_out_dists) {
  if (m == 0 || m > 128) { set_error("dense top-m: m must be in [1,128]"); return PANN_ERR_BAD_ARG; }
  if (na == 0) return PANN_OK;
  const uint32_t mcap = (m + 15) / 16 * 16;
  const size_t pbytes = (size_t)na * nsplit * m * 8;
  if (int rc = ws.ensure(pbytes + 256)) return rc;
  DenseArgs A{};
  A.points = ix.points; A.pstride = ix.pstride; A.dbytes = ix.dbytes;
  A.a_ext = d_a_ext; A.a_stride = a_stride; A.a_ids = d_a_ids; A.b_ids = d_b_ids;
  A.a_off = d_a_off; A.b_off = d_b_off; A.na = na; A.nb = nb; A.nsplit = nsplit; A.m = m; A.mcap = mcap;
  A.exclude_same_id = exclude_same; A.exact = ix.exact; A.partial = (uint64_t*)ws.buf; A.tile_seg = d_tile_seg; A.tile_a0 = d_tile_a0;
  const size_t lds = dense_lds_bytes(mcap);
  const dim3 grid(ntiles, nsplit);
  if (dense_gt_eligible(ix, m, d_b_ids != nullptr, d_a_off || d_b_off || d_tile_seg || d_tile_a0, exclude_same)) {
    // register-list ground-truth kernel; |b|^2 of every point first (after the partial lists in the workspace)
    const size_t poff = (pbytes + 255) / 256 * 256;
    const uint64_t nnorm = nb;
    const size_t noff = poff + ((size_t)nnorm * 4 + 255) / 256 * 256;
    if (int rc = ws.ensure(noff + (size_t)na * nsplit * 4 + 256)) return rc;
    A.partial = (uint64_t*)ws.buf;
    float* d_norm = reinterpret_cast<float*>(static_cast<uint8_t*>(ws.buf) + poff);
    uint32_t* d_gtau = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ws.buf) + noff);      // [A row][piece]: the piece's ceil(m/nsplit)-th best so far
    PANN_HIP(hipMemsetAsync(d_gtau, 0xFF, (size_t)na * nsplit * 4, st));
    const dim3 ng((uint32_t)((nnorm + 15) / 16));
    if (ix.metric == PANN_L2) {       // |b|^2 of every point: f32 bits for the two-byte floats, exact int32 for the one-byte types
      if (ix.dtype == PANN_BF16) hipLaunchKernelGGL(row_norms_kernel<true>, ng, dim3(256), 0, st, ix.points, ix.pstride, nnorm, d_norm);
      else if (ix.dtype == PANN_F16) hipLaunchKernelGGL(row_norms_kernel<false>, ng, dim3(256), 0, st, ix.points, ix.pstride, nnorm, d_norm);
      else if (ix.dtype == PANN_U8) hipLaunchKernelGGL(row_norms_i8_kernel<PANN_U8>, ng, dim3(256), 0, st, ix.points, ix.pstride, nnorm, (int*)d_norm);
      else if (ix.dtype == PANN_I8) hipLaunchKernelGGL(row_norms_i8_kernel<PANN_I8>, ng, dim3(256), 0, st, ix.points, ix.pstride, nnorm, (int*)d_norm);
    }
    const size_t glds = dense_gt_lds(ix, m);
    dense_gt_pick(ix, m, [&](auto kern) -> int {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
      using norm_t = std::conditional_t<std::is_invocable_v<decltype(kern), DenseArgs, const float*, uint32_t*>, const float*, const int*>;
      hipLaunchKernelGGL(kern, grid, dim3(256), glds, st, A, (norm_t)d_norm, d_gtau);
      return 0;
    });
    PANN_HIP(hipGetLastError());
    hipLaunchKernelGGL(dense_merge_kernel, dim3((uint32_t)na), dim3(64), (size_t)mcap * 8, st,
                       (const uint64_t*)ws.buf, na, nsplit, m, mcap, d_out_ids, d_out_dists);
    PANN_HIP(hipGetLastError());
    return PANN_OK;
  }
#define CALL_DENSE(DT, MT)                                                                              \
  do {                                                                                                   \
    auto kern = dense_topk_kernel<DT, MT>;                                                               \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, A);                                               \
  } while (0)
  if (ix.dtype == PANN_U8 && ix.metric == PANN_L2) CALL_DENSE(PANN_U8, PANN_L2);
  else if (ix.dtype == PANN_U8) CALL_DENSE(PANN_U8, PANN_MIPS);
  else if (ix.dtype == PANN_I8 && ix.metric == PANN_L2) CALL_DENSE(PANN_I8, PANN_L2);
  else if (ix.dtype == PANN_I8) CALL_DENSE(PANN_I8, PANN_MIPS);
  else if (ix.dtype == PANN_F32 && ix.metric == PANN_L2) CALL_DENSE(PANN_F32, PANN_L2);
  else if (ix.dtype == PANN_F32) CALL_DENSE(PANN_F32, PANN_MIPS);
  else if (ix.dtype == PANN_F16 && ix.exact && ix.metric == PANN_L2) CALL_DENSE(PANN_F16, PANN_L2);      // exact order: VALU path, not MFMA
  else if (ix.dtype == PANN_F16 && ix.exact) CALL_DENSE(PANN_F16, PANN_MIPS);
  else if (ix.dtype == PANN_BF16 && ix.exact && ix.metric == PANN_L2) CALL_DENSE(PANN_BF16, PANN_L2);
  else if (ix.dtype == PANN_BF16 && ix.exact) CALL_DENSE(PANN_BF16, PANN_MIPS);
  else {
    const size_t lds2 = (size_t)(DT_A + DT_B) * DT_BSTRIDE + (DT_A + DT_B) * 8 + (size_t)DT_A * mcap * 8 +
                        4 * (size_t)DT_AW * 68 * 4;       // + the epilogue's transposition scratch K (single-segment rows keep it apart from the B tile)
#define CALL_MFMA(MT, BF)                                                                               \
  do {                                                                                                   \
    auto kern = dense_topk_mfma_f16_kernel<MT, BF>;                                                      \
    if (lds2 > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds2, st, A);                                              \
  } while (0)
    if (ix.dtype == PANN_BF16) { if (ix.metric == PANN_L2) CALL_MFMA(PANN_L2, true); else CALL_MFMA(PANN_MIPS, true); }
    else { if (ix.metric == PANN_L2) CALL_MFMA(PANN_L2, false); else CALL_MFMA(PANN_MIPS, false); }
#undef CALL_MFMA
  }
#undef CALL_DENSE
  PANN_HIP(hipGetLastError());
  hipLaunchKernelGGL(dense_merge_kernel, dim3((uint32_t)na), dim3(64), (size_t)mcap * 8, st,
                     (const uint64_t*)ws.buf, na, nsplit, m, mcap, d_out_ids, d_out_dists);
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

// pair / query-vs-ids distances (Point::distance): one group of LPC lanes per pair
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) query_distances_kernel(PointsView pv, uint32_t dbytes, const uint8_t* q_ext,
                                                                    uint64_t q_stride, const uint32_t* q_ids,
                                                                    const uint32_t* ids, uint64_t m, int paired,
                                                                    float* out) {
  // paired: out[i] = d(q_ids[i], ids[i]) -- one wave per 64 pairs is wasteful; one wave per QUERY otherwise
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  const uint64_t qi = blockIdx.x;
  const uint8_t* qrow = q_ids ? pv.points + (uint64_t)q_ids[qi] * pv.pstride : q_ext + qi * q_stride;
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(qrow, q_ids ? pv.pstride : dbytes, pv.nch, qreg, qlds, lane);
  __syncthreads();
  const uint64_t j_lo = paired ? qi : 0, j_hi = paired ? qi + 1 : m;
  for (uint64_t j0 = j_lo; j0 < j_hi; j0 += PANN_WAVE) {
    const uint32_t mm = (uint32_t)min((uint64_t)PANN_WAVE, j_hi - j0);
    if (lane < (int)mm) Pl[lane] = ids[j0 + lane];
    __syncthreads();
    gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
      [&](bool has, uint32_t ci, uint32_t, float dist) {
        if (has) out[paired ? qi : qi * m + j0 + ci] = dist;
      });
    __syncthreads();
  }
}

int query_distances_dev(const DeviceIndex& ix, hipStream_t st, const uint8_t* d_q_ext, uint64_t q_stride,
                        const uint32_t* d_q_ids, uint64_t nq, const uint32_t* d_ids, uint64_t m, int paired,
                        float* d_out) {
  if (nq == 0) return PANN_OK;
  const PointsView pv{ix.points, ix.pstride, ix.nch, ix.exact};
  const size_t qb = query_lds_bytes(ix);
#define CALL_QD(DT, MT, L, N1) hipLaunchKernelGGL((query_distances_kernel<DT, MT, L, N1>), dim3((uint32_t)nq), dim3(PANN_WAVE), qb, st, pv, ix.dbytes, d_q_ext, q_stride, d_q_ids, d_ids, m, paired, d_out)
  PANN_TYPE_SWITCH(ix, CALL_QD);
#undef CALL_QD
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

// HCNNG two-pivot split (clusterEdge.h:66-83): one wave per tile of <=64 cluster members; the two
// pivots are the wave's queries in turn, the members are the gathered candidates.
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) pivot_split_kernel(PointsView pv, uint32_t dbytes, const uint32_t* ids,
                                                                const uint32_t* tile_seg, const uint64_t* tile_lo,
                                                                const uint32_t* tile_cnt, const uint32_t* pivot_a,
                                                                const uint32_t* pivot_b, uint8_t* out_side) {
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  __shared__ float Da[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  const uint32_t t = blockIdx.x;
  const uint32_t seg = tile_seg[t];
  const uint64_t lo = tile_lo[t];
  const uint32_t mm = tile_cnt[t];
  if (lane < (int)mm) Pl[lane] = ids[lo + lane];
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(pv.points + (uint64_t)pivot_a[seg] * pv.pstride, pv.pstride, pv.nch, qreg, qlds, lane);
  __syncthreads();
  gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
    [&](bool has, uint32_t ci, uint32_t, float dist) { if (has) Da[ci] = dist; });
  __syncthreads();
  load_query<DT, LPC, NCH1>(pv.points + (uint64_t)pivot_b[seg] * pv.pstride, pv.pstride, pv.nch, qreg, qlds, lane);
  __syncthreads();
  gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
    [&](bool has, uint32_t ci, uint32_t, float dist) { if (has) out_side[lo + ci] = (Da[ci] <= dist) ? 0 : 1; });
}

int pivot_split_dev(const DeviceIndex& ix, hipStream_t st, const uint32_t* d_ids, const uint32_t* d_tile_seg,
                    const uint64_t* d_tile_lo, const uint32_t* d_tile_cnt, uint32_t ntiles, const uint32_t* d_pa,
                    const uint32_t* d_pb, uint8_t* d_side) {
  if (ntiles == 0) return PANN_OK;
  const PointsView pv{ix.points, ix.pstride, ix.nch, ix.exact};
  const size_t qb = query_lds_bytes(ix);
#define CALL_PS(DT, MT, L, N1) hipLaunchKernelGGL((pivot_split_kernel<DT, MT, L, N1>), dim3(ntiles), dim3(PANN_WAVE), qb, st, pv, ix.dbytes, d_ids, d_tile_seg, d_tile_lo, d_tile_cnt, d_pa, d_pb, d_side)
  PANN_TYPE_SWITCH(ix, CALL_PS);
#undef CALL_PS
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

// beam_search_rerank's re-scoring (beamSearch.h:426-452): one wave per query
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) rerank_kernel(PointsView pv, uint32_t dbytes, const uint8_t* q_ext,
                                                           uint64_t q_stride, const uint32_t* cand_ids, uint32_t c,
                                                           const uint32_t* cand_counts, uint32_t k, int resort,
                                                           uint32_t* out_ids, float* out_dists) {
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint64_t* K = reinterpret_cast<uint64_t*>(smem);                  // [c] keys
  uint4* qlds = reinterpret_cast<uint4*>(K + ((c + 1) & ~1u));
  const uint64_t qi = blockIdx.x;
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(q_ext + qi * q_stride, dbytes, pv.nch, qreg, qlds, lane);
  __syncthreads();
  const uint32_t cn = cand_counts ? min(cand_counts[qi], c) : c;
  const uint32_t* ids = cand_ids + qi * c;
  for (uint32_t j0 = 0; j0 < cn; j0 += PANN_WAVE) {
    const uint32_t mm = min(cn - j0, (uint32_t)PANN_WAVE);
    if (lane < (int)mm) Pl[lane] = ids[j0 + lane];
    __syncthreads();
    gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
      [&](bool has, uint32_t ci, uint32_t id, float dist) { if (has) K[j0 + ci] = make_key(dist, id); });
    __syncthreads();
  }
  for (uint32_t j0 = 0; j0 < max(cn, k); j0 += PANN_WAVE) {
    const uint32_t j = j0 + lane;
    if (j < cn) {
      const uint64_t key = K[j];
      uint32_t r = j;
      if (resort) { r = 0; for (uint32_t i = 0; i < cn; i++) { const uint64_t o = K[i]; r += (o < key || (o == key && i < j)) ? 1u : 0u; } }
      if (r < k) { out_ids[qi * k + r] = key_id(key); out_dists[qi * k + r] = key_dist(key); }
    } else if (j < k) {
      out_ids[qi * k + j] = SENTINEL; out_dists[qi * k + j] = __builtin_inff();
    }
  }
}

int rerank_dev(const DeviceIndex& ix, hipStream_t st, const uint8_t* d_q, uint64_t q_stride, uint64_t nq,
               const uint32_t* d_cand, uint32_t c, const uint32_t* d_cnt, uint32_t k, int resort, uint32_t* d_out_ids,
               float* d_out_dists) {
  if (nq == 0) return PANN_OK;
  if (c == 0 || c > 4096) { set_error("pann_rerank: candidates per query must be in [1,4096]"); return PANN_ERR_BAD_ARG; }
  const PointsView pv{ix.points, ix.pstride, ix.nch, ix.exact};
  const size_t lds = (size_t)((c + 1) & ~1u) * 8 + query_lds_bytes(ix);
#define CALL_RR(DT, MT, L, N1) hipLaunchKernelGGL((rerank_kernel<DT, MT, L, N1>), dim3((uint32_t)nq), dim3(PANN_WAVE), lds, st, pv, ix.dbytes, d_q, q_stride, d_cand, c, d_cnt, k, resort, d_out_ids, d_out_dists)
  PANN_TYPE_SWITCH(ix, CALL_RR);
#undef CALL_RR
  PANN_HIP(hipGetLastError());
  return PANN_OK;
}

}  // namespace pann

#ifdef PANN_GT_COUNTERS
extern "C" int pann_debug_gt_counters(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pann::gt_counters), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(pann::gt_counters), z, 64) != hipSuccess) return -1; }
  return 0;
}
#endif

