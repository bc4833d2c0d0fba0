// pann_device.h -- device helpers shared by the gfx950 kernels: the visited-filter hash, the
// (dist,id) sort key, DPP lane-group reductions and the per-dtype distance accumulators.
#pragma once
#include "pann_internal.h"

namespace pann {

#define PANN_WAVE 64

// parlay::hash64_2 as called by has_been_seen (beamSearch.h:55); kept in ONE place per SURVEY
// section 8c so it can be corrected if upstream parlaylib differs.
__device__ __forceinline__ uint64_t hash64_2(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  x = x ^ (x >> 31);
  return x;
}

// ---- (dist,id) total order as one 64-bit key: beamSearch.h:46-48 ----
// float -> uint32 that orders like the float (after -0.0 -> +0.0 so that keys of equal floats are
// equal); the id breaks ties in the low word.
__device__ __forceinline__ uint32_t f2ord(float d) {
  d = d + 0.0f;
  uint32_t u = __float_as_uint(d);
  return u ^ (((uint32_t)((int32_t)u >> 31)) | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
  return ((uint64_t)f2ord(d) << 32) | id;
}
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)k; }
__device__ __forceinline__ float key_dist(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
constexpr uint64_t KEY_INF = 0xFFFFFFFFFFFFFFFFull;

// ---- DPP butterfly sum over aligned groups of LPC consecutive lanes; every lane of the group
// ends with the group total (fp adds are commutative, so all lanes hold the same bits) ----
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ int xlane_add(int a, int b) { return a + b; }
__device__ __forceinline__ float xlane_add(float a, float b) { return a + b; }

template <int CTRL>
__device__ __forceinline__ int dpp_step(int v) { return v + dpp_mov<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
  return v + __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
}

template <int LPC, typename T>
__device__ __forceinline__ T group_sum(T v) {
  if (LPC >= 2) v = dpp_step<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  if (LPC >= 4) v = dpp_step<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  if (LPC >= 8) v = dpp_step<0x141>(v);   // row_half_mirror      : pairs the two quads of 8
  if (LPC >= 16) v = dpp_step<0x140>(v);  // row_mirror           : pairs the two halves of 16
  if (LPC >= 32) v = xlane_add(v, __shfl_xor(v, 16));
  if (LPC >= 64) v = xlane_add(v, __shfl_xor(v, 32));
  return v;
}

// ---- distance accumulators.  One call consumes 16 bytes of a base row and the matching 16
// bytes of the query.  Integer types are exact in int32 (euclidian_point.h:54-62,74-81;
// mips_point.h:43-57); float types accumulate in f32 (lane-partial sums then a butterfly: the
// summation ORDER differs from the CPU's left-to-right loop, see DESIGN.md "float order"). ----
template <int DT>
struct AccT { using type = int; };
template <>
struct AccT<PANN_F32> { using type = float; };
template <>
struct AccT<PANN_F16> { using type = float; };

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int DT, int METRIC>
__device__ __forceinline__ void dist_accum(typename AccT<DT>::type& acc, const uint4& a, const uint4& q) {
  if constexpr (DT == PANN_U8) {
    // L2: sum (a-q)^2 = a.a - 2 a.q + q.q, all three as packed 4x8-bit dot products (exact)
    const uint32_t av[4] = {a.x, a.y, a.z, a.w}, qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (METRIC == PANN_L2) {
        uint32_t aa = __builtin_amdgcn_udot4(av[i], av[i], 0u, false);
        uint32_t aq = __builtin_amdgcn_udot4(av[i], qv[i], 0u, false);
        uint32_t qq = __builtin_amdgcn_udot4(qv[i], qv[i], 0u, false);
        acc += (int)(aa + qq - 2u * aq);
      } else {
        acc += (int)__builtin_amdgcn_udot4(av[i], qv[i], 0u, false);
      }
    }
  } else if constexpr (DT == PANN_I8) {
    const int av[4] = {(int)a.x, (int)a.y, (int)a.z, (int)a.w}, qv[4] = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (METRIC == PANN_L2) {
        int aa = __builtin_amdgcn_sdot4(av[i], av[i], 0, false);
        int aq = __builtin_amdgcn_sdot4(av[i], qv[i], 0, false);
        int qq = __builtin_amdgcn_sdot4(qv[i], qv[i], 0, false);
        acc += aa + qq - 2 * aq;
      } else {
        acc += __builtin_amdgcn_sdot4(av[i], qv[i], 0, false);
      }
    }
  } else if constexpr (DT == PANN_F32) {
    const float av[4] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
    const float qv[4] = {__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (METRIC == PANN_L2) { float t = qv[i] - av[i]; acc = fmaf(t, t, acc); }
      else acc = fmaf(qv[i], av[i], acc);
    }
  } else {  // PANN_F16: halves are widened to f32 first (exact), arithmetic is f32
    half8 ah, qh;
    __builtin_memcpy(&ah, &a, 16);
    __builtin_memcpy(&qh, &q, 16);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float af = (float)ah[i], qf = (float)qh[i];
      if constexpr (METRIC == PANN_L2) { float t = qf - af; acc = fmaf(t, t, acc); }
      else acc = fmaf(qf, af, acc);
    }
  }
}

// final conversion to the reference's float distanceType (cast once for integers, negate for MIPS)
template <int DT, int METRIC>
__device__ __forceinline__ float dist_finish(typename AccT<DT>::type acc) {
  float f = (float)acc;
  if constexpr (METRIC == PANN_MIPS) f = -f;
  return f;
}

// 16 bytes of a row; bytes at or beyond `valid` read as zero and are not touched in memory
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* row, uint32_t off, uint32_t valid) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (off + 16 <= valid) {
    v = *reinterpret_cast<const uint4*>(row + off);
  } else if (off < valid) {
    uint8_t tmp[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tmp[i] = (off + i < valid) ? row[off + i] : (uint8_t)0;
    __builtin_memcpy(&v, tmp, 16);
  }
  return v;
}

// ---- gather-distance tile: distances from ONE query vector (registers qreg when the row is a
// single chunk, else LDS qlds) to the m ids in Pl[0..m).  LPC lanes share a candidate, each reads
// 16 B per chunk, so one load instruction covers 64/LPC whole row segments; U candidates-groups
// are in flight per lane before the first use.  emit(has, ci, id, dist) is called by ALL lanes
// (uniform control flow); `has` is true on the first lane of each candidate group. ----
struct PointsView { const uint8_t* points; uint32_t pstride; uint32_t nch; };

template <int DT, int METRIC, int LPC, bool NCH1, int U, typename Emit>
__device__ __forceinline__ void gather_tile(const PointsView& PV, const uint4& qreg, const uint4* qlds,
                                            const uint32_t* Pl, uint32_t m, int lane, Emit&& emit) {
  using acc_t = typename AccT<DT>::type;
  constexpr int G = PANN_WAVE / LPC;
  const int grp = lane / LPC, sub = lane % LPC;
  for (uint32_t s0 = 0; s0 < m; s0 += G * U) {
    acc_t acc[U];
    uint32_t ids[U];
    if constexpr (NCH1) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t ci = s0 + u * G + grp;
        ids[u] = ci < m ? Pl[ci] : SENTINEL;
        v[u] = make_uint4(0, 0, 0, 0);
        if (ci < m) v[u] = *reinterpret_cast<const uint4*>(PV.points + (uint64_t)ids[u] * PV.pstride + sub * 16);
      }
#pragma unroll
      for (int u = 0; u < U; u++) { acc[u] = 0; dist_accum<DT, METRIC>(acc[u], v[u], qreg); }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t ci = s0 + u * G + grp;
        ids[u] = ci < m ? Pl[ci] : SENTINEL;
        acc[u] = 0;
      }
      for (uint32_t ch = 0; ch < PV.nch; ch++) {
        uint4 v[U];
        const uint4 qv = qlds[ch * LPC + sub];
#pragma unroll
        for (int u = 0; u < U; u++) {
          v[u] = make_uint4(0, 0, 0, 0);
          if (ids[u] != SENTINEL)
            v[u] = *reinterpret_cast<const uint4*>(PV.points + (uint64_t)ids[u] * PV.pstride + (ch * LPC + sub) * 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++) dist_accum<DT, METRIC>(acc[u], v[u], qv);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const acc_t tot = group_sum<LPC>(acc[u]);
      const float dist = dist_finish<DT, METRIC>(tot);
      emit((sub == 0) && (ids[u] != SENTINEL), s0 + u * G + grp, ids[u], dist);
    }
  }
}

// load one row (device layout, or an external query row of `valid` bytes) as the wave's query:
// registers when NCH1, else LDS qlds[nch*LPC].  Caller syncs before using qlds.
template <int LPC, bool NCH1>
__device__ __forceinline__ void load_query(const uint8_t* qrow, uint32_t valid, uint32_t nch, uint4& qreg,
                                           uint4* qlds, int lane) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(qrow) & 15) == 0);
  auto fetch = [&](uint32_t j) -> uint4 {
    if (aligned) return load16_guarded(qrow, j * 16, valid);
    uint8_t tmp[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tmp[i] = (j * 16 + i < valid) ? qrow[j * 16 + i] : (uint8_t)0;
    uint4 v; __builtin_memcpy(&v, tmp, 16); return v;
  };
  if constexpr (NCH1) {
    qreg = fetch(lane % LPC);
  } else {
    for (uint32_t j = lane; j < nch * LPC; j += PANN_WAVE) qlds[j] = fetch(j);
  }
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t m, int lane) {
  return __popcll(m & ((1ull << lane) - 1ull));
}

// dispatch helper shared by the launchers: calls F.template run<DT,METRIC,LPC,NCH1>() for the index
#define PANN_LAYOUT_SWITCH(ix, DT, MT, CALL)                                   \
  do {                                                                          \
    if ((ix).nch == 1 && (ix).lpc == 8) { CALL(DT, MT, 8, true); }              \
    else if ((ix).nch == 1 && (ix).lpc == 16) { CALL(DT, MT, 16, true); }       \
    else if ((ix).nch == 1 && (ix).lpc == 32) { CALL(DT, MT, 32, true); }       \
    else if ((ix).lpc == 4) { CALL(DT, MT, 4, false); }                         \
    else { CALL(DT, MT, 16, false); }                                           \
  } while (0)
#define PANN_TYPE_SWITCH(ix, CALL)                                                             \
  do {                                                                                          \
    if ((ix).dtype == PANN_U8 && (ix).metric == PANN_L2) PANN_LAYOUT_SWITCH(ix, PANN_U8, PANN_L2, CALL);        \
    else if ((ix).dtype == PANN_U8 && (ix).metric == PANN_MIPS) PANN_LAYOUT_SWITCH(ix, PANN_U8, PANN_MIPS, CALL);  \
    else if ((ix).dtype == PANN_I8 && (ix).metric == PANN_L2) PANN_LAYOUT_SWITCH(ix, PANN_I8, PANN_L2, CALL);    \
    else if ((ix).dtype == PANN_I8 && (ix).metric == PANN_MIPS) PANN_LAYOUT_SWITCH(ix, PANN_I8, PANN_MIPS, CALL);  \
    else if ((ix).dtype == PANN_F32 && (ix).metric == PANN_L2) PANN_LAYOUT_SWITCH(ix, PANN_F32, PANN_L2, CALL);  \
    else if ((ix).dtype == PANN_F32 && (ix).metric == PANN_MIPS) PANN_LAYOUT_SWITCH(ix, PANN_F32, PANN_MIPS, CALL); \
    else if ((ix).dtype == PANN_F16 && (ix).metric == PANN_L2) PANN_LAYOUT_SWITCH(ix, PANN_F16, PANN_L2, CALL);  \
    else PANN_LAYOUT_SWITCH(ix, PANN_F16, PANN_MIPS, CALL);                                     \
  } while (0)

}  // namespace pann
