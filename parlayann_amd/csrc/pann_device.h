// pann_device.h -- device helpers shared by the gfx950 kernels: the visited-filter hash, the
// (dist,id) sort key, DPP lane-group reductions and the per-dtype distance accumulators.
#pragma once
#include <type_traits>
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

// the same butterfly for U independent values, step by step across all of them: a DPP operand must have been
// written two cycles earlier, and the other chains fill those slots (the one-value form costs an s_nop per step)
template <int LPC, int U, typename T>
__device__ __forceinline__ void group_sum_multi(T (&v)[U]) {
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 2) v[u] = dpp_step<0xB1>(v[u]);
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 4) v[u] = dpp_step<0x4E>(v[u]);
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 8) v[u] = dpp_step<0x141>(v[u]);
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 16) v[u] = dpp_step<0x140>(v[u]);
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 32) v[u] = xlane_add(v[u], __shfl_xor(v[u], 16));
#pragma unroll
  for (int u = 0; u < U; u++) if (LPC >= 64) v[u] = xlane_add(v[u], __shfl_xor(v[u], 32));
}

// ---- distance accumulators.  One dist_accum call consumes 16 bytes of a base row and the
// matching 16 bytes of the query (held pre-digested in a QReg).  Integer types are exact in int32
// (euclidian_point.h:54-62,74-81; mips_point.h:43-57); float types accumulate in f32: each lane
// keeps two partial sums (even / odd elements), lanes are then combined by a butterfly, so the
// summation ORDER differs from the CPU's left-to-right loop (DESIGN.md "float order"). ----
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

template <int DT> struct QReg;          // 16 query bytes, converted once per query
template <> struct QReg<PANN_U8>  { uint4 raw; uint32_t qq; };   // qq = sum of squares of these 16 bytes
template <> struct QReg<PANN_I8>  { uint4 raw; uint32_t qq; };
template <> struct QReg<PANN_F32> { float f[4]; };
template <> struct QReg<PANN_F16> { uint4 raw; };        // halves stay packed; widened inside v_fma_mix
template <> struct QReg<PANN_BF16> { float f[8]; };      // widened once per query (a shift / a mask per element)

template <int DT> struct Acc;           // per-lane partial state
template <> struct Acc<PANN_U8>  { uint32_t aa, aq; __device__ __forceinline__ void clear() { aa = 0; aq = 0; } };
template <> struct Acc<PANN_I8>  { int aa, aq;      __device__ __forceinline__ void clear() { aa = 0; aq = 0; } };
template <> struct Acc<PANN_F32> { float2v s;       __device__ __forceinline__ void clear() { s = float2v{0.f, 0.f}; } };
template <> struct Acc<PANN_F16> { float2v s;       __device__ __forceinline__ void clear() { s = float2v{0.f, 0.f}; } };
template <> struct Acc<PANN_BF16> { float2v s;      __device__ __forceinline__ void clear() { s = float2v{0.f, 0.f}; } };

template <int DT> struct AccT { using type = int; };             // type that crosses lanes
template <> struct AccT<PANN_F32> { using type = float; };
template <> struct AccT<PANN_F16> { using type = float; };
template <> struct AccT<PANN_BF16> { using type = float; };

// bfloat16 pair in a dword -> the two floats (exact: the 16 bits are the float's upper half)
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

template <int DT>
__device__ __forceinline__ QReg<DT> make_qreg(const uint4& q) {
  QReg<DT> r;
  if constexpr (DT == PANN_U8) {
    r.raw = q;
    r.qq = __builtin_amdgcn_udot4(q.x, q.x, 0u, false);
    r.qq = __builtin_amdgcn_udot4(q.y, q.y, r.qq, false);
    r.qq = __builtin_amdgcn_udot4(q.z, q.z, r.qq, false);
    r.qq = __builtin_amdgcn_udot4(q.w, q.w, r.qq, false);
  } else if constexpr (DT == PANN_I8) {
    r.raw = q;
    int t = __builtin_amdgcn_sdot4((int)q.x, (int)q.x, 0, false);
    t = __builtin_amdgcn_sdot4((int)q.y, (int)q.y, t, false);
    t = __builtin_amdgcn_sdot4((int)q.z, (int)q.z, t, false);
    t = __builtin_amdgcn_sdot4((int)q.w, (int)q.w, t, false);
    r.qq = (uint32_t)t;
  } else if constexpr (DT == PANN_F32) {
    r.f[0] = __uint_as_float(q.x); r.f[1] = __uint_as_float(q.y);
    r.f[2] = __uint_as_float(q.z); r.f[3] = __uint_as_float(q.w);
  } else if constexpr (DT == PANN_BF16) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { r.f[2 * i] = bf16_lo(w[i]); r.f[2 * i + 1] = bf16_hi(w[i]); }
  } else {
    r.raw = q;
  }
  return r;
}

// (float)half(q) - (float)half(a), both taken from the lo or the hi half of their dword: one
// v_fma_mix_f32 (f16 operands widened inside the FMA: a*(-1)+q, a single rounding of the exact
// difference, i.e. bit-identical to cvt + cvt + sub)
template <int HI>
__device__ __forceinline__ float sub_widen_f16(uint32_t a_pair, uint32_t q_pair) {
  float t;
  if constexpr (HI == 0)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a_pair), "v"(q_pair));
  else
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a_pair), "v"(q_pair));
  return t;
}
// acc + (float)half(a) * (float)half(q)   (MIPS term), same instruction
template <int HI>
__device__ __forceinline__ float fma_widen_f16(uint32_t a_pair, uint32_t q_pair, float acc) {
  float t;
  if constexpr (HI == 0)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(t) : "v"(a_pair), "v"(q_pair), "v"(acc));
  else
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(t) : "v"(a_pair), "v"(q_pair), "v"(acc));
  return t;
}

template <int DT, int METRIC>
__device__ __forceinline__ void dist_accum(Acc<DT>& acc, const uint4& a, const QReg<DT>& q) {
  if constexpr (DT == PANN_U8) {
    // L2: sum (a-q)^2 = a.a - 2 a.q + q.q as packed 4x8-bit dot products with accumulate (exact)
    acc.aq = __builtin_amdgcn_udot4(a.x, q.raw.x, acc.aq, false);
    acc.aq = __builtin_amdgcn_udot4(a.y, q.raw.y, acc.aq, false);
    acc.aq = __builtin_amdgcn_udot4(a.z, q.raw.z, acc.aq, false);
    acc.aq = __builtin_amdgcn_udot4(a.w, q.raw.w, acc.aq, false);
    if constexpr (METRIC == PANN_L2) {
      acc.aa = __builtin_amdgcn_udot4(a.x, a.x, acc.aa, false);
      acc.aa = __builtin_amdgcn_udot4(a.y, a.y, acc.aa, false);
      acc.aa = __builtin_amdgcn_udot4(a.z, a.z, acc.aa, false);
      acc.aa = __builtin_amdgcn_udot4(a.w, a.w, acc.aa, false);
      acc.aa += q.qq;
    }
  } else if constexpr (DT == PANN_I8) {
    acc.aq = __builtin_amdgcn_sdot4((int)a.x, (int)q.raw.x, acc.aq, false);
    acc.aq = __builtin_amdgcn_sdot4((int)a.y, (int)q.raw.y, acc.aq, false);
    acc.aq = __builtin_amdgcn_sdot4((int)a.z, (int)q.raw.z, acc.aq, false);
    acc.aq = __builtin_amdgcn_sdot4((int)a.w, (int)q.raw.w, acc.aq, false);
    if constexpr (METRIC == PANN_L2) {
      acc.aa = __builtin_amdgcn_sdot4((int)a.x, (int)a.x, acc.aa, false);
      acc.aa = __builtin_amdgcn_sdot4((int)a.y, (int)a.y, acc.aa, false);
      acc.aa = __builtin_amdgcn_sdot4((int)a.z, (int)a.z, acc.aa, false);
      acc.aa = __builtin_amdgcn_sdot4((int)a.w, (int)a.w, acc.aa, false);
      acc.aa += (int)q.qq;
    }
  } else if constexpr (DT == PANN_F32) {
    const float2v a01{__uint_as_float(a.x), __uint_as_float(a.y)}, a23{__uint_as_float(a.z), __uint_as_float(a.w)};
    const float2v q01{q.f[0], q.f[1]}, q23{q.f[2], q.f[3]};
    if constexpr (METRIC == PANN_L2) {
      const float2v t01 = q01 - a01, t23 = q23 - a23;
      acc.s = __builtin_elementwise_fma(t01, t01, acc.s);
      acc.s = __builtin_elementwise_fma(t23, t23, acc.s);
    } else {
      acc.s = __builtin_elementwise_fma(q01, a01, acc.s);
      acc.s = __builtin_elementwise_fma(q23, a23, acc.s);
    }
  } else if constexpr (DT == PANN_BF16) {   // widened by a shift / a mask (exact), arithmetic is f32: even / odd elements in s.x / s.y
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float2v av{bf16_lo(w[i]), bf16_hi(w[i])}, qv{q.f[2 * i], q.f[2 * i + 1]};
      if constexpr (METRIC == PANN_L2) { const float2v t = qv - av; acc.s = __builtin_elementwise_fma(t, t, acc.s); }
      else acc.s = __builtin_elementwise_fma(qv, av, acc.s);
    }
  } else {  // PANN_F16: halves widened to f32 inside the FMA (exact), arithmetic is f32
    const uint32_t w[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.raw.x, q.raw.y, q.raw.z, q.raw.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (METRIC == PANN_L2) {
        const float2v t{sub_widen_f16<0>(w[i], qw[i]), sub_widen_f16<1>(w[i], qw[i])};
        acc.s = __builtin_elementwise_fma(t, t, acc.s);
      } else {
        acc.s = float2v{fma_widen_f16<0>(w[i], qw[i], acc.s.x), fma_widen_f16<1>(w[i], qw[i], acc.s.y)};
      }
    }
  }
}

// the per-lane value that is summed across the lanes of a candidate group
template <int DT, int METRIC>
__device__ __forceinline__ typename AccT<DT>::type acc_lane_value(const Acc<DT>& acc) {
  if constexpr (DT == PANN_U8) {
    if constexpr (METRIC == PANN_L2) return (int)(acc.aa - 2u * acc.aq); else return (int)acc.aq;
  } else if constexpr (DT == PANN_I8) {
    if constexpr (METRIC == PANN_L2) return acc.aa - 2 * acc.aq; else return acc.aq;
  } else {
    return acc.s.x + acc.s.y;
  }
}

// final conversion to the reference's float distanceType (cast once for integers, negate for MIPS)
template <int DT, int METRIC>
__device__ __forceinline__ float dist_finish(typename AccT<DT>::type tot) {
  float f = (float)tot;
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
struct PointsView { const uint8_t* points; uint32_t pstride; uint32_t nch; uint32_t exact; };

// "Exact float order" (debug / validation mode, float element types only): one LANE per candidate sums
// the whole row strictly left to right with one rounding per subtract, multiply and add -- the reference's
// scalar loop (euclidian_point.h:83-90, mips_point.h:59-65) -- so results on REAL-valued data are
// bit-identical to the CPU path.  The zero padding adds (0-0)^2 = +0 terms, which change nothing.
template <int DT, int METRIC>
__device__ __forceinline__ void dist_accum_exact(float& acc, const uint4& a, const uint4& q) {
#pragma clang fp contract(off)
  if constexpr (DT == PANN_F32) {
    const float av[4] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
    const float qv[4] = {__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (METRIC == PANN_L2) { const float t = qv[i] - av[i]; const float p = t * t; acc = acc + p; }
      else { const float p = qv[i] * av[i]; acc = acc + p; }
    }
  } else if constexpr (DT == PANN_F16) {
    half8 ah, qh;
    __builtin_memcpy(&ah, &a, 16);
    __builtin_memcpy(&qh, &q, 16);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float af = (float)ah[i], qf = (float)qh[i];
      if constexpr (METRIC == PANN_L2) { const float t = qf - af; const float p = t * t; acc = acc + p; }
      else { const float p = qf * af; acc = acc + p; }
    }
  } else if constexpr (DT == PANN_BF16) {
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float af = (i & 1) ? bf16_hi(aw[i >> 1]) : bf16_lo(aw[i >> 1]);
      const float qf = (i & 1) ? bf16_hi(qw[i >> 1]) : bf16_lo(qw[i >> 1]);
      if constexpr (METRIC == PANN_L2) { const float t = qf - af; const float p = t * t; acc = acc + p; }
      else { const float p = qf * af; acc = acc + p; }
    }
  }
}
template <int DT> constexpr bool is_float_dt() { return DT == PANN_F32 || DT == PANN_F16 || DT == PANN_BF16; }


// 16 bytes of a gathered base-point row.  -DPANN_NT_ROWS (A/B builds): non-temporal, so the streamed rows do not displace
// the per-query filter tables from the XCD's L2.
__device__ __forceinline__ uint4 row_load16(const uint8_t* p) {
#ifdef PANN_NT_ROWS
  typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
  const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}

// one iteration of the gather: U groups of G = 64/LPC candidates starting at Pl[s0]
template <int DT, int METRIC, int LPC, bool NCH1, int U, typename Emit>
__device__ __forceinline__ void gather_iter(const PointsView& PV, const QReg<DT>& qreg, const uint4* qlds,
                                            const uint32_t* Pl, uint32_t m, uint32_t s0, int lane, Emit&& emit) {
  constexpr int G = PANN_WAVE / LPC;
  const int grp = lane / LPC, sub = lane % LPC;
  Acc<DT> acc[U];
  uint32_t ids[U];
  // groups past the end re-read the last candidate (a cache hit) instead of branching around the load
#pragma unroll
  for (int u = 0; u < U; u++) ids[u] = Pl[min(s0 + u * G + grp, m - 1)];
  if constexpr (NCH1) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++)
      v[u] = row_load16(PV.points + (uint64_t)ids[u] * PV.pstride + sub * 16);
#pragma unroll
    for (int u = 0; u < U; u++) { acc[u].clear(); dist_accum<DT, METRIC>(acc[u], v[u], qreg); }
  } else {
    // rows of several chunks per lane: CB chunks of every group are requested before the first is used, so a
    // row costs ceil(nch/CB) memory round trips instead of nch; a batch running past the row re-reads its last
    // chunk (a cache hit) and skips the arithmetic (wave-uniform branch)
    constexpr int CB = 3;
#pragma unroll
    for (int u = 0; u < U; u++) acc[u].clear();
    const uint8_t* rp[U];
#pragma unroll
    for (int u = 0; u < U; u++) rp[u] = PV.points + (uint64_t)ids[u] * PV.pstride + sub * 16;
    for (uint32_t ch0 = 0; ch0 < PV.nch; ch0 += CB) {
      uint4 v[CB][U];
#pragma unroll
      for (int cb = 0; cb < CB; cb++) {
        const uint32_t chx = min(ch0 + cb, PV.nch - 1);
#pragma unroll
        for (int u = 0; u < U; u++) v[cb][u] = row_load16(rp[u] + chx * (LPC * 16));
      }
#pragma unroll
      for (int cb = 0; cb < CB; cb++) {
        if (ch0 + cb < PV.nch) {
          const QReg<DT> qv = make_qreg<DT>(qlds[(ch0 + cb) * LPC + sub]);
#pragma unroll
          for (int u = 0; u < U; u++) dist_accum<DT, METRIC>(acc[u], v[cb][u], qv);
        }
      }
    }
  }
  if constexpr (std::is_invocable_v<Emit&, bool, uint32_t, uint32_t, float, uint64_t>) {
    // ONE emit for the G*U candidates of the iteration: after the butterfly every lane of a group holds the group's U
    // results, lane `sub` (< U) presents result `sub`.  The fifth argument is the set of lanes that hold a candidate
    // with a SMALLER index (u-major order), so an ordered append is position = base + popcount(pass_mask & before).
    float dsel = 0.0f; uint32_t isel = 0;
    typename AccT<DT>::type tot[U];
#pragma unroll
    for (int u = 0; u < U; u++) tot[u] = acc_lane_value<DT, METRIC>(acc[u]);
    group_sum_multi<LPC, U>(tot);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float dist = dist_finish<DT, METRIC>(tot[u]);
      if (sub == u) { dsel = dist; isel = ids[u]; }
    }
    constexpr uint64_t REP = LPC == 4 ? 0x1111111111111111ull : LPC == 8 ? 0x0101010101010101ull
                           : LPC == 16 ? 0x0001000100010001ull : LPC == 32 ? 0x0000000100000001ull : 1ull;
    const uint64_t before = REP * ((1ull << sub) - 1ull) | ((REP << sub) & ((1ull << (grp * LPC)) - 1ull));
    const uint32_t ci = s0 + (uint32_t)sub * G + grp;
    emit((sub < U) && (ci < m), ci, isel, dsel, before);
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const auto tot = group_sum<LPC>(acc_lane_value<DT, METRIC>(acc[u]));
      const float dist = dist_finish<DT, METRIC>(tot);
      const uint32_t ci = s0 + u * G + grp;
      emit((sub == 0) && (ci < m), ci, ids[u], dist);
    }
  }
}

// the last, partial iteration: as few groups as cover the remaining candidates (the kernels are bound by
// instructions issued per candidate slot, so idle groups are not free)
template <int DT, int METRIC, int LPC, bool NCH1, int U, typename Emit>
__device__ __forceinline__ void gather_span(const PointsView& PV, const QReg<DT>& qreg, const uint4* qlds,
                                            const uint32_t* Pl, uint32_t m, uint32_t s0, int lane, Emit&& emit) {
  constexpr int G = PANN_WAVE / LPC;
  if constexpr (U > 1) {
    if (m - s0 <= (uint32_t)(G * (U / 2))) {
      gather_span<DT, METRIC, LPC, NCH1, U / 2>(PV, qreg, qlds, Pl, m, s0, lane, emit);
      return;
    }
  }
  gather_iter<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, s0, lane, emit);
}

template <int DT, int METRIC, int LPC, bool NCH1, int U_, typename Emit>
__device__ __forceinline__ void gather_tile(const PointsView& PV, const QReg<DT>& qreg, const uint4* qlds,
                                            const uint32_t* Pl, uint32_t m, int lane, Emit&& emit) {
  if constexpr (is_float_dt<DT>() && !NCH1) {
    if (PV.exact) {      // lane-per-candidate, sequential sum (the whole query sits in qlds in this mode)
      const uint32_t nchunks = PV.pstride / 16;
      for (uint32_t s0 = 0; s0 < m; s0 += PANN_WAVE) {
        const uint32_t ci = s0 + lane;
        const uint32_t id = Pl[min(ci, m - 1)];
        const uint8_t* row = PV.points + (uint64_t)id * PV.pstride;
        float acc = 0.0f;
        for (uint32_t ch = 0; ch < nchunks; ch++)
          dist_accum_exact<DT, METRIC>(acc, *reinterpret_cast<const uint4*>(row + ch * 16), qlds[ch]);
        if constexpr (std::is_invocable_v<Emit&, bool, uint32_t, uint32_t, float, uint64_t>)
          emit(ci < m, ci, id, METRIC == PANN_MIPS ? -acc : acc, (1ull << lane) - 1ull);   // lane order is candidate order
        else
          emit(ci < m, ci, id, METRIC == PANN_MIPS ? -acc : acc);
      }
      return;
    }
  }
  constexpr int G = PANN_WAVE / LPC;
  constexpr int U = NCH1 ? U_ : (U_ > 1 ? U_ / 2 : 1);      // multi-chunk rows keep CB x U loads in flight per lane
  uint32_t s0 = 0;
  for (; s0 + G * U <= m; s0 += G * U) gather_iter<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, s0, lane, emit);
  if (s0 < m) gather_span<DT, METRIC, LPC, NCH1, U>(PV, qreg, qlds, Pl, m, s0, lane, emit);
}

#ifndef PANN_PRUNE_U_NCH1
#define PANN_PRUNE_U_NCH1 2      /* candidate groups in flight per lane in gather_tile_multi: one-chunk rows */
#endif
#ifndef PANN_PRUNE_U_MULTI
#define PANN_PRUNE_U_MULTI 1     /* ... rows of several chunks per lane (3 chunks of every group are requested together) */
#endif
// gather_tile for B query vectors at once: every candidate vector is fetched ONCE and scored against all of
// them (robustPrune with several speculative picks per pass).  Query b: qreg[b] when the row is one chunk per
// lane, else qlds[b * qstride4 ...].  Per (query, candidate) the arithmetic is exactly gather_tile's.
// emit(has, ci, id, d[B]) is called by all lanes; `has` on the first lane of each candidate group.
template <int DT, int METRIC, int LPC, bool NCH1, int B, typename Emit>
__device__ __forceinline__ void gather_tile_multi(const PointsView& PV, const QReg<DT> (&qreg)[B], const uint4* qlds,
                                                  uint32_t qstride4, const uint32_t* Pl, uint32_t m, int lane, Emit&& emit) {
  constexpr int G = PANN_WAVE / LPC;
  constexpr int U = NCH1 ? PANN_PRUNE_U_NCH1 : PANN_PRUNE_U_MULTI;
  const int grp = lane / LPC, sub = lane % LPC;
  for (uint32_t s0 = 0; s0 < m; s0 += G * U) {
    Acc<DT> acc[B][U];
    uint32_t ids[U];
#pragma unroll
    for (int u = 0; u < U; u++) ids[u] = Pl[min(s0 + u * G + grp, m - 1)];
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
      for (int u = 0; u < U; u++) acc[b][u].clear();
    if constexpr (NCH1) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        v[u] = *reinterpret_cast<const uint4*>(PV.points + (uint64_t)ids[u] * PV.pstride + sub * 16);
#pragma unroll
      for (int b = 0; b < B; b++)
#pragma unroll
        for (int u = 0; u < U; u++) dist_accum<DT, METRIC>(acc[b][u], v[u], qreg[b]);
    } else {
      constexpr int CB = 3;
      const uint8_t* rp[U];
#pragma unroll
      for (int u = 0; u < U; u++) rp[u] = PV.points + (uint64_t)ids[u] * PV.pstride + sub * 16;
      for (uint32_t ch0 = 0; ch0 < PV.nch; ch0 += CB) {
        uint4 v[CB][U];
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
          const uint32_t chx = min(ch0 + cb, PV.nch - 1);
#pragma unroll
          for (int u = 0; u < U; u++) v[cb][u] = *reinterpret_cast<const uint4*>(rp[u] + chx * (LPC * 16));
        }
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
          if (ch0 + cb < PV.nch) {
#pragma unroll
            for (int b = 0; b < B; b++) {
              const QReg<DT> qv = make_qreg<DT>(qlds[b * qstride4 + (ch0 + cb) * LPC + sub]);
#pragma unroll
              for (int u = 0; u < U; u++) dist_accum<DT, METRIC>(acc[b][u], v[cb][u], qv);
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float d[B];
#pragma unroll
      for (int b = 0; b < B; b++) d[b] = dist_finish<DT, METRIC>(group_sum<LPC>(acc_lane_value<DT, METRIC>(acc[b][u])));
      const uint32_t ci = s0 + u * G + grp;
      emit((sub == 0) && (ci < m), ci, ids[u], d);
    }
  }
}

// load one row (device layout, or an external query row of `valid` bytes) as the wave's query:
// registers when NCH1, else LDS qlds[nch*LPC].  Caller syncs before using qlds.
template <int DT, int LPC, bool NCH1>
__device__ __forceinline__ void load_query(const uint8_t* qrow, uint32_t valid, uint32_t nch, QReg<DT>& qreg,
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
    qreg = make_qreg<DT>(fetch(lane % LPC));
  } else {
    for (uint32_t j = lane; j < nch * LPC; j += PANN_WAVE) qlds[j] = fetch(j);
  }
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

// Memory scope for atomics / coherent loads and stores on data that ONE workgroup owns for the whole kernel (a wave's filter
// table, dropped list, candidate list, seen set, result row).  Workgroup scope is what the data needs: the access bypasses the
// CU's L1 (sc0: a slot written a moment ago must not be read back stale) and nothing has to be coherent across XCDs.  Measured
// against agent scope (sc1) on the Vamana build and the range search: the same within noise, so this is about stating the
// requirement, not about speed.  (Build with -DPANN_PRIVATE_SCOPE=__HIP_MEMORY_SCOPE_AGENT for the A/B.)
#ifndef PANN_PRIVATE_SCOPE
#define PANN_PRIVATE_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif

// Block-wide sync for kernels whose workgroup is ONE wavefront: LDS instructions of a wave execute in order, so all
// that is needed between a lane's ds_write and another lane's ds_read is that the compiler keeps the order.
// Unlike __syncthreads() this does not drain outstanding global loads / stores (s_waitcnt vmcnt(0)), so memory
// requests stay in flight across the phases of an iteration.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
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
    else if ((ix).lpc == 8) { CALL(DT, MT, 8, false); }                         \
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
    else if ((ix).dtype == PANN_F16) PANN_LAYOUT_SWITCH(ix, PANN_F16, PANN_MIPS, CALL);         \
    else if ((ix).metric == PANN_L2) PANN_LAYOUT_SWITCH(ix, PANN_BF16, PANN_L2, CALL);          \
    else PANN_LAYOUT_SWITCH(ix, PANN_BF16, PANN_MIPS, CALL);                                    \
  } while (0)

}  // namespace pann
