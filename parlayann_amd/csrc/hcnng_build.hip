// hcnng_build.hip -- HCNNG index construction on gfx950.
//
//   cluster::random_clustering   HCNNG/clusterEdge.h:99-144   -> level-synchronous tree: pivots chosen on the host
//        (a few thousand clusters per level), tree_split_kernel (two-pivot distance test, :71-83),
//        rocprim exclusive scan, tree_scatter_kernel (parlay::filter keeps order -> stable partition)
//   hcnng_index::MSTk            HCNNG/hcnng_index.h:134-229  -> dense_topk (dense.hip) for the 10-NN of every
//        leaf, leaf_edges_kernel + rocprim segmented sort (less_dup order :183-201), leaf_mst_kernel
//        (unique, degree-bounded Kruskal with the reference's DisjointSet :36-89, process_edges :117-131)
//
// The sequential part of Kruskal runs on lane 0 of one wave per leaf (union-find arrays in LDS for
// leaves up to 4096 members, in HBM scratch beyond); thousands of leaves run concurrently.
#include <chrono>
#include <cstring>
#include <vector>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "pann_device.h"

namespace pann {

__device__ __forceinline__ uint64_t hc_mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
static inline uint64_t hc_mix_host(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

struct SplitCluster { uint32_t lo, len, pa, pb; };   // pa/pb: POSITIONS of the two pivots in the ids array

// tile t of a level belongs to the cluster c with tile_base[c] <= t < tile_base[c+1] (clusters in position order,
// ceil(len/64) tiles each); the host uploads only the ncl+1 prefix sums
__device__ __forceinline__ uint32_t tile_cluster(const uint32_t* tile_base, uint32_t ncl, uint32_t t) {
  uint32_t lo = 0, hi = ncl;            // invariant: tile_base[lo] <= t < tile_base[hi]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile_base[mid] <= t) lo = mid; else hi = mid; }
  return lo;
}

// side[pos] = 0 when d(ids[pos], pivot_a) <= d(ids[pos], pivot_b)   (clusterEdge.h:71-83); also flags
// clusters whose two pivot vectors are identical (:107)
template <int DT, int METRIC, int LPC, bool NCH1>
__global__ void __launch_bounds__(PANN_WAVE) tree_split_kernel(PointsView pv, uint32_t dbytes, const uint32_t* ids,
                                                               const SplitCluster* cl, const uint32_t* tile_base,
                                                               uint32_t ncl, uint32_t* is_first,
                                                               uint32_t* same_flag) {
  const int lane = threadIdx.x;
  __shared__ uint32_t Pl[PANN_WAVE];
  __shared__ float Da[PANN_WAVE];
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* qlds = reinterpret_cast<uint4*>(smem);
  const uint32_t t = blockIdx.x;
  const uint32_t ci = tile_cluster(tile_base, ncl, t);
  const SplitCluster c = cl[ci];
  const uint32_t off = (t - tile_base[ci]) * PANN_WAVE;
  const uint32_t mm = min(c.len - off, (uint32_t)PANN_WAVE);
  const uint32_t ida = ids[c.pa], idb = ids[c.pb];
  if (lane < (int)mm) Pl[lane] = ids[c.lo + off + lane];
  if (off == 0) {   // first tile of the cluster: are the two pivot vectors identical?
    const uint8_t* ra = pv.points + (uint64_t)ida * pv.pstride;
    const uint8_t* rb = pv.points + (uint64_t)idb * pv.pstride;
    bool diff = false;
    for (uint32_t b = lane * 16; b < pv.pstride; b += PANN_WAVE * 16) {
      const uint4 x = *reinterpret_cast<const uint4*>(ra + b), y = *reinterpret_cast<const uint4*>(rb + b);
      diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    const uint64_t dm = __ballot(diff);
    if (lane == 0) same_flag[ci] = dm ? 0u : 1u;
  }
  QReg<DT> qreg{};
  load_query<DT, LPC, NCH1>(pv.points + (uint64_t)ida * pv.pstride, pv.pstride, pv.nch, qreg, qlds, lane);
  __syncthreads();
  gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
    [&](bool has, uint32_t ci, uint32_t, float dist) { if (has) Da[ci] = dist; });
  __syncthreads();
  load_query<DT, LPC, NCH1>(pv.points + (uint64_t)idb * pv.pstride, pv.pstride, pv.nch, qreg, qlds, lane);
  __syncthreads();
  gather_tile<DT, METRIC, LPC, NCH1, 4>(pv, qreg, qlds, Pl, mm, lane,
    [&](bool has, uint32_t ci, uint32_t, float dist) { if (has) is_first[c.lo + off + ci] = (Da[ci] <= dist) ? 1u : 0u; });
  (void)dbytes;
}

__global__ void cluster_counts_kernel(const uint32_t* scan0, const SplitCluster* cl, uint32_t ncl, uint32_t* n0_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncl) n0_out[i] = scan0[cl[i].lo + cl[i].len] - scan0[cl[i].lo];
}

struct ScatterCluster { uint32_t lo, len, n0, half; };
// stable partition of every splitting cluster: first-side members keep their order at [lo, lo+n0),
// the others at [lo+n0, lo+len); "halved" clusters (:108-115) and finished clusters do not move
__global__ void tree_scatter_kernel(const uint32_t* ids, uint32_t* newids, const uint32_t* is_first, const uint32_t* scan0,
                                    const ScatterCluster* cl, const uint32_t* tile_base, uint32_t ncl) {
  const uint32_t t = blockIdx.x;
  const uint32_t ci = tile_cluster(tile_base, ncl, t);
  const ScatterCluster c = cl[ci];
  const uint32_t i = (t - tile_base[ci]) * PANN_WAVE + threadIdx.x;
  if (i >= c.len) return;
  const uint32_t pos = c.lo + i;
  uint32_t np = pos;
  if (!c.half) {
    const uint32_t r0 = scan0[pos] - scan0[c.lo];         // first-side members before me in my cluster
    np = is_first[pos] ? c.lo + r0 : c.lo + c.n0 + (i - r0);
  }
  newids[np] = ids[pos];
}

__global__ void fill_u32(uint32_t* p, uint64_t n, uint32_t v) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void iota_u32(uint32_t* p, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
__global__ void iota_mod_u32(uint32_t* p, uint64_t total, uint32_t n) {      // ids of every tree of a forest: 0..n-1 repeated
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < total) p[i] = (uint32_t)(i % n);
}
__global__ void invert_perm(const uint32_t* ids, uint32_t* pos, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) pos[ids[i]] = (uint32_t)i;
}
__global__ void degree_init_kernel(const uint32_t* graph, uint32_t gstride, uint32_t* deg, uint64_t n) {
  const uint64_t v = blockIdx.x;
  if (v >= n) return;
  const uint32_t* row = graph + v * gstride;
  uint32_t d = 0;
  for (uint32_t i0 = 0; i0 < gstride; i0 += 64) {
    const uint32_t i = i0 + threadIdx.x;
    d += __popcll(__ballot(i < gstride && row[i] != SENTINEL));
  }
  if (threadIdx.x == 0) deg[v] = d;
}

// edge keys of one leaf member: (ord(dist) << 32) | (min(i,j) << 16) | max(i,j), local indices (:160-170)
__global__ void leaf_edges_kernel(const uint32_t* nn_ids, const float* nn_d, const uint32_t* pos, const uint32_t* leaf_lo_of,
                                  uint32_t m, uint64_t total, uint64_t* keys) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= total * m) return;
  const uint64_t p = t / m;                    // position of the member in the ids array
  const uint32_t nb = nn_ids[t];
  uint64_t key = KEY_INF;
  if (nb != SENTINEL) {
    const uint32_t lo = leaf_lo_of[p];
    const uint32_t i = (uint32_t)(p - lo), j = pos[nb] - lo;
    key = ((uint64_t)f2ord(nn_d[t]) << 32) | ((uint64_t)min(i, j) << 16) | max(i, j);
  }
  keys[t] = key;
}
__global__ void leaf_lo_kernel(const uint64_t* leaf_off, uint32_t nleaves, uint32_t* leaf_lo_of) {
  const uint32_t l = blockIdx.x;
  if (l >= nleaves) return;
  for (uint64_t p = leaf_off[l] + threadIdx.x; p < leaf_off[l + 1]; p += blockDim.x) leaf_lo_of[p] = (uint32_t)leaf_off[l];
}

struct MstArgs {
  const uint64_t* keys;            // sorted per leaf: [leaf_off[l]*m, leaf_off[l+1]*m)
  const uint64_t* leaf_off; uint32_t nleaves; uint32_t m;
  const uint32_t* ids;             // position -> vertex id
  uint32_t* graph; uint32_t gstride; uint32_t max_deg; uint32_t* deg;
  uint32_t mst_deg;
  uint32_t lds_cap;                // leaves up to this many members keep their Kruskal state in LDS
  int32_t* g_parent; uint8_t* g_rank; uint8_t* g_degree;    // HBM scratch indexed by position (large leaves)
};

// modified Kruskal of one leaf (hcnng_index.h:202-228) + process_edges (:117-131).
// One wave per leaf.  Everything the sequential loop touches sits in LDS for leaves up to A.lds_cap members
// (union-find arrays, the members' vertex ids and their degrees before this tree); the sorted edge keys are
// fetched 64 at a time (one coalesced load per chunk, then wave-uniform readlane), so the walk never waits for
// HBM; only the row appends are global stores.  Larger leaves keep the arrays in HBM scratch.
__global__ void __launch_bounds__(PANN_WAVE) leaf_mst_kernel(MstArgs A) {
  const int lane = threadIdx.x;
  const uint32_t l = blockIdx.x;
  extern __shared__ __align__(16) uint8_t smem[];
  const uint64_t lo = A.leaf_off[l];
  const uint32_t N = (uint32_t)(A.leaf_off[l + 1] - lo);
  if (N < 2) return;
  const uint32_t cap = A.lds_cap;
  const bool in_lds = N <= cap;
  int32_t* parent = in_lds ? reinterpret_cast<int32_t*>(smem) : A.g_parent + lo;
  uint32_t* lids = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 4);                  // [cap] vertex ids (LDS mode)
  uint16_t* deg0 = reinterpret_cast<uint16_t*>(smem + (size_t)cap * 8);                  // [cap] row degree before this tree
  uint8_t* rnk = in_lds ? reinterpret_cast<uint8_t*>(smem + (size_t)cap * 10) : A.g_rank + lo;
  uint8_t* dgr = in_lds ? reinterpret_cast<uint8_t*>(smem + (size_t)cap * 11) : A.g_degree + lo;
  if (in_lds) {
    for (uint32_t i = lane; i < N; i += PANN_WAVE) {
      parent[i] = (int32_t)i; rnk[i] = 0; dgr[i] = 0;
      const uint32_t v = A.ids[lo + i];
      lids[i] = v;
      deg0[i] = (uint16_t)min(A.deg[v], 65535u);
    }
  } else {
    for (uint32_t i = lane; i < N; i += PANN_WAVE) {
      __hip_atomic_store(parent + i, (int32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rnk + i, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dgr + i, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
  }
  __threadfence_block();
  __syncthreads();
  const uint64_t* K = A.keys + lo * A.m;
  const uint64_t ne = (uint64_t)N * A.m;
  // accessors: LDS for leaves up to lds_cap members; HBM scratch (L1-bypassing loads/stores) beyond
  auto ldp = [&](int i) -> int { return in_lds ? parent[i] : __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto stp = [&](int i, int v) { if (in_lds) parent[i] = v; else __hip_atomic_store(parent + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld8 = [&](uint8_t* a, int i) -> uint8_t { return in_lds ? a[i] : __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st8 = [&](uint8_t* a, int i, uint8_t v) { if (in_lds) a[i] = v; else __hip_atomic_store(a + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto find = [&](int x) { for (;;) { const int px = ldp(x); if (px == x) return x; const int gp = ldp(px); stp(x, gp); x = gp; } };
  auto root_ro = [&](int x) { for (;;) { const int px = ldp(x); if (px == x) return x; x = px; } };
  uint64_t prev = KEY_INF;
  uint64_t e_unique = 0;      // index in the de-duplicated sequence (remove_duplicates_ordered :202-203)
  bool done = false;
  for (uint64_t e0 = 0; e0 < ne && !done; e0 += PANN_WAVE) {
    const uint64_t kv = (e0 + lane < ne) ? K[e0 + lane] : KEY_INF;       // 64 sorted keys, one per lane
    uint64_t before = __shfl_up(kv, 1);
    if (lane == 0) before = prev;
    const bool uniq = (kv != KEY_INF) && (kv != before);                 // padding sorts last
    const int a = (int)((kv >> 16) & 0xFFFF), b = (int)(kv & 0xFFFF);
    // Read-only pre-filter, all lanes at once.  Degrees only grow and sets only merge, so an edge that fails
    // the degree or the same-set test NOW also fails it when its turn comes; the sequential walk below only
    // visits the others (and re-tests them against the current state).
    bool maybe = uniq;
    if (maybe) {
      if (ld8(dgr, a) >= A.mst_deg || ld8(dgr, b) >= A.mst_deg) maybe = false;
      else if (root_ro(a) == root_ro(b)) maybe = false;
    }
    const uint64_t um = __ballot(uniq), mm = __ballot(maybe);
    const uint64_t my_index = e_unique + lanes_below(um, lane);          // position in the de-duplicated sequence
    const uint64_t cm = __ballot(uniq && (my_index % N == 0));           // :221-225 is due after these edges
    uint64_t work = mm | cm;
    while (work) {
      const int j = __ffsll((unsigned long long)work) - 1;
      work &= work - 1;
      if ((mm >> j) & 1ull) {
        const uint64_t key = readlane64(kv, j);                          // wave-uniform
        const int ea = (int)((key >> 16) & 0xFFFF), eb = (int)(key & 0xFFFF);
        if (lane == 0) {
          const uint8_t da = ld8(dgr, ea), db = ld8(dgr, eb);
          if (da < A.mst_deg && db < A.mst_deg) {
            const int ra = find(ea), rb = find(eb);
            if (ra != rb) {
              // MST_edges gets (a,b) then (b,a); process_edges appends while the row has room.  A vertex sits in
              // exactly one leaf per tree, so its row position is (degree before this tree) + (edges of this leaf).
              const uint32_t va = in_lds ? lids[ea] : A.ids[lo + ea], vb = in_lds ? lids[eb] : A.ids[lo + eb];
              const uint32_t pa = (in_lds ? (uint32_t)deg0[ea] : A.deg[va]) + da, pb = (in_lds ? (uint32_t)deg0[eb] : A.deg[vb]) + db;
              if (pa < A.max_deg) A.graph[(size_t)va * A.gstride + pa] = vb;
              if (pb < A.max_deg) A.graph[(size_t)vb * A.gstride + pb] = va;
              st8(dgr, ea, da + 1); st8(dgr, eb, db + 1);
              const uint8_t ka = ld8(rnk, ea), kb = ld8(rnk, eb);   // the reference reads rank[x], not rank[root] (:53-54)
              if (ka < kb) stp(ra, rb); else { stp(rb, ra); if (ka == kb) st8(rnk, ra, ld8(rnk, ra) + 1); }
            }
          }
        }
      }
      if ((cm >> j) & 1ull) {   // :221-225  every N processed edges: stop when everything is in one set
        __threadfence_block();
        __syncthreads();
        const int r0 = root_ro(0);
        bool all = true;
        for (uint32_t i = lane; i < N; i += PANN_WAVE) all &= (root_ro((int)i) == r0);
        const bool full = __ballot(!all) == 0ull;
        __syncthreads();
        if (full) { done = true; break; }
      }
    }
    e_unique += __popcll(um);
    prev = readlane64(kv, PANN_WAVE - 1);
    if (prev == KEY_INF) done = true;                                    // the rest is padding
    __threadfence_block();
    __syncthreads();                                                     // lane 0's updates before the next pre-filter
  }
  __threadfence_block();
  __syncthreads();
  for (uint32_t i = lane; i < N; i += PANN_WAVE) {          // commit the rows' new degrees (capped like the appends)
    const uint32_t v = A.ids[lo + i];
    A.deg[v] = min(A.deg[v] + (uint32_t)ld8(dgr, (int)i), A.max_deg);
  }
}

// ---------------------------------------------------------------------------------------------

int dense_topk_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, const uint8_t* d_a_ext, uint64_t a_stride,
                   const uint32_t* d_a_ids, const uint32_t* d_b_ids, const uint64_t* d_a_off, const uint64_t* d_b_off,
                   const uint32_t* d_tile_seg, const uint32_t* d_tile_a0, uint32_t ntiles, uint64_t na, uint64_t nb,
                   uint32_t nsplit, uint32_t m, int exclude_same, uint32_t* d_out_ids, float* d_out_dists);

__global__ void fill_u32_hc(uint32_t* p, uint64_t n, uint32_t v) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// Tree-parallel HCNNG, assembly: slabs[w] (n rows of slab_stride slots) holds rank w's trees w, w + W, w + 2W, ... in slots
// [j*mst_deg, (j+1)*mst_deg) of every row.  Row v of the graph = the edges of trees 0, 1, 2, ... in that order (the append order
// of the single-process build, hcnng_index.h:117-131), continuing after the row's current neighbours.  One wave per vertex.
__global__ void __launch_bounds__(PANN_WAVE) hcnng_assemble_kernel(uint32_t* graph, uint32_t gstride, uint32_t max_deg, const uint32_t* slabs,
                                                                   uint64_t n, uint32_t W, uint32_t slab_stride, uint32_t ntrees,
                                                                   uint32_t mst_deg) {
  const uint64_t v = blockIdx.x;
  const int lane = threadIdx.x;
  uint32_t* row = graph + v * gstride;
  uint32_t deg = 0;
  for (uint32_t i0 = 0; i0 < gstride; i0 += PANN_WAVE) {
    const uint32_t i = i0 + lane;
    deg += __popcll(__ballot(i < gstride && row[i] != SENTINEL));
  }
  for (uint32_t t = 0; t < ntrees; t++) {
    const uint32_t* src = slabs + ((uint64_t)(t % W) * n + v) * slab_stride + (uint64_t)(t / W) * mst_deg;
    for (uint32_t i0 = 0; i0 < mst_deg; i0 += PANN_WAVE) {
      const uint32_t i = i0 + lane;
      const uint32_t a = i < mst_deg ? src[i] : SENTINEL;
      const uint64_t am = __ballot(a != SENTINEL);
      const uint32_t pos = deg + lanes_below(am, lane);
      if (a != SENTINEL && pos < max_deg) row[pos] = a;                 // process_edges: rows never exceed maxDeg (:121-124)
      deg = min(deg + (uint32_t)__popcll(am), max_deg);
    }
  }
}

int hcnng_assemble_dev(const DeviceIndex& ix, hipStream_t st, const uint32_t* d_slabs, uint32_t W, uint32_t slab_stride,
                       uint32_t ntrees, uint32_t mst_deg) {
  if (W == 0 || (uint64_t)((ntrees + W - 1) / W) * mst_deg > slab_stride) { set_error("pann_hcnng_assemble_dev: slab rows too short"); return PANN_ERR_BAD_ARG; }
  hipLaunchKernelGGL(hcnng_assemble_kernel, dim3((uint32_t)ix.n), dim3(PANN_WAVE), 0, st, ix.graph, ix.gstride, ix.max_deg, d_slabs,
                     ix.n, W, slab_stride, ntrees, mst_deg);
  PANN_HIP(hipGetLastError());
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

struct HBuf {   // small RAII device buffer
  void* p = nullptr;
  ~HBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : 1; }
  template <typename T> T* as() { return (T*)p; }
};
// The builder's scratch arrays are carved out of ONE allocation: at 10M points the forest needs ~6 GB in twenty arrays, and
// twenty hipMalloc / hipFree pairs of that size cost more than the tree phase itself.
struct ABuf {
  void* p = nullptr;
  template <typename T> T* as() { return (T*)p; }
};
struct Arena {
  HBuf mem;
  std::vector<std::pair<ABuf*, size_t>> want;
  void add(ABuf& b, size_t bytes) { want.emplace_back(&b, (std::max<size_t>(bytes, 16) + 255) / 256 * 256); }
  int commit() {
    size_t tot = 0;
    for (auto& w : want) tot += w.second;
    if (mem.alloc(tot)) return 1;
    size_t off = 0;
    for (auto& w : want) { w.first->p = static_cast<uint8_t*>(mem.p) + off; off += w.second; }
    return 0;
  }
};

// Trees first_tree, first_tree + tree_step, ... (num_clusters of them) of the forest seeded by `seed` (tree t: mix(mix(seed + t))).
// slab == nullptr: the edges are appended to the index's graph rows (pann_hcnng_build).  slab != nullptr (tree-parallel build over
// ranks): tree j of THIS call writes vertex v's edges into slots [j*mst_deg, (j+1)*mst_deg) of row v of the slab (n rows of
// slab_stride uint32, SENTINEL = empty) -- a vertex gains at most mst_deg edges per tree (hcnng_index.h:213) -- so the trees of all
// ranks can be interleaved in tree order afterwards (hcnng_assemble_dev).
int hcnng_build_dev(const DeviceIndex& ix, Workspace& ws, hipStream_t st, uint32_t num_clusters, uint32_t cluster_size,
                    uint32_t mst_deg, uint64_t seed, double* times3, uint32_t first_tree, uint32_t tree_step, uint32_t* slab,
                    uint32_t slab_stride) {
  const uint64_t n = ix.n;
  const uint32_t m = 10;                                  // hcnng_index.h:140
  if (cluster_size < 2 || cluster_size > 65535) { set_error("pann_hcnng_build: cluster_size must be in [2,65535]"); return PANN_ERR_BAD_ARG; }
  if (!slab && (uint64_t)num_clusters * mst_deg > ix.max_deg) { set_error("pann_hcnng_build: max_deg < num_clusters * mst_deg"); return PANN_ERR_BAD_ARG; }
  if (slab && (uint64_t)num_clusters * mst_deg > slab_stride) { set_error("pann_hcnng_build_trees: slab rows shorter than ntrees * mst_deg"); return PANN_ERR_BAD_ARG; }
  if (mst_deg > 255) { set_error("pann_hcnng_build: mst_deg > 255"); return PANN_ERR_BAD_ARG; }
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  const PointsView pv{ix.points, ix.pstride, ix.nch, ix.exact};
  const size_t qb = query_lds_bytes(ix);
  const uint32_t nb256 = (uint32_t)((n + 255) / 256);

  // The cluster trees are independent (clusterEdge.h:146-153): all trees of a group are split level by level
  // TOGETHER -- one ids array of group*n positions, clusters of every tree in one launch -- so a level costs one
  // host round trip for the whole forest instead of one per tree.  Positions are 32-bit: group*n < 2^31.
  uint32_t group = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(num_clusters, (1ull << 31) / std::max<uint64_t>(n, 1)));
  if (ix.forest_group) group = std::min<uint32_t>(group, ix.forest_group);   // pann_index_set_forest_group: bounds the forest's scratch
  const uint64_t gn = (uint64_t)group * n;
  ABuf b_ids, b_new, b_first, b_scan, b_pos, b_deg, b_leaflo, b_nnids, b_nnd, b_ka, b_kb, b_par, b_rnk, b_dgr, b_tmp;
  ABuf d_sc, d_tbase, d_same, d_scat, d_n0, d_loff, d_seg_b, d_seg_e;
  Arena arena;
  size_t scan_tmp = 0, sort_tmp = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)gn + 1, rocprim::plus<uint32_t>(), st);
  (void)rocprim::segmented_radix_sort_keys(nullptr, sort_tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned)(n * m), (unsigned)(n / 2 + 2),
                                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, 64, st);
  if (n * m >= 0xFFFFFFF0ull) { set_error("pann_hcnng_build: n too large for 32-bit edge offsets"); return PANN_ERR_BAD_ARG; }
  // per-level buffers, sized once for the worst level: clusters that still split are longer than cluster_size; a tree has at
  // most n / (cluster_size / 2) + 1 leaves only when every split is even, so the leaf arrays take the safe bound n + 1
  const size_t ncl_cap = (size_t)(gn / cluster_size) + 2 * group + 2;
  arena.add(b_ids, gn * 4); arena.add(b_new, gn * 4); arena.add(b_first, (gn + 1) * 4); arena.add(b_scan, (gn + 1) * 4);
  arena.add(b_pos, n * 4); arena.add(b_deg, n * 4); arena.add(b_leaflo, n * 4); arena.add(b_nnids, n * m * 4); arena.add(b_nnd, n * m * 4);
  arena.add(b_ka, n * m * 8); arena.add(b_kb, n * m * 8); arena.add(b_par, n * 4); arena.add(b_rnk, n); arena.add(b_dgr, n);
  arena.add(b_tmp, std::max(scan_tmp, sort_tmp) + 256);
  arena.add(d_sc, ncl_cap * sizeof(SplitCluster)); arena.add(d_tbase, (ncl_cap + 1) * 4); arena.add(d_same, ncl_cap * 4);
  arena.add(d_scat, ncl_cap * sizeof(ScatterCluster)); arena.add(d_n0, ncl_cap * 4);
  arena.add(d_loff, (n + 2) * 8); arena.add(d_seg_b, (n + 1) * 4); arena.add(d_seg_e, (n + 1) * 4);
  if (arena.commit()) { set_error("pann_hcnng_build: hipMalloc failed"); return PANN_ERR_HIP; }
  hipLaunchKernelGGL(degree_init_kernel, dim3((uint32_t)n), dim3(64), 0, st, ix.graph, ix.gstride, b_deg.as<uint32_t>(), n);
  PANN_HIP(hipGetLastError());

  struct Cl { uint32_t lo, len; uint64_t rnd; };
  double t_tree = 0, t_leaf = 0, t_mst = 0;
  for (uint32_t tg = 0; tg < num_clusters; tg += group) {
   const uint32_t ng = std::min(group, num_clusters - tg);            // trees tg .. tg+ng-1 in this forest
   const auto tf0 = now();
   uint32_t* fids = b_ids.as<uint32_t>();
   uint32_t* fnew = b_new.as<uint32_t>();
   std::vector<std::vector<uint64_t>> forest_leaves(ng);              // leaf start positions per tree (local to the tree)
   {
    const uint64_t tot = (uint64_t)ng * n;
    hipLaunchKernelGGL(iota_mod_u32, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, st, fids, tot, (uint32_t)n);
    std::vector<Cl> level;
    for (uint32_t j = 0; j < ng; j++)
      level.push_back(Cl{(uint32_t)(j * n), (uint32_t)n, hc_mix_host(hc_mix_host(seed + first_tree + (uint64_t)(tg + j) * tree_step))});
    while (!level.empty()) {
      std::vector<SplitCluster> sc; std::vector<Cl> src;
      for (const Cl& c : level) {
        if (c.len <= cluster_size) { forest_leaves[c.lo / n].push_back(c.lo % n); continue; }   // clusterEdge.h:103-104
        const uint64_t fi = hc_mix_host(c.rnd + 0) % c.len;                            // select_two_random :40-50
        const uint64_t su = hc_mix_host(c.rnd + 1) % (c.len - 1);
        const uint64_t si = su < fi ? su : su + 1;
        sc.push_back(SplitCluster{c.lo, c.len, (uint32_t)(c.lo + fi), (uint32_t)(c.lo + si)});
        src.push_back(c);
      }
      if (sc.empty()) break;
      const size_t ncl = sc.size();
      if (ncl > ncl_cap) { set_error("pann_hcnng_build: internal level buffer overflow"); return PANN_ERR_OVERFLOW; }
      std::vector<uint32_t> tile_base(ncl + 1);
      uint64_t ntiles = 0;
      for (size_t ci = 0; ci < ncl; ci++) { tile_base[ci] = (uint32_t)ntiles; ntiles += (sc[ci].len + 63) / 64; }
      tile_base[ncl] = (uint32_t)ntiles;
      const size_t nt = (size_t)ntiles;
      PANN_HIP(hipMemcpyAsync(d_sc.p, sc.data(), ncl * sizeof(SplitCluster), hipMemcpyHostToDevice, st));
      PANN_HIP(hipMemcpyAsync(d_tbase.p, tile_base.data(), (ncl + 1) * 4, hipMemcpyHostToDevice, st));
      PANN_HIP(hipMemsetAsync(b_first.p, 0, (tot + 1) * 4, st));
#define CALL_TS(DT, MT, L, N1) hipLaunchKernelGGL((tree_split_kernel<DT, MT, L, N1>), dim3((uint32_t)nt), dim3(PANN_WAVE), qb, st, pv, ix.dbytes, fids, d_sc.as<SplitCluster>(), d_tbase.as<uint32_t>(), (uint32_t)ncl, b_first.as<uint32_t>(), d_same.as<uint32_t>())
      PANN_TYPE_SWITCH(ix, CALL_TS);
#undef CALL_TS
      PANN_HIP(hipGetLastError());
      size_t tb = scan_tmp;
      PANN_HIP(rocprim::exclusive_scan(b_tmp.p, tb, b_first.as<uint32_t>(), b_scan.as<uint32_t>(), 0u, (size_t)tot + 1, rocprim::plus<uint32_t>(), st));
      // per-cluster first-side counts and identical-pivot flags back to the host (a few KB)
      std::vector<uint32_t> h_same(ncl), h_n0(ncl);
      hipLaunchKernelGGL(cluster_counts_kernel, dim3((uint32_t)((ncl + 255) / 256)), dim3(256), 0, st, b_scan.as<uint32_t>(),
                         d_sc.as<SplitCluster>(), (uint32_t)ncl, d_n0.as<uint32_t>());
      PANN_HIP(hipMemcpyAsync(h_same.data(), d_same.p, ncl * 4, hipMemcpyDeviceToHost, st));
      PANN_HIP(hipMemcpyAsync(h_n0.data(), d_n0.p, ncl * 4, hipMemcpyDeviceToHost, st));
      PANN_HIP(hipStreamSynchronize(st));
      std::vector<ScatterCluster> scat(ncl);
      std::vector<Cl> next;
      next.reserve(2 * ncl);
      for (size_t ci = 0; ci < ncl; ci++) {
        uint32_t n0 = h_n0[ci];
        const bool half = h_same[ci] || n0 == 0 || n0 == sc[ci].len;                  // :107-115 (+ empty-side guard)
        if (half) n0 = sc[ci].len / 2;
        scat[ci] = ScatterCluster{sc[ci].lo, sc[ci].len, n0, half ? 1u : 0u};
        const uint64_t r = src[ci].rnd;
        next.push_back(Cl{sc[ci].lo, n0, hc_mix_host(hc_mix_host(r) + 0 + 17)});     // rnd.fork(0) :85
        next.push_back(Cl{sc[ci].lo + n0, sc[ci].len - n0, hc_mix_host(hc_mix_host(r) + 1 + 17)});
      }
      PANN_HIP(hipMemcpyAsync(d_scat.p, scat.data(), ncl * sizeof(ScatterCluster), hipMemcpyHostToDevice, st));
      PANN_HIP(hipMemcpyAsync(fnew, fids, tot * 4, hipMemcpyDeviceToDevice, st));     // finished clusters keep their place
      hipLaunchKernelGGL(tree_scatter_kernel, dim3((uint32_t)nt), dim3(64), 0, st, fids, fnew, b_first.as<uint32_t>(),
                         b_scan.as<uint32_t>(), d_scat.as<ScatterCluster>(), d_tbase.as<uint32_t>(), (uint32_t)ncl);
      PANN_HIP(hipGetLastError());
      PANN_HIP(hipStreamSynchronize(st));   // the host vectors of this level go out of scope
      std::swap(fids, fnew);
      level.swap(next);
    }
   }
   t_tree += secs(tf0, now());
   for (uint32_t j = 0; j < ng; j++) {
    const auto t0 = now();
    const auto t1 = t0;
    uint32_t* ids = fids + (size_t)j * n;
    std::vector<uint64_t>& leaf_off = forest_leaves[j];
    std::sort(leaf_off.begin(), leaf_off.end());
    leaf_off.push_back(n);
    const uint32_t nleaves = (uint32_t)leaf_off.size() - 1;

    // ---- all-pairs 10-NN of every leaf (device ids, no host copy) ----
    HBuf d_tseg, d_ta0;
    PANN_HIP(hipMemcpyAsync(d_loff.p, leaf_off.data(), (nleaves + 1) * 8, hipMemcpyHostToDevice, st));
    if (leaf_knn_rows_eligible(ix, m)) {     // one-byte element types: lane-owns-row kernel (leaf_knn.hip)
      if (int rc = leaf_knn_rows_dev(ix, ws, st, ids, d_loff.as<uint64_t>(), leaf_off.data(), nleaves, m, 1, b_nnids.as<uint32_t>(), b_nnd.as<float>())) return rc;
    } else {
      std::vector<uint32_t> tseg, ta0;
      for (uint32_t l = 0; l < nleaves; l++)
        for (uint64_t a = leaf_off[l]; a < leaf_off[l + 1]; a += 64) { tseg.push_back(l); ta0.push_back((uint32_t)a); }
      if (d_tseg.alloc(tseg.size() * 4) || d_ta0.alloc(ta0.size() * 4)) { set_error("pann_hcnng_build: hipMalloc failed"); return PANN_ERR_HIP; }
      PANN_HIP(hipMemcpyAsync(d_tseg.p, tseg.data(), tseg.size() * 4, hipMemcpyHostToDevice, st));
      PANN_HIP(hipMemcpyAsync(d_ta0.p, ta0.data(), ta0.size() * 4, hipMemcpyHostToDevice, st));
      if (int rc = dense_topk_dev(ix, ws, st, nullptr, 0, ids, ids, d_loff.as<uint64_t>(), d_loff.as<uint64_t>(), d_tseg.as<uint32_t>(),
                                  d_ta0.as<uint32_t>(), (uint32_t)tseg.size(), n, n, 1, m, 1, b_nnids.as<uint32_t>(), b_nnd.as<float>())) return rc;
      PANN_HIP(hipStreamSynchronize(st));        // the host tile vectors go out of scope
    }
    PANN_HIP(hipStreamSynchronize(st));
    const auto t2 = now();

    // ---- edges, per-leaf sort (less_dup order), Kruskal, process_edges ----
    hipLaunchKernelGGL(invert_perm, dim3(nb256), dim3(256), 0, st, ids, b_pos.as<uint32_t>(), n);
    hipLaunchKernelGGL(leaf_lo_kernel, dim3(nleaves), dim3(64), 0, st, d_loff.as<uint64_t>(), nleaves, b_leaflo.as<uint32_t>());
    const uint64_t ne = n * m;
    hipLaunchKernelGGL(leaf_edges_kernel, dim3((uint32_t)((ne + 255) / 256)), dim3(256), 0, st, b_nnids.as<uint32_t>(), b_nnd.as<float>(),
                       b_pos.as<uint32_t>(), b_leaflo.as<uint32_t>(), m, n, b_ka.as<uint64_t>());
    PANN_HIP(hipGetLastError());
    std::vector<uint32_t> seg_b(nleaves), seg_e(nleaves);
    for (uint32_t l = 0; l < nleaves; l++) { seg_b[l] = (uint32_t)(leaf_off[l] * m); seg_e[l] = (uint32_t)(leaf_off[l + 1] * m); }
    PANN_HIP(hipMemcpyAsync(d_seg_b.p, seg_b.data(), nleaves * 4, hipMemcpyHostToDevice, st));
    PANN_HIP(hipMemcpyAsync(d_seg_e.p, seg_e.data(), nleaves * 4, hipMemcpyHostToDevice, st));
    size_t tb = sort_tmp;
    PANN_HIP(rocprim::segmented_radix_sort_keys(b_tmp.p, tb, b_ka.as<uint64_t>(), b_kb.as<uint64_t>(), (unsigned)ne, nleaves,
                                                (const uint32_t*)d_seg_b.as<uint32_t>(), (const uint32_t*)d_seg_e.as<uint32_t>(), 0, 64, st));
    MstArgs ma{};
    ma.keys = b_kb.as<uint64_t>(); ma.leaf_off = d_loff.as<uint64_t>(); ma.nleaves = nleaves; ma.m = m; ma.ids = ids;
    ma.graph = ix.graph; ma.gstride = ix.gstride; ma.max_deg = ix.max_deg; ma.deg = b_deg.as<uint32_t>(); ma.mst_deg = mst_deg;
    if (slab) {   // slotted rows: local tree tg + j owns slots [(tg+j)*mst_deg, (tg+j+1)*mst_deg)
      hipLaunchKernelGGL(fill_u32_hc, dim3(nb256), dim3(256), 0, st, b_deg.as<uint32_t>(), n, (tg + j) * mst_deg);
      ma.graph = slab; ma.gstride = slab_stride; ma.max_deg = (tg + j + 1) * mst_deg;
    }
    ma.g_parent = b_par.as<int32_t>(); ma.g_rank = b_rnk.as<uint8_t>(); ma.g_degree = b_dgr.as<uint8_t>();
    uint64_t max_leaf = 2;
    for (uint32_t l = 0; l < nleaves; l++) max_leaf = std::max<uint64_t>(max_leaf, leaf_off[l + 1] - leaf_off[l]);
    ma.lds_cap = (uint32_t)std::min<uint64_t>((max_leaf + 63) / 64 * 64, 4096);     // 12 bytes of LDS per member
    hipLaunchKernelGGL(leaf_mst_kernel, dim3(nleaves), dim3(PANN_WAVE), (size_t)ma.lds_cap * 12, st, ma);
    PANN_HIP(hipGetLastError());
    PANN_HIP(hipStreamSynchronize(st));
    const auto t3 = now();
    t_leaf += secs(t1, t2); t_mst += secs(t2, t3);
   }
  }
  if (times3) { times3[0] = t_tree; times3[1] = t_leaf; times3[2] = t_mst; }
  return PANN_OK;
}

}  // namespace pann
