// api.hip -- the extern "C" boundary declared in include/pann.h: handle management (device
// mirrors of PointRange / Graph), host<->device staging, and dispatch into the gfx950 kernels.
// No CPU compute path exists here: without a HIP device every entry point fails loudly.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "pann_internal.h"

namespace pann {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return PANN_ERR_HIP;
}

int Workspace::ensure(size_t need) {
  if (need <= bytes) return PANN_OK;
  if (buf) { PANN_HIP(hipFree(buf)); buf = nullptr; bytes = 0; }
  size_t cap = need + need / 4 + 4096;
  PANN_HIP(hipMalloc(&buf, cap));
  bytes = cap;
  return PANN_OK;
}
void Workspace::release() { if (buf) (void)hipFree(buf); buf = nullptr; bytes = 0; }

// a growable device buffer used for host-pointer entry points
struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return PANN_OK;
    if (p) { PANN_HIP(hipFree(p)); p = nullptr; bytes = 0; }
    size_t cap = need + need / 4 + 256;
    PANN_HIP(hipMalloc(&p, cap));
    bytes = cap;
    return PANN_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <typename T> T* as() { return (T*)p; }
};

// growable pinned host buffer: host-pointer calls pack their inputs / outputs here so that each direction is ONE
// DMA transfer (copies from or to pageable memory are staged and synchronised one by one by the runtime)
struct PinnedBuf {
  void* p = nullptr; size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return PANN_OK;
    if (p) { PANN_HIP(hipHostFree(p)); p = nullptr; bytes = 0; }
    size_t cap = need + need / 4 + 4096;
    PANN_HIP(hipHostMalloc(&p, cap, hipHostMallocDefault));
    bytes = cap;
    return PANN_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

}  // namespace pann

using namespace pann;

struct pann_index {
  DeviceIndex ix;
  int device = 0;
  hipStream_t stream = nullptr;      // the stream every call of this handle runs on: own_stream, or the caller's (pann_index_set_stream)
  hipStream_t own_stream = nullptr;
  Workspace ws, ws2, ws3, ws4;   // kernel scratch (search / prune / re-prune / rows of a batch)
  uint32_t vcap = 0;        // visited-list capacity used by the builder (grows on overflow)
  uint32_t dcap = 256;      // dropped-list capacity of the searches (pann_index_reserve_dropped; grows on overflow)
  uint32_t gt_pieces = 0;   // pann_index_set_option("gt_pieces"): pieces of the base per query tile in pann_bruteforce_knn (0 = auto)
  DevBuf cell_buf;          // locality cell of every point (ensure_locality_cells)
  uint32_t locality_groups = 32;    // ... whose pivots are grouped by their nearest of this many top pivots (0 / 1: no grouping)
  uint32_t locality_pivots = 1024;   // cells of the locality order (pann_index_set_option("locality_pivots"): measurement knob)
  int cells_state = 0;      // 0 not tried, 1 computed, -1 not worth it (table small enough to be cache resident) or switched off
  DevBuf code_rank, code_rows;   // filter-code table (filter_codes.hip): rank16[n], gcode[n][gstride]
  int codes_state = 0;      // 0 not tried, 1 available (rank16 built), -1 unavailable (a slot class has 4 095 or more members) or switched off
  DevBuf stage[12];      // staging for host-pointer calls
  PinnedBuf pin_in, pin_out;   // packed pinned staging of pann_batch_search
};

namespace {

static uint32_t esize_of(int dtype) {
  switch (dtype) { case PANN_U8: case PANN_I8: return 1; case PANN_F16: case PANN_BF16: return 2; case PANN_F32: return 4; }
  return 0;
}

struct DeviceGuard {
  int prev = -1; bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// reference layout n x (max_deg+1), slot 0 = degree  ->  device layout n x gstride, SENTINEL padded
// A neighbour id >= n would be gathered unchecked by every kernel (points + id * pstride): such a row is left empty
// and *bad is raised, the host then returns PANN_ERR_BAD_ARG (a graph file of another dataset, a truncated file).
__global__ void graph_to_device_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                       uint64_t nrows, uint32_t max_deg, uint32_t gstride,
                                       const uint32_t* __restrict__ row_ids, uint64_t n, uint32_t* __restrict__ bad) {
  const uint64_t r = blockIdx.x;
  if (r >= nrows) return;
  const uint32_t* s = src + r * (uint64_t)(max_deg + 1);
  const uint64_t target = row_ids ? row_ids[r] : r;
  uint32_t* d = dst + target * (uint64_t)gstride;
  const uint32_t deg = min(s[0], max_deg);
  bool ok = true;                                      // one wave per row
  for (uint32_t i0 = 0; i0 < deg; i0 += 64) {
    const uint32_t i = i0 + threadIdx.x;
    ok = ok && (__ballot(i < deg && (uint64_t)s[1 + i] >= n) == 0ull);
  }
  if (!ok && threadIdx.x == 0) atomicOr(bad, 1u);
  for (uint32_t i = threadIdx.x; i < gstride; i += blockDim.x) d[i] = (ok && i < deg) ? s[1 + i] : SENTINEL;
}

__global__ void graph_to_host_layout_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                            uint64_t nrows, uint32_t max_deg, uint32_t gstride) {
  const uint64_t r = blockIdx.x;
  if (r >= nrows) return;
  const uint32_t* s = src + r * (uint64_t)gstride;
  uint32_t* d = dst + r * (uint64_t)(max_deg + 1);
  // one wave per row: degree = number of non-sentinel slots (packed at the front)
  uint32_t deg = 0;
  for (uint32_t i0 = 0; i0 < gstride; i0 += 64) {
    const uint32_t i = i0 + threadIdx.x;
    const uint32_t a = i < gstride ? s[i] : SENTINEL;
    deg += __popcll(__ballot(a != SENTINEL));
    if (i < max_deg) d[1 + i] = (a != SENTINEL) ? a : 0u;   // Graph slabs are zero filled (graph.h:138)
  }
  if (threadIdx.x == 0) d[0] = deg;
}

__global__ void fill_u32_kernel(uint32_t* p, uint64_t n, uint32_t v) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) p[i] = v;
}

int check_idx(const pann_index* idx, const char* fn) {
  if (!idx) { set_error(std::string(fn) + ": null index handle"); return PANN_ERR_BAD_ARG; }
  return PANN_OK;
}

int upload_graph_rows(pann_index* idx, const uint32_t* h_rows, uint64_t m, const uint32_t* h_row_ids) {
  DeviceIndex& ix = idx->ix;
  if (m == 0) return PANN_OK;
  ix.codes_valid = 0;                                   // rows change without their filter codes (filter_codes.hip)
  const size_t row_bytes = (size_t)(ix.max_deg + 1) * 4;
  // stream in slices so the staging buffer stays bounded (288 GB HBM, but host slabs can be huge)
  const uint64_t slice = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
  if (int rc = idx->stage[9].ensure(4)) return rc;
  PANN_HIP(hipMemsetAsync(idx->stage[9].p, 0, 4, idx->stream));
  for (uint64_t r0 = 0; r0 < m; r0 += slice) {
    const uint64_t cnt = std::min(slice, m - r0);
    int rc = idx->stage[0].ensure(cnt * row_bytes); if (rc) return rc;
    PANN_HIP(hipMemcpyAsync(idx->stage[0].p, h_rows + r0 * (ix.max_deg + 1), cnt * row_bytes, hipMemcpyHostToDevice, idx->stream));
    const uint32_t* d_ids = nullptr;
    if (h_row_ids) {
      rc = idx->stage[1].ensure(cnt * 4); if (rc) return rc;
      PANN_HIP(hipMemcpyAsync(idx->stage[1].p, h_row_ids + r0, cnt * 4, hipMemcpyHostToDevice, idx->stream));
      d_ids = idx->stage[1].as<uint32_t>();
    }
    uint32_t* dst = h_row_ids ? ix.graph : ix.graph + r0 * (uint64_t)ix.gstride;
    hipLaunchKernelGGL(graph_to_device_kernel, dim3((uint32_t)cnt), dim3(64), 0, idx->stream,
                       idx->stage[0].as<uint32_t>(), dst, cnt, ix.max_deg, ix.gstride, d_ids, ix.n, idx->stage[9].as<uint32_t>());
    PANN_HIP(hipGetLastError());
    PANN_HIP(hipStreamSynchronize(idx->stream));
  }
  uint32_t bad = 0;
  PANN_HIP(hipMemcpy(&bad, idx->stage[9].p, 4, hipMemcpyDeviceToHost));
  if (bad) { set_error("graph upload: neighbour id out of range (>= number of points); those rows were left empty"); return PANN_ERR_BAD_ARG; }
  return PANN_OK;
}

// Locality cells: every point's nearest of 256 pivot points (evenly spaced ids), one dense top-1 pass per handle (~15 ms at
// 10M x 96 f32).  The Vamana builder launches the searches of a batch in cell order (vamana_build.hip): a 10M-point launch reads
// every row ~60 times, and queries that run side by side then read rows of the same few regions -- the 256 MiB Infinity Cache
// holds what a few cells need, not what 200 000 queries scattered over the whole table need (search phase -20 %; the graph does
// not depend on the launch order).  Only for tables beyond the cache.
__global__ void gather_rows_kernel(const uint8_t* points, uint32_t pstride, uint64_t n, uint32_t npiv, uint8_t* out) {
  const uint64_t src = (uint64_t)blockIdx.x * (n / npiv);
  for (uint32_t b = threadIdx.x * 16; b < pstride; b += blockDim.x * 16)
    *reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * pstride + b) = *reinterpret_cast<const uint4*>(points + src * pstride + b);
}
// cell <- (group of the cell's pivot << 16) | cell: cells whose pivots share a nearest "top" pivot sort next to each other
__global__ void group_cells_kernel(uint32_t* cell, uint64_t n, const uint32_t* pivot_group) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t c = cell[i]; cell[i] = (pivot_group[c] << 16) | c; }
}
int ensure_locality_cells(pann_index* idx) {
  DeviceIndex& ix = idx->ix;
  const uint32_t NPIV = idx->locality_pivots;
  if (ix.cell || idx->cells_state != 0 || ix.exact) return PANN_OK;
  static const bool off = ab_env("PANN_NO_LOCALITY") != nullptr;          // diagnostic A/B switch
  const bool forced = ix.cell_min_batch < 4096;                           // pann_index_set_option("locality_order", 2): tests
  if (off || ix.n < 2 * NPIV || (!forced && ((uint64_t)ix.n * ix.pstride < (1ull << 30) || ix.n < 256ull * NPIV))) {
    idx->cells_state = -1; return PANN_OK;
  }
  hipStream_t st = idx->stream;
  if (int rc = idx->stage[10].ensure((size_t)NPIV * ix.pstride)) return rc;
  if (int rc = idx->stage[11].ensure((size_t)ix.n * 4)) return rc;                      // the distances (not kept)
  if (int rc = idx->cell_buf.ensure((size_t)ix.n * 4 + 256)) return rc;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(NPIV), dim3(64), 0, st, ix.points, ix.pstride, ix.n, NPIV,
                     idx->stage[10].as<uint8_t>());
  PANN_HIP(hipGetLastError());
  DeviceIndex pix = ix;                       // the pivots as a 256-point table; A rows = all base points, as external rows
  pix.points = idx->stage[10].as<uint8_t>(); pix.n = NPIV; pix.graph = nullptr; pix.gcode = nullptr; pix.rank16 = nullptr; pix.cell = nullptr;
  if (int rc = dense_topk_dev(pix, idx->ws2, st, ix.points, ix.pstride, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              (uint32_t)((ix.n + 63) / 64), ix.n, NPIV, 1, 1, 0, idx->cell_buf.as<uint32_t>(),
                              idx->stage[11].as<float>())) return rc;
  if (idx->locality_groups > 1 && NPIV >= 4 * idx->locality_groups && NPIV <= 65536) {
    // the pivots themselves by their nearest of the first `groups` pivots (a prefix of the pivot slab is a table too)
    DeviceIndex gix = pix; gix.n = idx->locality_groups;
    uint32_t* d_pg = idx->stage[11].as<uint32_t>();                       // [NPIV] group of every pivot, then [NPIV] distances
    if (int rc = dense_topk_dev(gix, idx->ws2, st, pix.points, ix.pstride, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                (NPIV + 63) / 64, NPIV, gix.n, 1, 1, 0, d_pg, reinterpret_cast<float*>(d_pg + NPIV))) return rc;
    hipLaunchKernelGGL(group_cells_kernel, dim3((uint32_t)((ix.n + 255) / 256)), dim3(256), 0, st, idx->cell_buf.as<uint32_t>(), ix.n, d_pg);
    PANN_HIP(hipGetLastError());
  }
  PANN_HIP(hipStreamSynchronize(st));
  idx->stage[11].release();
  ix.cell = idx->cell_buf.as<uint32_t>();
  idx->cells_state = 1;
  return PANN_OK;
}

// The builder's L = 91..128 searches use the 12-bit filter-code table (filter_codes.hip) when every slot class is small
// enough.  Called by the Vamana entry points before their searches: builds rank16 once per handle, and gcode from the
// current graph whenever something outside the builder's row writers has changed the graph since.
int ensure_filter_codes(pann_index* idx, uint32_t L) {
  DeviceIndex& ix = idx->ix;
  if (int rc = ensure_locality_cells(idx)) return rc;     // (every Vamana entry point comes through here before its searches)
  if (L <= 90 || L > 128 || idx->codes_state < 0) return PANN_OK;
  if (ix.codes_valid) return PANN_OK;
  if (idx->codes_state == 0) {
    static const bool off = ab_env("PANN_NO_FILTER_CODES") != nullptr;      // diagnostic A/B switch
    if (off) { idx->codes_state = -1; return PANN_OK; }
    if (int rc = idx->code_rank.ensure((size_t)ix.n * 2 + 256)) return rc;
    uint32_t max_rank = 0;
    if (int rc = filter_codes_build_ranks(ix.n, FILTER_CODE_BITS, idx->ws2, idx->stream, idx->code_rank.as<uint16_t>(), &max_rank)) return rc;
    if (max_rank >= 0xFFFu) { idx->codes_state = -1; idx->code_rank.release(); return PANN_OK; }      // a code would not fit 12 bits
    if (int rc = idx->code_rows.ensure((size_t)ix.n * ix.gstride * 2 + 256)) return rc;
    idx->codes_state = 1;
    ix.rank16 = idx->code_rank.as<uint16_t>(); ix.gcode = idx->code_rows.as<uint16_t>();
  }
  if (int rc = filter_codes_rebuild_rows(ix, idx->stream)) return rc;
  ix.codes_valid = 1;
  return PANN_OK;
}

}  // namespace

extern "C" {

int pann_abi_version(void) { return PANN_ABI_VERSION; }
const char* pann_last_error(void) { return g_err.c_str(); }

int pann_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int index_create_impl(pann_index** out, const void* points, uint64_t n, uint32_t d, int dtype,
                             uint64_t row_stride_bytes, int metric, const uint32_t* graph,
                             uint32_t max_deg, int device);

int pann_index_create(pann_index** out, const void* points, uint64_t n, uint32_t d, int dtype,
                      uint64_t row_stride_bytes, int metric, const uint32_t* graph,
                      uint32_t max_deg, int device) {
  if (!points) { set_error("pann_index_create: null/empty argument"); return PANN_ERR_BAD_ARG; }
  return index_create_impl(out, points, n, d, dtype, row_stride_bytes, metric, graph, max_deg, device);
}

int pann_index_create_empty(pann_index** out, uint64_t n, uint32_t d, int dtype, int metric, uint32_t max_deg, int device) {
  return index_create_impl(out, nullptr, n, d, dtype, (uint64_t)d * esize_of(dtype), metric, nullptr, max_deg, device);
}

int pann_index_upload_points(pann_index* idx, uint64_t first_row, const void* rows, uint64_t nrows, uint64_t row_stride_bytes) {
  if (int rc = check_idx(idx, "pann_index_upload_points")) return rc;
  if (nrows == 0) return PANN_OK;
  const DeviceIndex& ix = idx->ix;
  if (!rows) { set_error("pann_index_upload_points: null argument"); return PANN_ERR_BAD_ARG; }
  if (first_row > ix.n || nrows > ix.n - first_row) { set_error("pann_index_upload_points: row range outside the index"); return PANN_ERR_BAD_ARG; }
  if (row_stride_bytes < ix.dbytes) { set_error("pann_index_upload_points: row stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  PANN_HIP(hipStreamSynchronize(idx->stream));                  // no search may be reading the rows being replaced
  const uint64_t slice = std::max<uint64_t>(1, (1ull << 30) / std::max<uint64_t>(row_stride_bytes, 1));
  for (uint64_t r0 = 0; r0 < nrows; r0 += slice) {
    const uint64_t cnt = std::min(slice, nrows - r0);
    PANN_HIP(hipMemcpy2D(ix.points + (first_row + r0) * ix.pstride, ix.pstride, (const uint8_t*)rows + r0 * row_stride_bytes,
                         row_stride_bytes, ix.dbytes, cnt, hipMemcpyHostToDevice));
  }
  return PANN_OK;
}

static int index_create_impl(pann_index** out, const void* points, uint64_t n, uint32_t d, int dtype,
                             uint64_t row_stride_bytes, int metric, const uint32_t* graph,
                             uint32_t max_deg, int device) {
  if (!out || n == 0 || d == 0) { set_error("pann_index_create: null/empty argument"); return PANN_ERR_BAD_ARG; }
  const uint32_t es = esize_of(dtype);
  if (es == 0) { set_error("pann_index_create: unknown dtype"); return PANN_ERR_BAD_ARG; }
  if (metric != PANN_L2 && metric != PANN_MIPS) { set_error("pann_index_create: unknown metric"); return PANN_ERR_BAD_ARG; }
  if (row_stride_bytes < (uint64_t)d * es) { set_error("pann_index_create: row stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  if (n >= 0x7FFFFFFFull) { set_error("pann_index_create: n must be < 2^31 (robustPrune uses int ids, vamana/index.h:97)"); return PANN_ERR_BAD_ARG; }
  if (max_deg == 0 || max_deg > 4096) { set_error("pann_index_create: max_deg out of range [1,4096]"); return PANN_ERR_BAD_ARG; }
  int ndev = pann_device_count();
  if (ndev <= 0) { set_error("pann_index_create: no HIP device visible (this library has no CPU path)"); return PANN_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { set_error("pann_index_create: device ordinal out of range"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(device);
  if (!g.ok) { set_error("pann_index_create: hipSetDevice failed"); return PANN_ERR_HIP; }

  pann_index* idx = new pann_index();
  idx->device = device;
  DeviceIndex& ix = idx->ix;
  ix.n = n; ix.d = d; ix.dtype = dtype; ix.metric = metric; ix.esize = es; ix.dbytes = d * es;
  choose_point_layout(ix.dbytes, &ix.lpc, &ix.nch);
  ix.pstride = ix.lpc * ix.nch * 16;
  ix.max_deg = max_deg; ix.gstride = (max_deg + 15) / 16 * 16;
  auto fail = [&](int rc) { pann_index_destroy(idx); return rc; };
  if (hipStreamCreateWithFlags(&idx->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(PANN_ERR_HIP); }
  idx->stream = idx->own_stream;
  hipError_t e;
  if ((e = hipMalloc((void**)&ix.points, n * (size_t)ix.pstride)) != hipSuccess) return fail(hip_fail(e, "hipMalloc(points)"));
  if ((e = hipMalloc((void**)&ix.graph, n * (size_t)ix.gstride * 4)) != hipSuccess) return fail(hip_fail(e, "hipMalloc(graph)"));
  // points: strided copy into the zero-padded device rows
  if ((e = hipMemsetAsync(ix.points, 0, n * (size_t)ix.pstride, idx->stream)) != hipSuccess) return fail(hip_fail(e, "hipMemset(points)"));
  if ((e = hipStreamSynchronize(idx->stream)) != hipSuccess) return fail(hip_fail(e, "sync"));
  if (points) {
    const uint64_t slice = std::max<uint64_t>(1, (1ull << 30) / std::max<uint64_t>(row_stride_bytes, 1));
    for (uint64_t r0 = 0; r0 < n; r0 += slice) {
      const uint64_t cnt = std::min(slice, n - r0);
      e = hipMemcpy2D(ix.points + r0 * ix.pstride, ix.pstride, (const uint8_t*)points + r0 * row_stride_bytes,
                      row_stride_bytes, ix.dbytes, cnt, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpy2D(points)"));
    }
  }
  if (graph) {
    int rc = upload_graph_rows(idx, graph, n, nullptr);
    if (rc) return fail(rc);
  } else {
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, idx->stream, ix.graph, n * (uint64_t)ix.gstride, SENTINEL);
    if ((e = hipStreamSynchronize(idx->stream)) != hipSuccess) return fail(hip_fail(e, "fill graph"));
  }
  *out = idx;
  return PANN_OK;
}

void pann_index_destroy(pann_index* idx) {
  if (!idx) return;
  DeviceGuard g(idx->device);
  if (idx->stream) (void)hipStreamSynchronize(idx->stream);
  if (idx->ix.points) (void)hipFree(idx->ix.points);
  if (idx->ix.graph) (void)hipFree(idx->ix.graph);
  idx->ws.release(); idx->ws2.release(); idx->ws3.release(); idx->ws4.release();
  for (auto& s : idx->stage) s.release();
  idx->pin_in.release(); idx->pin_out.release();
  idx->code_rank.release(); idx->code_rows.release(); idx->cell_buf.release();
  if (idx->own_stream) (void)hipStreamDestroy(idx->own_stream);
  delete idx;
}

uint64_t pann_index_size(const pann_index* idx) { return idx ? idx->ix.n : 0; }
uint32_t pann_index_dims(const pann_index* idx) { return idx ? idx->ix.d : 0; }
uint32_t pann_index_max_degree(const pann_index* idx) { return idx ? idx->ix.max_deg : 0; }
int pann_index_device(const pann_index* idx) { return idx ? idx->device : -1; }

int pann_index_set_exact_float_order(pann_index* idx, int on) {
  if (int rc = check_idx(idx, "pann_index_set_exact_float_order")) return rc;
  DeviceIndex& ix = idx->ix;
  const bool is_float = ix.dtype == PANN_F32 || ix.dtype == PANN_F16 || ix.dtype == PANN_BF16;
  if (on && is_float) { ix.exact = 1; ix.lpc = 4; ix.nch = ix.pstride / 64; }   // whole query in LDS, lane-per-candidate sums
  else { ix.exact = 0; choose_point_layout(ix.dbytes, &ix.lpc, &ix.nch); }
  return PANN_OK;
}

int pann_index_reserve_dropped(pann_index* idx, uint32_t cap) {
  if (int rc = check_idx(idx, "pann_index_reserve_dropped")) return rc;
  if (cap > (1u << 30)) { set_error("pann_index_reserve_dropped: capacity out of range"); return PANN_ERR_BAD_ARG; }
  if (cap > idx->dcap) idx->dcap = (cap + 63) / 64 * 64;
  return PANN_OK;
}
uint32_t pann_index_dropped_capacity(const pann_index* idx) { return idx ? idx->dcap : 0; }

int pann_index_set_graph(pann_index* idx, const uint32_t* graph) {
  if (int rc = check_idx(idx, "pann_index_set_graph")) return rc;
  if (!graph) { set_error("pann_index_set_graph: null graph"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  return upload_graph_rows(idx, graph, idx->ix.n, nullptr);
}

int pann_index_update_rows(pann_index* idx, const uint32_t* row_ids, const uint32_t* rows, uint64_t m) {
  if (int rc = check_idx(idx, "pann_index_update_rows")) return rc;
  if (m && (!row_ids || !rows)) { set_error("pann_index_update_rows: null argument"); return PANN_ERR_BAD_ARG; }
  for (uint64_t i = 0; i < m; i++)
    if (row_ids[i] >= idx->ix.n) { set_error("ERROR: graph index out of range"); return PANN_ERR_BAD_ARG; }  // graph.h:235-238
  DeviceGuard g(idx->device);
  return upload_graph_rows(idx, rows, m, row_ids);
}

int pann_index_clear_graph(pann_index* idx) {
  if (int rc = check_idx(idx, "pann_index_clear_graph")) return rc;
  DeviceGuard g(idx->device);
  PANN_HIP(hipMemsetAsync(idx->ix.graph, 0xFF, (size_t)idx->ix.n * idx->ix.gstride * 4, idx->stream));
  if (idx->ix.gcode) PANN_HIP(hipMemsetAsync(idx->ix.gcode, 0xFF, (size_t)idx->ix.n * idx->ix.gstride * 2, idx->stream));   // codes of an empty graph
  PANN_HIP(hipStreamSynchronize(idx->stream));
  return PANN_OK;
}

int64_t pann_index_get_option(const pann_index* idx, const char* name) {
  if (!idx || !name) return -1;
  const std::string nm = name;
  if (nm == "forest_group") return idx->ix.forest_group;
  if (nm == "gt_pieces") return idx->gt_pieces;
  if (nm == "locality_order") return idx->ix.cell ? 1 : 0;
  if (nm == "filter_codes") return idx->ix.codes_valid ? 1 : 0;         // are the class codes in step with the graph right now?
  return -1;
}

int pann_index_set_stream(pann_index* idx, void* stream, int use_private) {
  if (int rc = check_idx(idx, "pann_index_set_stream")) return rc;
  DeviceGuard g(idx->device);
  PANN_HIP(hipStreamSynchronize(idx->stream));                    // nothing of this handle is left on the stream it leaves
  idx->stream = use_private ? idx->own_stream : (hipStream_t)stream;
  return PANN_OK;
}

int pann_index_set_option(pann_index* idx, const char* name, int64_t value) {
  if (int rc = check_idx(idx, "pann_index_set_option")) return rc;
  const std::string nm = name ? name : "";
  if (value < 0 || value > 0x7FFFFFFF) { set_error("pann_index_set_option: value out of range"); return PANN_ERR_BAD_ARG; }
  if (nm == "forest_group") idx->ix.forest_group = (uint32_t)value;
  else if (nm == "gt_pieces") idx->gt_pieces = (uint32_t)value;
  else if (nm == "locality_pivots") { idx->locality_pivots = std::max<uint32_t>(2, std::min<uint32_t>((uint32_t)value, 65536)); idx->ix.cell = nullptr; idx->cells_state = idx->cells_state < 0 ? idx->cells_state : 0; }
  else if (nm == "locality_groups") { idx->locality_groups = (uint32_t)std::min<int64_t>(value, 4096); idx->ix.cell = nullptr; idx->cells_state = idx->cells_state < 0 ? idx->cells_state : 0; }
  else if (nm == "locality_order") {        // 0: the builder launches a batch's searches in batch order
    idx->ix.cell = nullptr;
    idx->ix.cell_min_batch = value == 2 ? 64u : 4096u;       // 2: also on small tables and small batches (tests)
    idx->cells_state = value ? ((idx->cell_buf.p && idx->cells_state == 1) ? 1 : 0) : -2;
    if (idx->cells_state == 1) idx->ix.cell = idx->cell_buf.as<uint32_t>();
  }
  else if (nm == "filter_codes") {          // 0: the beam-91..128 searches use the id table even where the class codes are available
    idx->ix.codes_valid = 0;
    idx->codes_state = value ? (idx->ix.rank16 ? 1 : 0) : -2;
  }
  else { set_error("pann_index_set_option: unknown option '" + nm + "'"); return PANN_ERR_BAD_ARG; }
  return PANN_OK;
}

int pann_index_get_graph(pann_index* idx, uint32_t* graph_out) {
  if (int rc = check_idx(idx, "pann_index_get_graph")) return rc;
  if (!graph_out) { set_error("pann_index_get_graph: null output"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  DeviceIndex& ix = idx->ix;
  const size_t row_bytes = (size_t)(ix.max_deg + 1) * 4;
  const uint64_t slice = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
  for (uint64_t r0 = 0; r0 < ix.n; r0 += slice) {
    const uint64_t cnt = std::min(slice, ix.n - r0);
    if (int rc = idx->stage[0].ensure(cnt * row_bytes)) return rc;
    hipLaunchKernelGGL(graph_to_host_layout_kernel, dim3((uint32_t)cnt), dim3(64), 0, idx->stream,
                       ix.graph + r0 * (uint64_t)ix.gstride, idx->stage[0].as<uint32_t>(), cnt, ix.max_deg, ix.gstride);
    PANN_HIP(hipGetLastError());
    PANN_HIP(hipMemcpyAsync(graph_out + r0 * (ix.max_deg + 1), idx->stage[0].p, cnt * row_bytes, hipMemcpyDeviceToHost, idx->stream));
    PANN_HIP(hipStreamSynchronize(idx->stream));
  }
  return PANN_OK;
}

// ---------------------------------------------------------------------------------------------
// batched beam search
// ---------------------------------------------------------------------------------------------

static int search_common_checks(pann_index* idx, uint64_t nq, const pann_query_params* qp, const pann_search_out* out) {
  if (int rc = check_idx(idx, "pann_batch_search")) return rc;
  if (!qp || !out) { set_error("pann_batch_search: null params/out"); return PANN_ERR_BAD_ARG; }
  if (qp->k > qp->beam) {  // beamSearch.h:368-372, :549-553
    set_error("Error: beam search parameter Q = " + std::to_string(qp->beam) + " same size or smaller than k = " + std::to_string(qp->k));
    return PANN_ERR_BAD_ARG;
  }
  (void)nq;
  return PANN_OK;
}

int pann_batch_search_dev(pann_index* idx, const void* d_queries, const uint32_t* d_query_ids,
                          uint64_t nq, uint64_t q_stride_bytes, const uint32_t* d_starts,
                          uint32_t nstarts, const pann_query_params* qp,
                          const pann_search_out* d_out, void* stream) {
  if (int rc = search_common_checks(idx, nq, qp, d_out)) return rc;
  if (!d_starts) { set_error("beam search expects at least one start point"); return PANN_ERR_BAD_ARG; }
  if (d_queries && q_stride_bytes < idx->ix.dbytes) { set_error("pann_batch_search: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  SearchArgs a;
  a.queries = (const uint8_t*)d_queries; a.qstride = q_stride_bytes; a.query_ids = d_query_ids;
  a.nq = nq; a.starts = d_starts; a.nstarts = nstarts;
  a.k = qp->k; a.beam = qp->beam; a.limit = qp->limit; a.degree_limit = qp->degree_limit; a.cut = qp->cut;
  a.dcap = idx->dcap;
  a.out = *d_out;
  if (int rc = idx->ws.ensure(search_workspace_bytes(idx->ix, a))) return rc;
  return launch_beam_search(idx->ix, a, idx->ws.buf, idx->ws.bytes, (hipStream_t)stream);
}

}  // extern "C"

static int batch_search_host(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                             uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts, int per_query,
                             const pann_query_params* qp, const pann_search_out* out) {
  if (int rc = search_common_checks(idx, nq, qp, out)) return rc;
  if (!starts || nstarts == 0) { set_error("beam search expects at least one start point"); return PANN_ERR_BAD_ARG; }
  if ((queries == nullptr) == (query_ids == nullptr)) { set_error("pann_batch_search: exactly one of queries / query_ids must be given"); return PANN_ERR_BAD_ARG; }
  const uint64_t nst_total = per_query ? nq * nstarts : nstarts;
  for (uint64_t i = 0; i < nst_total; i++)
    if (starts[i] >= idx->ix.n) { set_error("pann_batch_search: start point out of range"); return PANN_ERR_BAD_ARG; }
  if (query_ids)
    for (uint64_t i = 0; i < nq; i++)
      if (query_ids[i] >= idx->ix.n) { set_error("pann_batch_search: query id out of range"); return PANN_ERR_BAD_ARG; }
  if (nq == 0) return PANN_OK;
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  const DeviceIndex& ix = idx->ix;
  // ---- inputs: packed into pinned memory, one H2D transfer ----
  if (queries && q_stride_bytes < ix.dbytes) { set_error("pann_batch_search: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t qbytes = queries ? (nq - 1) * q_stride_bytes + ix.dbytes : nq * 4;
  const size_t off_st = al(qbytes + 16), in_bytes = off_st + al((size_t)nst_total * 4);
  if (int rc = idx->pin_in.ensure(in_bytes)) return rc;
  if (int rc = idx->stage[2].ensure(in_bytes)) return rc;
  std::memcpy(idx->pin_in.p, queries ? queries : (const void*)query_ids, qbytes);
  std::memcpy((uint8_t*)idx->pin_in.p + off_st, starts, (size_t)nst_total * 4);
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, idx->pin_in.p, in_bytes, hipMemcpyHostToDevice, st));
  const void* d_q = queries ? idx->stage[2].p : nullptr;
  const uint32_t* d_qid = queries ? nullptr : idx->stage[2].as<uint32_t>();
  const uint32_t* d_starts = (const uint32_t*)((uint8_t*)idx->stage[2].p + off_st);

  // ---- outputs: one packed device region, one D2H transfer into pinned memory, then host copies ----
  const size_t ok = out->out_k, vc = out->visited_cap;
  struct Piece { void* host; size_t bytes; size_t off; };
  Piece pc[8] = {{out->ids, nq * ok * 4, 0}, {out->dists, nq * ok * 4, 0}, {out->frontier_size, nq * 4, 0},
                 {out->visited_count, nq * 4, 0}, {out->dist_cmps, nq * 4, 0}, {out->degree_sum, nq * 4, 0},
                 {out->visited_ids, nq * vc * 4, 0}, {out->visited_dists, nq * vc * 4, 0}};
  size_t out_bytes = 0;
  for (auto& x : pc) { if (!x.host) x.bytes = 0; x.off = out_bytes; out_bytes += al(x.bytes); }
  if (int rc = idx->stage[4].ensure(out_bytes + 256)) return rc;
  if (int rc = idx->pin_out.ensure(out_bytes + 256)) return rc;
  auto dptr = [&](int i) -> void* { return pc[i].bytes ? (void*)((uint8_t*)idx->stage[4].p + pc[i].off) : nullptr; };
  pann_search_out d = *out;
  d.ids = (uint32_t*)dptr(0); d.dists = (float*)dptr(1); d.frontier_size = (uint32_t*)dptr(2);
  d.visited_count = (uint32_t*)dptr(3); d.dist_cmps = (uint32_t*)dptr(4); d.degree_sum = (uint32_t*)dptr(5);
  d.visited_ids = (uint32_t*)dptr(6); d.visited_dists = (float*)dptr(7);
  if (!d.visited_ids && !d.visited_dists) d.visited_cap = 0;

  d.status = nullptr;   // read from the workspace below
  uint32_t status = 0;
  // The "dropped" scratch is nq * dcap * 8 bytes.  When a launch reports that it was too small the list is grown (x8, up to
  // min(limit, n): a query drops at most one entry per visited vertex) and the batch runs again; a grown list that would take
  // more than kDropBudget for the whole batch makes the batch run in ranges of queries instead, and the handle keeps at most
  // kDropKeep entries per query for later calls (10K queries x 2048 x 8 B = 160 MB), not the worst case of one odd batch.
  constexpr uint64_t kDropBudget = 1ull << 30;
  constexpr uint32_t kDropKeep = 2048;
  const uint64_t dneed = (uint64_t)std::min<int64_t>(std::max<int64_t>(qp->limit, 1), (int64_t)ix.n);
  uint32_t dcap = idx->dcap;
  bool results_home = false;        // the packed outputs already sit in pin_out
  for (;;) {
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(nq, kDropBudget / ((uint64_t)std::max<uint32_t>(dcap, 64) * 8)));
    status = 0;
    for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
      const uint64_t cnt = std::min(chunk, nq - q0);
      SearchArgs a;
      a.queries = d_q ? (const uint8_t*)d_q + q0 * q_stride_bytes : nullptr; a.qstride = q_stride_bytes;
      a.query_ids = d_qid ? d_qid + q0 : nullptr;
      a.nq = cnt; a.starts = per_query ? d_starts + q0 * nstarts : d_starts; a.nstarts = nstarts; a.starts_per_query = per_query;
      a.k = qp->k; a.beam = qp->beam; a.limit = qp->limit; a.degree_limit = qp->degree_limit; a.cut = qp->cut;
      a.dcap = dcap;
      a.out = d;
      if (a.out.ids) a.out.ids += q0 * ok;
      if (a.out.dists) a.out.dists += q0 * ok;
      if (a.out.frontier_size) a.out.frontier_size += q0;
      if (a.out.visited_count) a.out.visited_count += q0;
      if (a.out.dist_cmps) a.out.dist_cmps += q0;
      if (a.out.degree_sum) a.out.degree_sum += q0;
      if (a.out.visited_ids) a.out.visited_ids += q0 * vc;
      if (a.out.visited_dists) a.out.visited_dists += q0 * vc;
      if (int rc = idx->ws.ensure(search_workspace_bytes(idx->ix, a))) return rc;
      if (int rc = launch_beam_search(idx->ix, a, idx->ws.buf, idx->ws.bytes, st)) return rc;
      uint32_t st_word = 0;      // the launch's status word (the next launch clears it)
      if (cnt == nq) {           // the whole batch in one launch (the normal case): the word travels with the results, ONE transfer
        PANN_HIP(hipMemcpyAsync((uint8_t*)idx->stage[4].p + out_bytes, (uint8_t*)idx->ws.buf + 64, 4, hipMemcpyDeviceToDevice, st));
        PANN_HIP(hipMemcpyAsync(idx->pin_out.p, idx->stage[4].p, out_bytes + 4, hipMemcpyDeviceToHost, st));
        PANN_HIP(hipStreamSynchronize(st));
        std::memcpy(&st_word, (uint8_t*)idx->pin_out.p + out_bytes, 4);
        results_home = true;
      } else {
        PANN_HIP(hipMemcpyAsync(&st_word, (uint8_t*)idx->ws.buf + 64, 4, hipMemcpyDeviceToHost, st));
        PANN_HIP(hipStreamSynchronize(st));
        results_home = false;
      }
      status |= st_word;
      if (status & PANN_STATUS_DROPPED_OVERFLOW) break;
    }
    if (!(status & PANN_STATUS_DROPPED_OVERFLOW)) break;
    // The reference has no such list (its `visited` vector grows as needed, beamSearch.h:80,113): grow ours and run the batch again.
    if ((uint64_t)dcap >= dneed) { set_error("pann_batch_search: internal dropped-list overflow"); return PANN_ERR_OVERFLOW; }
    dcap = (uint32_t)std::min<uint64_t>((uint64_t)dcap * 8, (dneed + 63) / 64 * 64);
  }
  idx->dcap = std::max(idx->dcap, std::min(dcap, kDropKeep));
  if (!results_home) {
    PANN_HIP(hipMemcpyAsync(idx->pin_out.p, idx->stage[4].p, out_bytes, hipMemcpyDeviceToHost, st));
    PANN_HIP(hipStreamSynchronize(st));
  }
  if (idx->ws.bytes > (2ull << 30)) idx->ws.release();          // a one-off worst-case scratch is not kept on the handle
  for (auto& x : pc) if (x.bytes) std::memcpy(x.host, (uint8_t*)idx->pin_out.p + x.off, x.bytes);
  if (out->status) *out->status = status;
  if (status & PANN_STATUS_VISITED_OVERFLOW) { set_error("pann_batch_search: visited list longer than visited_cap"); return PANN_ERR_OVERFLOW; }
  return PANN_OK;
}

extern "C" {

int pann_batch_search(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                      uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts,
                      const pann_query_params* qp, const pann_search_out* out) {
  return batch_search_host(idx, queries, query_ids, nq, q_stride_bytes, starts, nstarts, 0, qp, out);
}

int pann_batch_search_per_query_starts(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                                       uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts,
                                       const pann_query_params* qp, const pann_search_out* out) {
  return batch_search_host(idx, queries, query_ids, nq, q_stride_bytes, starts, nstarts, 1, qp, out);
}


// ---------------------------------------------------------------------------------------------
// robustPrune / Vamana build
// ---------------------------------------------------------------------------------------------

int pann_robust_prune_batch(pann_index* idx, const uint32_t* owners, uint64_t m, const uint32_t* cand_ids,
                            const float* cand_dists, const uint64_t* cand_offsets, double alpha, uint32_t R,
                            int add_out_nbrs, uint32_t* out_rows, uint32_t* out_dist_cmps) {
  if (int rc = check_idx(idx, "pann_robust_prune_batch")) return rc;
  if (m && (!owners || !cand_offsets || !out_rows)) { set_error("pann_robust_prune_batch: null argument"); return PANN_ERR_BAD_ARG; }
  if (m && cand_offsets[m] && !cand_ids) { set_error("pann_robust_prune_batch: null candidate ids"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  return robust_prune_batch_host(idx->ix, idx->ws2, idx->stream, owners, m, cand_ids, cand_dists, cand_offsets,
                                 alpha, R, add_out_nbrs, out_rows, out_dist_cmps);
}

static uint32_t default_vcap(uint32_t L) { return std::max<uint32_t>(2 * L, 128); }

int pann_vamana_insert_batch(pann_index* idx, const uint32_t* batch_ids, uint64_t m, uint32_t start, uint32_t R,
                             uint32_t L, double alpha, pann_build_stats* stats) {
  if (int rc = check_idx(idx, "pann_vamana_insert_batch")) return rc;
  if (m == 0) return PANN_OK;
  if (!batch_ids) { set_error("pann_vamana_insert_batch: null batch"); return PANN_ERR_BAD_ARG; }
  if (L == 0 || L > 65536) { set_error("pann_vamana_insert_batch: L out of range"); return PANN_ERR_BAD_ARG; }
  if (start >= idx->ix.n) { set_error("pann_vamana_insert_batch: start out of range"); return PANN_ERR_BAD_ARG; }
  for (uint64_t i = 0; i < m; i++)
    if (batch_ids[i] >= idx->ix.n) {  // vamana/index.h:193-198
      set_error("ERROR: invalid point " + std::to_string(batch_ids[i]) + " given to batch_insert"); return PANN_ERR_BAD_ARG;
    }
  DeviceGuard g(idx->device);
  if (int rc = ensure_filter_codes(idx, L)) return rc;
  if (int rc = idx->stage[2].ensure(m * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, batch_ids, m * 4, hipMemcpyHostToDevice, idx->stream));
  if (idx->vcap < default_vcap(L)) idx->vcap = default_vcap(L);
  return insert_batch_dev(idx->ix, idx->ws2, idx->ws3, idx->ws, idx->ws4, idx->stream, idx->stage[2].as<uint32_t>(), (uint32_t)m,
                          start, R, L, alpha, &idx->vcap, stats);
}

// ---- the two phases of a batch on device pointers: the seam of the multi-GPU build (parlayann_amd/distributed.py) ----

int pann_vamana_search_prune_dev(pann_index* idx, const uint32_t* d_batch_ids, uint64_t m, uint32_t start, uint32_t R, uint32_t L,
                                 double alpha, uint32_t* d_rows_out, pann_build_stats* stats) {
  if (int rc = check_idx(idx, "pann_vamana_search_prune_dev")) return rc;
  if (m == 0) return PANN_OK;
  if (!d_batch_ids || !d_rows_out) { set_error("pann_vamana_search_prune_dev: null argument"); return PANN_ERR_BAD_ARG; }
  if (L == 0 || L > 65536) { set_error("pann_vamana_search_prune_dev: L out of range"); return PANN_ERR_BAD_ARG; }
  if (start >= idx->ix.n || m > 0xFFFFFFF0ull) { set_error("pann_vamana_search_prune_dev: start / batch size out of range"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  if (int rc = ensure_filter_codes(idx, L)) return rc;
  if (idx->vcap < default_vcap(L)) idx->vcap = default_vcap(L);
  return vamana_search_prune_dev(idx->ix, idx->ws2, idx->ws, idx->stream, d_batch_ids, (uint32_t)m, start, R, L, alpha, &idx->vcap,
                                 d_rows_out, stats);
}

int pann_vamana_apply_rows_dev(pann_index* idx, const uint32_t* d_batch_ids, uint64_t m, const uint32_t* d_rows, uint32_t R,
                               double alpha, pann_build_stats* stats) {
  if (int rc = check_idx(idx, "pann_vamana_apply_rows_dev")) return rc;
  if (m == 0) return PANN_OK;
  if (!d_batch_ids || !d_rows) { set_error("pann_vamana_apply_rows_dev: null argument"); return PANN_ERR_BAD_ARG; }
  if (m > 0xFFFFFFF0ull) { set_error("pann_vamana_apply_rows_dev: batch too large"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  return vamana_apply_rows_dev(idx->ix, idx->ws2, idx->ws3, idx->stream, d_batch_ids, (uint32_t)m, d_rows, R, alpha, stats);
}

int pann_vamana_sort_neighbors(pann_index* idx) {
  if (int rc = check_idx(idx, "pann_vamana_sort_neighbors")) return rc;
  DeviceGuard g(idx->device);
  idx->ix.codes_valid = 0;                              // the rows are permuted without their filter codes
  return sort_neighbors_dev(idx->ix, idx->stream);
}

// the insertion order of this build: Fisher-Yates driven by splitmix64(seed) (DESIGN.md "Build
// determinism"; parlay::random_permutation, vamana/index.h:212, is not reproducible offline)
static void build_permutation(uint64_t m, uint64_t seed, uint32_t* out) {
  for (uint64_t i = 0; i < m; i++) out[i] = (uint32_t)i;
  uint64_t s = seed;
  auto next = [&]() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  // The swap partners do not depend on the array, so they are drawn a block ahead and their cache lines requested before the
  // swaps are made in order: the same permutation, without a cache miss per step (10M ids: 0.32 -> 0.1 s of every build).
  constexpr uint64_t BLK = 64;
  uint64_t part[BLK];
  for (uint64_t hi = m; hi > 1;) {
    const uint64_t cnt = std::min<uint64_t>(BLK, hi - 1);
    for (uint64_t k = 0; k < cnt; k++) { part[k] = next() % (hi - k); __builtin_prefetch(&out[part[k]], 1); }
    for (uint64_t k = 0; k < cnt; k++) std::swap(out[hi - k - 1], out[part[k]]);
    hi -= cnt;
  }
}

void pann_build_permutation(uint64_t n, uint64_t seed, uint32_t* out) {
  if (!out) return;
  build_permutation(n, seed, out);
}

uint64_t pann_vamana_batch_schedule(uint64_t n, uint64_t m, uint64_t* bounds, uint64_t cap) {
  // vamana/index.h:206-209, :223-234 with base 2 and max_fraction .02 (the values build_index passes, :174-177)
  size_t max_batch = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);
  if (max_batch == 0) max_batch = n;
  uint64_t nb = 0;
  size_t count = 0, inc = 0;
  while (count < m) {
    size_t floor, ceiling;
    if (std::pow(2.0, (double)inc) <= (double)max_batch) {
      floor = (size_t)std::pow(2.0, (double)inc) - 1;
      ceiling = std::min((size_t)std::pow(2.0, (double)(inc + 1)) - 1, (size_t)m);
      count = ceiling;
    } else {
      floor = count;
      ceiling = std::min(count + max_batch, (size_t)m);
      count += max_batch;
    }
    if (bounds && nb < cap) { bounds[2 * nb] = floor; bounds[2 * nb + 1] = ceiling; }
    nb++;
    inc++;
  }
  return nb;
}

namespace {
// vamana/index.h:156-170 with this build's own generator: edge j of vertex i = splitmix64(seed + golden * (i*degree + j + 1)) mod n
__global__ void random_edges_kernel(uint32_t* graph, uint32_t gstride, uint64_t n, uint32_t degree, uint64_t seed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * gstride) return;
  const uint64_t i = t / gstride, j = t % gstride;
  uint32_t v = SENTINEL;
  if (j < degree) {
    uint64_t z = seed + 0x9e3779b97f4a7c15ull * (i * degree + j + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    v = (uint32_t)(z % n);
  }
  graph[t] = v;
}
}  // namespace

int pann_vamana_build_single_batch(pann_index* idx, uint32_t R, uint32_t L, double alpha, int num_passes, uint32_t degree,
                                   uint64_t seed, int sort_neighbors, pann_build_stats* stats) {
  if (int rc = check_idx(idx, "pann_vamana_build_single_batch")) return rc;
  if (L == 0 || L > 65536 || num_passes < 1) { set_error("pann_vamana_build_single_batch: bad L / num_passes"); return PANN_ERR_BAD_ARG; }
  if (degree == 0 || degree > idx->ix.max_deg) { set_error("pann_vamana_build_single_batch: degree must be in [1, max_deg]"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  const uint64_t n = idx->ix.n;
  if (n >= 0xFFFFFFFFull / 2) { set_error("pann_vamana_build_single_batch: n too large for one batch"); return PANN_ERR_BAD_ARG; }
  // (generated into the handle's pinned staging: no first-touch page faults per build -- they cost more than the shuffle -- and a
  // DMA transfer without a bounce buffer)
  if (int rc = idx->pin_in.ensure(n * 4)) return rc;
  uint32_t* perm = static_cast<uint32_t*>(idx->pin_in.p);
  build_permutation(n, seed, perm);
  if (int rc = idx->stage[2].ensure(n * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, perm, n * 4, hipMemcpyHostToDevice, idx->stream));
  {
    const uint64_t tot = n * idx->ix.gstride;
    hipLaunchKernelGGL(random_edges_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, idx->stream, idx->ix.graph,
                       idx->ix.gstride, n, degree, seed);
    PANN_HIP(hipGetLastError());
  }
  PANN_HIP(hipStreamSynchronize(idx->stream));
  idx->ix.codes_valid = 0;                              // random_edges_kernel wrote rows
  if (int rc = ensure_filter_codes(idx, L)) return rc;
  if (idx->vcap < default_vcap(L)) idx->vcap = default_vcap(L);
  for (int pass = 0; pass < num_passes; pass++) {
    const double a = (pass == num_passes - 1) ? alpha : 1.0;   // :173-178
    if (int rc = insert_batch_dev(idx->ix, idx->ws2, idx->ws3, idx->ws, idx->ws4, idx->stream, idx->stage[2].as<uint32_t>(),
                                  (uint32_t)n, 0u, R, L, a, &idx->vcap, stats))      // floor = 0, ceiling = m (:236-240)
      return rc;
  }
  if (sort_neighbors) { idx->ix.codes_valid = 0; return sort_neighbors_dev(idx->ix, idx->stream); }
  return PANN_OK;
}

int pann_vamana_build(pann_index* idx, uint32_t R, uint32_t L, double alpha, int num_passes, uint64_t seed,
                      int sort_neighbors, pann_build_stats* stats) {
  if (int rc = check_idx(idx, "pann_vamana_build")) return rc;
  if (L == 0 || L > 65536 || num_passes < 1) { set_error("pann_vamana_build: bad L / num_passes"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  const uint64_t n = idx->ix.n;
  // (generated into the handle's pinned staging: no first-touch page faults per build -- they cost more than the shuffle -- and a
  // DMA transfer without a bounce buffer)
  if (int rc = idx->pin_in.ensure(n * 4)) return rc;
  uint32_t* perm = static_cast<uint32_t*>(idx->pin_in.p);
  build_permutation(n, seed, perm);
  if (int rc = idx->stage[2].ensure(n * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, perm, n * 4, hipMemcpyHostToDevice, idx->stream));
  PANN_HIP(hipStreamSynchronize(idx->stream));
  const uint32_t* d_perm = idx->stage[2].as<uint32_t>();
  if (int rc = ensure_filter_codes(idx, L)) return rc;
  if (idx->vcap < default_vcap(L)) idx->vcap = default_vcap(L);
  // vamana/index.h:206-209
  size_t max_batch = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);
  if (max_batch == 0) max_batch = n;
  for (int pass = 0; pass < num_passes; pass++) {
    const double a = (pass == num_passes - 1) ? alpha : 1.0;   // :173-178
    size_t count = 0, inc = 0;
    while (count < n) {                                         // :223-234
      size_t floor, ceiling;
      if (std::pow(2.0, (double)inc) <= (double)max_batch) {
        floor = (size_t)std::pow(2.0, (double)inc) - 1;
        ceiling = std::min((size_t)std::pow(2.0, (double)(inc + 1)) - 1, (size_t)n);
        count = ceiling;
      } else {
        floor = count;
        ceiling = std::min(count + max_batch, (size_t)n);
        count += max_batch;
      }
      if (int rc = insert_batch_dev(idx->ix, idx->ws2, idx->ws3, idx->ws, idx->ws4, idx->stream, d_perm + floor,
                                    (uint32_t)(ceiling - floor), 0u /* set_start(): vertex 0, :148 */, R, L, a,
                                    &idx->vcap, stats))
        return rc;
      inc++;
    }
  }
  if (sort_neighbors) { idx->ix.codes_valid = 0; return sort_neighbors_dev(idx->ix, idx->stream); }   // :180-185
  return PANN_OK;
}


// ---------------------------------------------------------------------------------------------
// distances, HCNNG leaf kNN, brute-force ground truth
// ---------------------------------------------------------------------------------------------

int pann_pair_distances(pann_index* idx, const uint32_t* a_ids, const uint32_t* b_ids, uint64_t m, float* out) {
  if (int rc = check_idx(idx, "pann_pair_distances")) return rc;
  if (m == 0) return PANN_OK;
  if (!a_ids || !b_ids || !out) { set_error("pann_pair_distances: null argument"); return PANN_ERR_BAD_ARG; }
  for (uint64_t i = 0; i < m; i++)
    if (a_ids[i] >= idx->ix.n || b_ids[i] >= idx->ix.n) { set_error("pann_pair_distances: id out of range"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  if (int rc = idx->stage[2].ensure(m * 4)) return rc;
  if (int rc = idx->stage[3].ensure(m * 4)) return rc;
  if (int rc = idx->stage[4].ensure(m * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, a_ids, m * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, b_ids, m * 4, hipMemcpyHostToDevice, st));
  if (int rc = query_distances_dev(idx->ix, st, nullptr, 0, idx->stage[2].as<uint32_t>(), m, idx->stage[3].as<uint32_t>(), m, 1,
                                   idx->stage[4].as<float>())) return rc;
  PANN_HIP(hipMemcpyAsync(out, idx->stage[4].p, m * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

int pann_query_distances(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes,
                         const uint32_t* ids, uint64_t m, float* out) {
  if (int rc = check_idx(idx, "pann_query_distances")) return rc;
  if (nq == 0 || m == 0) return PANN_OK;
  if (!queries || !ids || !out) { set_error("pann_query_distances: null argument"); return PANN_ERR_BAD_ARG; }
  if (q_stride_bytes < idx->ix.dbytes) { set_error("pann_query_distances: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  for (uint64_t i = 0; i < m; i++)
    if (ids[i] >= idx->ix.n) { set_error("pann_query_distances: id out of range"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  if (int rc = idx->stage[2].ensure(nq * q_stride_bytes + 16)) return rc;
  if (int rc = idx->stage[3].ensure(m * 4)) return rc;
  if (int rc = idx->stage[4].ensure(nq * m * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, queries, (nq - 1) * q_stride_bytes + idx->ix.dbytes, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, ids, m * 4, hipMemcpyHostToDevice, st));
  if (int rc = query_distances_dev(idx->ix, st, idx->stage[2].as<uint8_t>(), q_stride_bytes, nullptr, nq,
                                   idx->stage[3].as<uint32_t>(), m, 0, idx->stage[4].as<float>())) return rc;
  PANN_HIP(hipMemcpyAsync(out, idx->stage[4].p, nq * m * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

int pann_leaf_knn_batch(pann_index* idx, const uint32_t* ids, const uint64_t* leaf_offsets, uint64_t nleaves,
                        uint32_t m, uint32_t* out_ids, float* out_dists) {
  if (int rc = check_idx(idx, "pann_leaf_knn_batch")) return rc;
  if (nleaves == 0) return PANN_OK;
  if (!ids || !leaf_offsets || !out_ids || !out_dists) { set_error("pann_leaf_knn: null argument"); return PANN_ERR_BAD_ARG; }
  const uint64_t total = leaf_offsets[nleaves];
  if (total == 0) return PANN_OK;
  for (uint64_t i = 0; i < total; i++)
    if (ids[i] >= idx->ix.n) { set_error("pann_leaf_knn: id out of range"); return PANN_ERR_BAD_ARG; }
  std::vector<uint32_t> tseg, ta0;
  for (uint64_t s = 0; s < nleaves; s++) {
    if (leaf_offsets[s + 1] < leaf_offsets[s]) { set_error("pann_leaf_knn: offsets not monotone"); return PANN_ERR_BAD_ARG; }
    for (uint64_t a = leaf_offsets[s]; a < leaf_offsets[s + 1]; a += 64) { tseg.push_back((uint32_t)s); ta0.push_back((uint32_t)a); }
  }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  const size_t nt = tseg.size();
  if (int rc = idx->stage[2].ensure(total * 4)) return rc;
  if (int rc = idx->stage[3].ensure((nleaves + 1) * 8)) return rc;
  if (int rc = idx->stage[4].ensure(nt * 4)) return rc;
  if (int rc = idx->stage[5].ensure(nt * 4)) return rc;
  if (int rc = idx->stage[6].ensure(total * m * 4)) return rc;
  if (int rc = idx->stage[7].ensure(total * m * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, ids, total * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, leaf_offsets, (nleaves + 1) * 8, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[4].p, tseg.data(), nt * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[5].p, ta0.data(), nt * 4, hipMemcpyHostToDevice, st));
  if (leaf_knn_rows_eligible(idx->ix, m)) {     // one-byte element types: lane-owns-row kernel (leaf_knn.hip)
    if (int rc = leaf_knn_rows_dev(idx->ix, idx->ws2, st, idx->stage[2].as<uint32_t>(), idx->stage[3].as<uint64_t>(), leaf_offsets, nleaves, m, 1,
                                   idx->stage[6].as<uint32_t>(), idx->stage[7].as<float>())) return rc;
  } else if (int rc = dense_topk_dev(idx->ix, idx->ws2, st, nullptr, 0, idx->stage[2].as<uint32_t>(), idx->stage[2].as<uint32_t>(),
                              idx->stage[3].as<uint64_t>(), idx->stage[3].as<uint64_t>(), idx->stage[4].as<uint32_t>(),
                              idx->stage[5].as<uint32_t>(), (uint32_t)nt, total, total, 1, m, 1,
                              idx->stage[6].as<uint32_t>(), idx->stage[7].as<float>())) return rc;
  PANN_HIP(hipMemcpyAsync(out_ids, idx->stage[6].p, total * m * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipMemcpyAsync(out_dists, idx->stage[7].p, total * m * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}

int pann_leaf_knn(pann_index* idx, const uint32_t* ids, uint32_t N, uint32_t m, uint32_t* out_ids, float* out_dists) {
  const uint64_t off[2] = {0, N};
  return pann_leaf_knn_batch(idx, ids, off, 1, m, out_ids, out_dists);
}

int pann_bruteforce_knn(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes, uint32_t k,
                        uint32_t* out_ids, float* out_dists) {
  if (int rc = check_idx(idx, "pann_bruteforce_knn")) return rc;
  if (nq == 0) return PANN_OK;
  if (!queries || !out_ids || !out_dists) { set_error("pann_bruteforce_knn: null argument"); return PANN_ERR_BAD_ARG; }
  if (q_stride_bytes < idx->ix.dbytes) { set_error("pann_bruteforce_knn: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  if (int rc = idx->stage[2].ensure(nq * q_stride_bytes + 16)) return rc;
  if (int rc = idx->stage[6].ensure(nq * k * 4)) return rc;
  if (int rc = idx->stage[7].ensure(nq * k * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, queries, (nq - 1) * q_stride_bytes + idx->ix.dbytes, hipMemcpyHostToDevice, st));
  const uint32_t ntiles = (uint32_t)((nq + 63) / 64);
  // B is cut into nsplit pieces per A tile.  Every piece warms up its own top-k lists (about k * ln(piece / k) + k list inserts per
  // query: 10K x 1M, k = 100 spent 7.5 G instructions there at 14 pieces), and at k = 100 the lists leave room for ONE workgroup
  // per CU, so what matters is how evenly ntiles * nsplit workgroups fill whole rounds of the 256 CUs: the smallest count (<= 8,
  // or enough to reach every CU when there are few queries) with the best fill wins (10K queries: 157 tiles x 3 = 1.84 rounds,
  // 43 ms; x 1: 53 ms; x 14: 68 ms)
  uint32_t want = 1;
  {
    double best = -1.0;
    const double slots = (double)dense_gt_slots(idx->ix, k);       // 256 x the workgroups a CU holds (1 for the LDS-list kernels)
    const uint32_t smax = std::max<uint32_t>(8, ((uint32_t)slots + ntiles - 1) / ntiles);
    for (uint32_t sp = 1; sp <= smax; sp++) {
      const double wgs = (double)ntiles * sp;
      const double fill = wgs / (slots * std::ceil(wgs / slots)) - 0.02 * std::min<uint32_t>(sp, 8);
      if (fill > best + 1e-9) { best = fill; want = sp; }
    }
  }
  const uint32_t env_split = idx->gt_pieces;      // pann_index_set_option("gt_pieces")
  uint32_t nsplit = std::max<uint32_t>(1, std::min<uint32_t>(env_split ? env_split : want, (uint32_t)((idx->ix.n + 4095) / 4096)));
  nsplit = std::min<uint32_t>(nsplit, 64);
  if (int rc = dense_topk_dev(idx->ix, idx->ws2, st, idx->stage[2].as<uint8_t>(), q_stride_bytes, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, ntiles, nq, idx->ix.n, nsplit, k, 0, idx->stage[6].as<uint32_t>(),
                              idx->stage[7].as<float>())) return rc;
  PANN_HIP(hipMemcpyAsync(out_ids, idx->stage[6].p, nq * k * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipMemcpyAsync(out_dists, idx->stage[7].p, nq * k * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}


int pann_pivot_split(pann_index* idx, const uint32_t* ids, const uint64_t* seg_offsets, uint64_t nseg,
                     const uint32_t* pivot_a, const uint32_t* pivot_b, uint8_t* out_side) {
  if (int rc = check_idx(idx, "pann_pivot_split")) return rc;
  if (nseg == 0) return PANN_OK;
  if (!ids || !seg_offsets || !pivot_a || !pivot_b || !out_side) { set_error("pann_pivot_split: null argument"); return PANN_ERR_BAD_ARG; }
  const uint64_t total = seg_offsets[nseg];
  if (total == 0) return PANN_OK;
  for (uint64_t i = 0; i < total; i++)
    if (ids[i] >= idx->ix.n) { set_error("pann_pivot_split: id out of range"); return PANN_ERR_BAD_ARG; }
  std::vector<uint32_t> tseg, tcnt; std::vector<uint64_t> tlo;
  for (uint64_t s = 0; s < nseg; s++) {
    if (pivot_a[s] >= idx->ix.n || pivot_b[s] >= idx->ix.n) { set_error("pann_pivot_split: pivot out of range"); return PANN_ERR_BAD_ARG; }
    for (uint64_t a = seg_offsets[s]; a < seg_offsets[s + 1]; a += 64) {
      tseg.push_back((uint32_t)s); tlo.push_back(a); tcnt.push_back((uint32_t)std::min<uint64_t>(64, seg_offsets[s + 1] - a));
    }
  }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  const size_t nt = tseg.size();
  if (int rc = idx->stage[2].ensure(total * 4)) return rc;
  if (int rc = idx->stage[3].ensure(nt * 8)) return rc;
  if (int rc = idx->stage[4].ensure(nt * 4)) return rc;
  if (int rc = idx->stage[5].ensure(nt * 4)) return rc;
  if (int rc = idx->stage[6].ensure(nseg * 4)) return rc;
  if (int rc = idx->stage[7].ensure(nseg * 4)) return rc;
  if (int rc = idx->stage[8].ensure(total)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, ids, total * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, tlo.data(), nt * 8, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[4].p, tseg.data(), nt * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[5].p, tcnt.data(), nt * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[6].p, pivot_a, nseg * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[7].p, pivot_b, nseg * 4, hipMemcpyHostToDevice, st));
  if (int rc = pivot_split_dev(idx->ix, st, idx->stage[2].as<uint32_t>(), idx->stage[4].as<uint32_t>(), idx->stage[3].as<uint64_t>(),
                               idx->stage[5].as<uint32_t>(), (uint32_t)nt, idx->stage[6].as<uint32_t>(), idx->stage[7].as<uint32_t>(),
                               idx->stage[8].as<uint8_t>())) return rc;
  PANN_HIP(hipMemcpyAsync(out_side, idx->stage[8].p, total, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}


int pann_rerank(pann_index* idx, const void* queries, uint64_t nq, uint64_t q_stride_bytes, const uint32_t* cand_ids,
                uint32_t c, const uint32_t* cand_counts, uint32_t k, int resort, uint32_t* out_ids, float* out_dists) {
  if (int rc = check_idx(idx, "pann_rerank")) return rc;
  if (nq == 0) return PANN_OK;
  if (!queries || !cand_ids || !out_ids || !out_dists || k == 0) { set_error("pann_rerank: null argument"); return PANN_ERR_BAD_ARG; }
  if (q_stride_bytes < idx->ix.dbytes) { set_error("pann_rerank: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  for (uint64_t i = 0; i < nq; i++) {
    const uint32_t cn = cand_counts ? std::min(cand_counts[i], c) : c;
    for (uint32_t j = 0; j < cn; j++)
      if (cand_ids[i * c + j] >= idx->ix.n) { set_error("pann_rerank: candidate id out of range"); return PANN_ERR_BAD_ARG; }
  }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  if (int rc = idx->stage[2].ensure(nq * q_stride_bytes + 16)) return rc;
  if (int rc = idx->stage[3].ensure(nq * c * 4)) return rc;
  if (int rc = idx->stage[4].ensure(nq * 4)) return rc;
  if (int rc = idx->stage[6].ensure(nq * k * 4)) return rc;
  if (int rc = idx->stage[7].ensure(nq * k * 4)) return rc;
  PANN_HIP(hipMemcpyAsync(idx->stage[2].p, queries, (nq - 1) * q_stride_bytes + idx->ix.dbytes, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, cand_ids, nq * c * 4, hipMemcpyHostToDevice, st));
  if (cand_counts) PANN_HIP(hipMemcpyAsync(idx->stage[4].p, cand_counts, nq * 4, hipMemcpyHostToDevice, st));
  if (int rc = rerank_dev(idx->ix, st, idx->stage[2].as<uint8_t>(), q_stride_bytes, nq, idx->stage[3].as<uint32_t>(), c,
                          cand_counts ? idx->stage[4].as<uint32_t>() : nullptr, k, resort, idx->stage[6].as<uint32_t>(),
                          idx->stage[7].as<float>())) return rc;
  PANN_HIP(hipMemcpyAsync(out_ids, idx->stage[6].p, nq * k * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipMemcpyAsync(out_dists, idx->stage[7].p, nq * k * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}


int pann_range_search(pann_index* idx, const void* queries, const uint32_t* query_ids, uint64_t nq,
                      uint64_t q_stride_bytes, const uint32_t* starts, uint32_t nstarts, int starts_per_query,
                      float radius_2, uint32_t max_results, uint32_t* out_ids, uint32_t* out_counts,
                      uint32_t* out_dist_cmps, uint32_t* out_truncated) {
  if (int rc = check_idx(idx, "pann_range_search")) return rc;
  if (nq == 0) return PANN_OK;
  const DeviceIndex& ix = idx->ix;
  if ((queries == nullptr) == (query_ids == nullptr)) { set_error("pann_range_search: exactly one of queries / query_ids must be given"); return PANN_ERR_BAD_ARG; }
  if (!starts || nstarts == 0 || !out_ids || !out_counts || max_results == 0) { set_error("pann_range_search: null or empty argument"); return PANN_ERR_BAD_ARG; }
  if (queries && q_stride_bytes < ix.dbytes) { set_error("pann_range_search: query stride smaller than a row"); return PANN_ERR_BAD_ARG; }
  if (query_ids)
    for (uint64_t i = 0; i < nq; i++)
      if (query_ids[i] >= ix.n) { set_error("pann_range_search: query id out of range"); return PANN_ERR_BAD_ARG; }
  const uint64_t ns_total = (starts_per_query ? nq : 1) * (uint64_t)nstarts;
  for (uint64_t i = 0; i < ns_total; i++)
    if (starts[i] != SENTINEL && starts[i] >= ix.n) { set_error("pann_range_search: start id out of range"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  hipStream_t st = idx->stream;
  const size_t qbytes = queries ? nq * q_stride_bytes + 16 : nq * 4;
  if (int rc = idx->stage[2].ensure(qbytes)) return rc;
  if (int rc = idx->stage[3].ensure(ns_total * 4)) return rc;
  if (int rc = idx->stage[4].ensure(nq * (uint64_t)max_results * 4)) return rc;
  if (int rc = idx->stage[5].ensure(nq * 4)) return rc;
  if (int rc = idx->stage[6].ensure(nq * 4)) return rc;
  if (int rc = idx->stage[7].ensure(nq * 4)) return rc;
  if (queries) PANN_HIP(hipMemcpyAsync(idx->stage[2].p, queries, (nq - 1) * q_stride_bytes + ix.dbytes, hipMemcpyHostToDevice, st));
  else PANN_HIP(hipMemcpyAsync(idx->stage[2].p, query_ids, nq * 4, hipMemcpyHostToDevice, st));
  PANN_HIP(hipMemcpyAsync(idx->stage[3].p, starts, ns_total * 4, hipMemcpyHostToDevice, st));
  if (int rc = range_search_dev(ix, idx->ws, st, queries ? idx->stage[2].as<uint8_t>() : nullptr, q_stride_bytes,
                                queries ? nullptr : idx->stage[2].as<uint32_t>(), nq, idx->stage[3].as<uint32_t>(), nstarts,
                                starts_per_query, radius_2, max_results, idx->stage[4].as<uint32_t>(),
                                idx->stage[5].as<uint32_t>(), idx->stage[6].as<uint32_t>(), idx->stage[7].as<uint32_t>())) return rc;
  // only the columns any query filled come back (entries past a row's count are unspecified: include/pann.h)
  PANN_HIP(hipMemcpyAsync(out_counts, idx->stage[5].p, nq * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  uint32_t widest = 0;
  for (uint64_t i = 0; i < nq; i++) widest = std::max(widest, out_counts[i]);
  widest = std::min(widest, max_results);
  if (widest)
    PANN_HIP(hipMemcpy2DAsync(out_ids, (size_t)max_results * 4, idx->stage[4].p, (size_t)max_results * 4, (size_t)widest * 4, nq,
                              hipMemcpyDeviceToHost, st));
  if (out_dist_cmps) PANN_HIP(hipMemcpyAsync(out_dist_cmps, idx->stage[6].p, nq * 4, hipMemcpyDeviceToHost, st));
  if (out_truncated) PANN_HIP(hipMemcpyAsync(out_truncated, idx->stage[7].p, nq * 4, hipMemcpyDeviceToHost, st));
  PANN_HIP(hipStreamSynchronize(st));
  return PANN_OK;
}


int pann_hcnng_build_trees_dev(pann_index* idx, uint32_t first_tree, uint32_t tree_step, uint32_t ntrees, uint32_t cluster_size,
                               uint32_t mst_deg, uint64_t seed, uint32_t* d_slab, uint32_t slab_stride, double* times3) {
  if (int rc = check_idx(idx, "pann_hcnng_build_trees_dev")) return rc;
  if (mst_deg == 0 || tree_step == 0 || !d_slab) { set_error("pann_hcnng_build_trees_dev: null / zero argument"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, idx->stream, d_slab, idx->ix.n * (uint64_t)slab_stride, SENTINEL);
  PANN_HIP(hipGetLastError());
  if (ntrees == 0) { PANN_HIP(hipStreamSynchronize(idx->stream)); return PANN_OK; }
  return hcnng_build_dev(idx->ix, idx->ws2, idx->stream, ntrees, cluster_size, mst_deg, seed, times3, first_tree, tree_step, d_slab, slab_stride);
}

int pann_hcnng_assemble_dev(pann_index* idx, const uint32_t* d_slabs, uint32_t nslabs, uint32_t slab_stride, uint32_t ntrees,
                            uint32_t mst_deg) {
  if (int rc = check_idx(idx, "pann_hcnng_assemble_dev")) return rc;
  if (!d_slabs || mst_deg == 0) { set_error("pann_hcnng_assemble_dev: null / zero argument"); return PANN_ERR_BAD_ARG; }
  if ((uint64_t)ntrees * mst_deg > idx->ix.max_deg) { set_error("pann_hcnng_assemble_dev: max_deg < ntrees * mst_deg"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  idx->ix.codes_valid = 0;
  return hcnng_assemble_dev(idx->ix, idx->stream, d_slabs, nslabs, slab_stride, ntrees, mst_deg);
}

int pann_hcnng_build(pann_index* idx, uint32_t num_clusters, uint32_t cluster_size, uint32_t mst_deg, uint64_t seed,
                     double* times3) {
  if (int rc = check_idx(idx, "pann_hcnng_build")) return rc;
  if (num_clusters == 0 || mst_deg == 0) { set_error("pann_hcnng_build: num_clusters and mst_deg must be positive"); return PANN_ERR_BAD_ARG; }
  DeviceGuard g(idx->device);
  idx->ix.codes_valid = 0;
  return hcnng_build_dev(idx->ix, idx->ws2, idx->stream, num_clusters, cluster_size, mst_deg, seed, times3);
}

}  // extern "C"
