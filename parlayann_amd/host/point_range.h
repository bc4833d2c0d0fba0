// point_range.h -- host mirror of algorithms/utils/point_range.h:42-141 and the point "view" types
// of euclidian_point.h:92-242 / mips_point.h:67-139.  The views carry the type, metric and
// parameters; distance arithmetic is NOT done on the host (it lives in libpann.so).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "../../include/pann.h"

namespace parlayANN {

template <typename T> struct pann_dtype_of;
template <> struct pann_dtype_of<uint8_t> { static constexpr int value = PANN_U8; };
template <> struct pann_dtype_of<int8_t> { static constexpr int value = PANN_I8; };
template <> struct pann_dtype_of<float> { static constexpr int value = PANN_F32; };
struct half_t { uint16_t bits; };   // storage-only fp16 (this build's extension)
template <> struct pann_dtype_of<half_t> { static constexpr int value = PANN_F16; };
struct bf16_t { uint16_t bits; };   // storage-only bfloat16 (this build's extension)
template <> struct pann_dtype_of<bf16_t> { static constexpr int value = PANN_BF16; };

template <typename T_, int METRIC>
struct Point_ {
  using T = T_;
  using distanceType = float;
  using byte = uint8_t;
  struct parameters {
    int dims = 0;
    int num_bytes() const { return dims * (int)sizeof(T); }
    parameters() {}
    explicit parameters(int dims) : dims(dims) {}
  };
  static constexpr int metric = METRIC;
  static bool is_metric() { return METRIC == PANN_L2; }   // euclidian_point.h:112, mips_point.h:82
  const T* values = nullptr;
  long id_ = -1;
  parameters params;
  Point_() {}
  Point_(const byte* v, long id, parameters p) : values((const T*)v), id_(id), params(p) {}
  long id() const { return id_; }
  T operator[](long i) const { return values[i]; }
  bool same_as(const Point_& q) const { return values == q.values; }      // euclidian_point.h:178-180
  bool operator==(const Point_& q) const {                                // :170-176
    for (int i = 0; i < params.dims; i++) if (std::memcmp(&values[i], &q.values[i], sizeof(T)) != 0) return false;
    return true;
  }
};
template <typename T> using Euclidian_Point = Point_<T, PANN_L2>;
template <typename T> using Mips_Point = Point_<T, PANN_MIPS>;

template <class Point_T>
struct PointRange {
  using Point = Point_T;
  using parameters = typename Point::parameters;
  using byte = uint8_t;
  using T = typename Point::T;

  long dimension() const { return dims; }
  size_t size() const { return n; }
  unsigned int get_aligned_bytes() const { return aligned_bytes; }
  const byte* data() const { return values.get(); }
  byte* data() { return values.get(); }

  PointRange() {}

  // n x d slab from memory (row-major, tightly packed)
  PointRange(const T* src, size_t n_, unsigned int d) : dims(d), n(n_) {
    params = parameters((int)d);
    allocate();
    for (size_t i = 0; i < n; i++) std::memcpy(values.get() + i * aligned_bytes, src + i * d, (size_t)d * sizeof(T));
  }

  // [n:u32][d:u32][n*d*sizeof(T)]  (point_range.h:74-117)
  explicit PointRange(const char* filename) {
    if (filename == nullptr) return;
    std::ifstream reader(filename, std::ios::binary);
    if (!reader.is_open()) { std::cout << "Data file " << filename << " not found" << std::endl; std::abort(); }
    unsigned int num_points, d;
    reader.read((char*)&num_points, 4);
    reader.read((char*)&d, 4);
    n = num_points; dims = d;
    params = parameters((int)d);
    std::cout << "Detected " << num_points << " points with dimension " << d << std::endl;
    allocate();
    const size_t BLOCK = 1000000;
    std::unique_ptr<T[]> buf(new T[BLOCK * (size_t)d]);
    for (size_t lo = 0; lo < n; lo += BLOCK) {
      const size_t cnt = std::min(BLOCK, n - lo);
      reader.read((char*)buf.get(), (std::streamsize)(cnt * d * sizeof(T)));
      for (size_t i = 0; i < cnt; i++)
        std::memcpy(values.get() + (lo + i) * aligned_bytes, buf.get() + i * d, (size_t)d * sizeof(T));
    }
  }

  Point operator[](long i) const {
    if (i > (long)n) { std::cout << "ERROR: point index out of range: " << i << " from range " << n << std::endl; std::abort(); }
    return Point(values.get() + (size_t)i * aligned_bytes, i, params);
  }
  byte* location(long i) const { return values.get() + (size_t)i * aligned_bytes; }

  // -- device-mirror bookkeeping (no counterpart upstream): writes through location() / data() are not tracked;
  // call touch() after changing coordinates in place (normalize_range does)
  uint64_t version() const { return version_ ? version_->load(std::memory_order_relaxed) : 0; }
  void touch() const { if (version_) version_->fetch_add(1, std::memory_order_relaxed); }
  const std::shared_ptr<byte[]>& slab_handle() const { return values; }

  parameters params;

 private:
  void allocate() {
    const long num_bytes = (long)dims * (long)sizeof(T);
    aligned_bytes = (unsigned int)(64 * ((num_bytes - 1) / 64 + 1));     // point_range.h:94
    const size_t total = std::max<size_t>((n * (size_t)aligned_bytes + ((1ul << 21) - 1)) & ~((1ul << 21) - 1), 1ul << 21);
    byte* ptr = (byte*)aligned_alloc(1l << 21, total);
    std::memset(ptr, 0, total);
    values = std::shared_ptr<byte[]>(ptr, std::free);
    version_ = std::make_shared<std::atomic<uint64_t>>(1);
  }
  std::shared_ptr<byte[]> values;
  std::shared_ptr<std::atomic<uint64_t>> version_;
  unsigned int dims = 0;
  unsigned int aligned_bytes = 0;
  size_t n = 0;
};

}  // namespace parlayANN
