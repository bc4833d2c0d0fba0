// graph.h -- host mirror of the Graph / edgeRange SURFACE of algorithms/utils/graph.h (edgeRange :41-123,
// Graph :125-250): same type names, methods, in-memory row layout (n rows of maxDeg+1 ids, slot 0 = degree)
// and file format ([n][maxDeg][degree of every vertex][all edges], :155-160,:210-231), written for this
// repo: one zero-initialised slab, C stdio for the files, a single bounds/abort helper.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

namespace parlayANN {

namespace graph_detail {
[[noreturn]] inline void die(const char* what, long value) {
  // the reference reports on stdout and aborts (graph.h:56-58,63-66,235-238)
  std::cout << what << value << std::endl;
  std::abort();
}
inline size_t round_up_2mb(size_t bytes) {
  const size_t two_mb = size_t(1) << 21;
  return std::max(two_mb, (bytes + two_mb - 1) / two_mb * two_mb);
}
}  // namespace graph_detail

// view of one adjacency row: row[0] = degree, row[1 .. degree] = neighbours
template <typename indexType>
struct edgeRange {
  edgeRange() = default;
  edgeRange(indexType* first, indexType* last, indexType owner, std::atomic<uint64_t>* version = nullptr)
      : row_(first), cap_(last - first - 1), owner_(owner), version_(version) {}

  size_t size() const { return row_[0]; }
  indexType id() const { return owner_; }

  indexType operator[](indexType j) const {
    if (j > row_[0]) graph_detail::die("ERROR: index exceeds degree while accessing neighbors: ", (long)j);
    return row_[1 + j];
  }

  void append_neighbor(indexType nbh) {
    if ((long)row_[0] == cap_) graph_detail::die("ERROR in append_neighbor: cannot exceed max degree ", cap_);
    row_[++row_[0]] = nbh;
    dirty();
  }

  template <typename rangeType>
  void append_neighbors(const rangeType& r) {
    if ((long)(row_[0] + r.size()) > cap_)
      graph_detail::die("ERROR in append_neighbors: cannot exceed max degree ", cap_);
    indexType* out = row_ + 1 + row_[0];
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    row_[0] += (indexType)r.size();
    dirty();
  }

  template <typename rangeType>
  void update_neighbors(const rangeType& r) {
    if ((long)r.size() > cap_) graph_detail::die("ERROR in update_neighbors: cannot exceed max degree ", cap_);
    for (size_t i = 0; i < r.size(); i++) row_[1 + i] = r[i];
    row_[0] = (indexType)r.size();
    dirty();
  }

  void clear_neighbors() { row_[0] = 0; dirty(); }

  template <typename F>
  void sort(F&& less) { std::sort(begin(), end(), less); dirty(); }

  indexType* begin() { return row_ + 1; }
  indexType* end() { return row_ + 1 + row_[0]; }

 private:
  // every mutation bumps the owning Graph's version: device mirrors of the graph (device_mirror.h) re-upload
  // when the version they hold is stale
  void dirty() { if (version_) version_->fetch_add(1, std::memory_order_relaxed); }
  indexType* row_ = nullptr;
  long cap_ = 0;
  indexType owner_ = 0;
  std::atomic<uint64_t>* version_ = nullptr;
};

template <typename indexType_>
struct Graph {
  using indexType = indexType_;

  Graph() = default;
  Graph(long maxDeg, size_t n) : n_(n), max_deg_(maxDeg) { allocate(); }

  // [n][maxDeg][deg_0 .. deg_{n-1}][edges of vertex 0][edges of vertex 1]...
  explicit Graph(const char* gFile) {
    FILE* f = std::fopen(gFile, "rb");
    if (!f) { std::cout << "graph file " << gFile << " not found" << std::endl; std::abort(); }
    indexType header[2];
    if (std::fread(header, sizeof(indexType), 2, f) != 2) graph_detail::die("graph file too short: ", 0);
    n_ = header[0];
    max_deg_ = header[1];
    std::cout << "Graph: detected " << n_ << " points with max degree " << max_deg_ << std::endl;
    allocate();
    std::vector<indexType> degrees(n_);
    if (n_ && std::fread(degrees.data(), sizeof(indexType), n_, f) != n_) graph_detail::die("graph file too short: ", 1);
    size_t total = 0;
    for (size_t v = 0; v < n_; v++) {
      indexType* row = row_ptr(v);
      const size_t d = std::min<size_t>(degrees[v], (size_t)max_deg_);
      if (d && std::fread(row + 1, sizeof(indexType), d, f) != d) graph_detail::die("graph file too short: ", 2);
      if (degrees[v] > d) std::fseek(f, (long)((degrees[v] - d) * sizeof(indexType)), SEEK_CUR);
      row[0] = (indexType)d;
      total += d;
    }
    std::fclose(f);
    std::cout << "Total edges read from file: " << total << std::endl;
  }

  void save(const char* oFile) const {
    std::cout << "Writing graph with " << n_ << " points and max degree " << max_deg_ << std::endl;
    FILE* f = std::fopen(oFile, "wb");
    if (!f) graph_detail::die("cannot open graph output file: ", 0);
    const indexType header[2] = {(indexType)n_, (indexType)max_deg_};
    std::fwrite(header, sizeof(indexType), 2, f);
    std::vector<indexType> degrees(n_);
    for (size_t v = 0; v < n_; v++) degrees[v] = row_ptr(v)[0];
    std::fwrite(degrees.data(), sizeof(indexType), n_, f);
    for (size_t v = 0; v < n_; v++) std::fwrite(row_ptr(v) + 1, sizeof(indexType), degrees[v], f);
    std::fclose(f);
  }

  long max_degree() const { return max_deg_; }
  size_t size() const { return n_; }
  indexType* data() { return slab_.get(); }
  const indexType* data() const { return slab_.get(); }

  // -- device-mirror bookkeeping (no counterpart upstream): a change counter and the owning handle of the slab.
  // Writes through data() are not seen by the counter: call touch() after them.
  uint64_t version() const { return version_ ? version_->load(std::memory_order_relaxed) : 0; }
  void touch() const { if (version_) version_->fetch_add(1, std::memory_order_relaxed); }
  const std::shared_ptr<indexType[]>& slab_handle() const { return slab_; }

  edgeRange<indexType> operator[](indexType v) const {
    if (v > n_) graph_detail::die("ERROR: graph index out of range: ", (long)v);
    indexType* row = row_ptr(v);
    return edgeRange<indexType>(row, row + max_deg_ + 1, v, version_.get());
  }

 private:
  indexType* row_ptr(size_t v) const { return slab_.get() + v * (size_t)(max_deg_ + 1); }
  void allocate() {
    const size_t bytes = graph_detail::round_up_2mb(n_ * (size_t)(max_deg_ + 1) * sizeof(indexType));
    void* p = aligned_alloc(size_t(1) << 21, bytes);       // 2 MiB aligned like the reference slab (graph.h:136)
    std::memset(p, 0, bytes);
    slab_ = std::shared_ptr<indexType[]>(static_cast<indexType*>(p), std::free);
    version_ = std::make_shared<std::atomic<uint64_t>>(1);
  }
  size_t n_ = 0;
  long max_deg_ = 0;
  std::shared_ptr<indexType[]> slab_;
  std::shared_ptr<std::atomic<uint64_t>> version_;   // shared by the (shallow) copies of this Graph, like the slab
};

}  // namespace parlayANN
