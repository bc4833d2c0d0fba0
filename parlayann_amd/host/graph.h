// graph.h -- host mirror of algorithms/utils/graph.h: Graph<indexType> (:125-250) and edgeRange
// (:41-123).  Flat n x (maxDeg+1) slab, slot 0 = degree; same file format (:155-160,:210-231).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

namespace parlayANN {

template <typename indexType>
struct edgeRange {
  size_t size() const { return edges[0]; }
  indexType id() const { return id_; }
  edgeRange() : edges(nullptr), maxDeg(0) {}
  edgeRange(indexType* start, indexType* end, indexType id) : edges(start), maxDeg((long)(end - start - 1)), id_(id) {}

  indexType operator[](indexType j) const {
    if (j > edges[0]) { std::cout << "ERROR: index exceeds degree while accessing neighbors" << std::endl; abort(); }
    return edges[j + 1];
  }
  void append_neighbor(indexType nbh) {
    if (edges[0] == (indexType)maxDeg) { std::cout << "ERROR in append_neighbor: cannot exceed max degree " << maxDeg << std::endl; abort(); }
    edges[edges[0] + 1] = nbh;
    edges[0] += 1;
  }
  template <typename rangeType>
  void update_neighbors(const rangeType& r) {
    if ((long)r.size() > maxDeg) { std::cout << "ERROR in update_neighbors: cannot exceed max degree " << maxDeg << std::endl; abort(); }
    edges[0] = (indexType)r.size();
    for (size_t i = 0; i < r.size(); i++) edges[i + 1] = r[i];
  }
  template <typename rangeType>
  void append_neighbors(const rangeType& r) {
    if ((long)(r.size() + edges[0]) > maxDeg) { std::cout << "ERROR in append_neighbors for point " << id_ << ": cannot exceed max degree " << maxDeg << std::endl; abort(); }
    for (size_t i = 0; i < r.size(); i++) edges[edges[0] + i + 1] = r[i];
    edges[0] += (indexType)r.size();
  }
  void clear_neighbors() { edges[0] = 0; }
  template <typename F>
  void sort(F&& less) { std::sort(edges + 1, edges + 1 + edges[0], less); }
  indexType* begin() { return edges + 1; }
  indexType* end() { return edges + 1 + edges[0]; }

 private:
  indexType* edges;
  long maxDeg;
  indexType id_;
};

template <typename indexType_>
struct Graph {
  using indexType = indexType_;
  long max_degree() const { return maxDeg; }
  size_t size() const { return n; }
  indexType* data() { return graph.get(); }
  const indexType* data() const { return graph.get(); }

  Graph() {}
  Graph(long maxDeg, size_t n) : n(n), maxDeg(maxDeg) { allocate_graph(maxDeg, n); }

  explicit Graph(const char* gFile) {     // :147-204
    std::ifstream reader(gFile, std::ios::binary);
    if (!reader.is_open()) { std::cout << "graph file " << gFile << " not found" << std::endl; abort(); }
    indexType num_points, max_deg;
    reader.read((char*)&num_points, sizeof(indexType));
    reader.read((char*)&max_deg, sizeof(indexType));
    n = num_points; maxDeg = max_deg;
    std::cout << "Graph: detected " << num_points << " points with max degree " << max_deg << std::endl;
    std::vector<indexType> degrees(n);
    reader.read((char*)degrees.data(), (std::streamsize)(sizeof(indexType) * n));
    allocate_graph(maxDeg, n);
    size_t total = 0;
    std::vector<indexType> buf;
    for (size_t i = 0; i < n; i++) {
      indexType* row = graph.get() + i * (maxDeg + 1);
      row[0] = degrees[i];
      reader.read((char*)(row + 1), (std::streamsize)(sizeof(indexType) * degrees[i]));
      total += degrees[i];
    }
    std::cout << "Total edges read from file: " << total << std::endl;
  }

  void save(const char* oFile) {         // :206-232
    std::cout << "Writing graph with " << n << " points and max degree " << maxDeg << std::endl;
    std::ofstream writer(oFile, std::ios::binary | std::ios::out);
    indexType pre[2] = {(indexType)n, (indexType)maxDeg};
    writer.write((char*)pre, 2 * sizeof(indexType));
    std::vector<indexType> sizes(n);
    for (size_t i = 0; i < n; i++) sizes[i] = graph.get()[i * (maxDeg + 1)];
    writer.write((char*)sizes.data(), (std::streamsize)(n * sizeof(indexType)));
    for (size_t i = 0; i < n; i++)
      writer.write((char*)(graph.get() + i * (maxDeg + 1) + 1), (std::streamsize)(sizes[i] * sizeof(indexType)));
    writer.close();
  }

  edgeRange<indexType> operator[](indexType i) const {
    if (i > n) { std::cout << "ERROR: graph index out of range: " << i << std::endl; abort(); }
    return edgeRange<indexType>(graph.get() + (size_t)i * (maxDeg + 1), graph.get() + ((size_t)i + 1) * (maxDeg + 1), i);
  }

 private:
  void allocate_graph(long maxDeg_, size_t n_) {
    const size_t cnt = n_ * (size_t)(maxDeg_ + 1);
    const size_t bytes = std::max<size_t>((cnt * sizeof(indexType) + ((1ul << 21) - 1)) & ~((1ul << 21) - 1), 1ul << 21);
    indexType* ptr = (indexType*)aligned_alloc(1l << 21, bytes);
    std::memset(ptr, 0, bytes);
    graph = std::shared_ptr<indexType[]>(ptr, std::free);
  }
  size_t n = 0;
  long maxDeg = 0;
  std::shared_ptr<indexType[]> graph;
};

}  // namespace parlayANN
