// types.h -- host mirror of algorithms/utils/types.h (QueryParams :218-231, BuildParams :154-215,
// groundTruth :38-107).  Same field names, constructor argument orders and defaults.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace parlayANN {

struct BuildParams {
  long R = 0;              // vamana
  long L = 0;              // vamana
  double m_l = 0;          // HNSW (unused here)
  double alpha = 0;        // vamana
  int num_passes = 1;      // vamana
  long num_clusters = 0;   // HCNNG
  long cluster_size = 0;   // HCNNG
  long MST_deg = 0;        // HCNNG
  double delta = 0;
  bool verbose = false;
  int quantize = 0;
  double radius = 0, radius_2 = 0;
  bool self = false, range = false;
  int single_batch = 0;
  long Q = 0;
  double trim = 0.0;
  double rerank_factor = 100;
  std::string alg_type;
  // this build's additions (no counterpart upstream): the seed that replaces parlay::random_permutation /
  // std::random_device (DESIGN.md "Build determinism"), the out-neighbour seeding of the self range search
  // (beamSearch.h:260-262, commented upstream), the host-orchestrated HCNNG cross-check path
  uint64_t seed = 1;
  bool use_existing = false;
  bool host_tree = false;

  BuildParams() {}
  // types.h:181-190
  BuildParams(long R, long L, double a, int num_passes, long nc, long cs, long mst, double de, bool verbose = false,
              int quantize = 0, double radius = 0.0, double radius_2 = 0.0, bool self = false, bool range = false,
              int single_batch = 0, long Q = 0, double trim = 0.0, int rerank_factor = 100)
      : R(R), L(L), alpha(a), num_passes(num_passes), num_clusters(nc), cluster_size(cs), MST_deg(mst), delta(de),
        verbose(verbose), quantize(quantize), radius(radius), radius_2(radius_2), self(self), range(range),
        single_batch(single_batch), Q(Q), trim(trim), rerank_factor(rerank_factor) {
    if (R != 0 && L != 0 && alpha != 0) alg_type = "Vamana";
    else if (num_clusters != 0 && cluster_size != 0 && MST_deg != 0) alg_type = "HCNNG";
  }
  BuildParams(long R, long L, double a, int num_passes, bool verbose = false)   // :194-196
      : R(R), L(L), alpha(a), num_passes(num_passes), verbose(verbose), single_batch(0) { alg_type = "Vamana"; }
  BuildParams(long nc, long cs, long mst)                                       // :202-204
      : num_clusters(nc), cluster_size(cs), MST_deg(mst), verbose(false) { alg_type = "HCNNG"; }

  long max_degree() const {                                                     // :210-214
    if (alg_type == "HCNNG") return num_clusters * MST_deg;
    return R;
  }
};

struct QueryParams {                                                            // :218-231
  long k = 0;
  long beamSize = 0;
  double cut = 0;
  long limit = 0;
  long degree_limit = 0;
  int rerank_factor = 100;
  float pad = 1.0;
  QueryParams(long k, long Q, double cut, long limit, long dg, double rerank_factor = 100)
      : k(k), beamSize(Q), cut(cut), limit(limit), degree_limit(dg), rerank_factor((int)rerank_factor) {}
  QueryParams() {}
};

// ground truth file: [n:i32][k:i32][ids n*k u32][dists n*k f32]   (types.h:48-73)
template <typename T>
struct groundTruth {
  long n = 0, dim = 0;
  std::vector<T> ids;
  std::vector<float> dists;
  groundTruth() {}
  explicit groundTruth(const char* gtFile) {
    if (gtFile == nullptr) return;
    std::ifstream in(gtFile, std::ios::binary);
    if (!in.is_open()) { std::cout << "ground truth file " << gtFile << " not found" << std::endl; abort(); }
    int32_t hdr[2];
    in.read((char*)hdr, 8);
    n = hdr[0]; dim = hdr[1];
    ids.resize((size_t)n * dim); dists.resize((size_t)n * dim);
    in.read((char*)ids.data(), (std::streamsize)(ids.size() * sizeof(T)));
    in.read((char*)dists.data(), (std::streamsize)(dists.size() * 4));
  }
  T coordinates(long i, long j) const { return ids[(size_t)i * dim + j]; }
  float distances(long i, long j) const { return dists[(size_t)i * dim + j]; }
  size_t size() const { return (size_t)n; }
  long dimension() const { return dim; }
};

}  // namespace parlayANN
