// parse_results.h -- host mirror of the report types of algorithms/utils/parse_results.h: Graph_ (:35-58),
// nn_result (:110-160) and parse_result (:192-218: per recall bucket the fastest run), plus graph_stats_ (stats.h:47-55).
#pragma once
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "graph.h"
#include "parlay_compat.h"

namespace parlayANN {

struct Graph_ {
  std::string name, params; long size = 0; double avg_deg = 0; int max_deg = 0; double time = 0;
  Graph_() {}
  Graph_(std::string n, std::string p, long s, double ad, int md, double t) : name(n), params(p), size(s), avg_deg(ad), max_deg(md), time(t) {}
  void print() const {
    std::cout << name << " graph built with " << size << " points and parameters " << params << std::endl;
    std::cout << "Graph has average degree " << avg_deg << " and maximum degree " << max_deg << std::endl;
    std::cout << "Graph built in " << time << " seconds" << std::endl;
  }
};

template <typename indexType>
std::pair<double, int> graph_stats_(Graph<indexType>& G) {                   // stats.h:47-55
  size_t tot = 0, mx = 0;
  for (size_t i = 0; i < G.size(); i++) { const size_t d = G[(indexType)i].size(); tot += d; if (d > mx) mx = d; }
  return std::make_pair(G.size() ? (double)tot / (double)G.size() : 0.0, (int)mx);
}

struct nn_result {
  double recall = 0;
  unsigned avg_cmps = 0, tail_cmps = 0, avg_visited = 0, tail_visited = 0;
  float QPS = 0;
  int k = 0, beamQ = 0; float cut = 0; int limit = 0, degree_limit = 0, gtn = 0;
  long num_queries = 0;
  nn_result() {}
  // stats = {avg cmps, tail cmps, avg visited, tail visited}   (:130-146)
  nn_result(double r, const parlay::sequence<unsigned>& stats, float qps, int K, int Q, float c, long q, int limit, int degree_limit, int gtn)
      : recall(r), QPS(qps), k(K), beamQ(Q), cut(c), limit(limit), degree_limit(degree_limit), gtn(gtn), num_queries(q) {
    if (stats.size() != 4) abort();
    avg_cmps = stats[0]; tail_cmps = stats[1]; avg_visited = stats[2]; tail_visited = stats[3];
  }
  void print() const {                                                      // :148-153
    std::cout << "For " << gtn << "@" << gtn << " recall = " << recall << ", QPS = " << QPS << ", Q = " << beamQ << ", cut = " << cut;
    std::cout << ", visited limit = " << limit << ", degree limit: " << degree_limit;
    std::cout << ", average visited = " << avg_visited << ", average cmps = " << avg_cmps << std::endl;
  }
  void print_verbose() const {                                              // :155-164
    std::cout << "Over " << num_queries << " queries" << std::endl;
    std::cout << "k = " << k << ", Q = " << beamQ << ", cut = " << cut << ", throughput = " << QPS << "/second" << std::endl;
    std::cout << "Recall: " << recall << std::endl;
    std::cout << "Average dist cmps: " << avg_cmps << ", 99th percentile dist cmps: " << tail_cmps << std::endl;
    std::cout << "Average num visited: " << avg_visited << ", 99th percentile num visited: " << tail_visited << std::endl;
  }
};

// for bucket b_i the fastest result with b_i <= recall <= b_{i+1} (the last bucket, or one nothing reaches, is open
// above); prints one line per non-empty bucket
template <typename res>
std::pair<parlay::sequence<res>, parlay::sequence<float>> parse_result(const parlay::sequence<res>& results,
                                                                       const parlay::sequence<float>& buckets) {
  parlay::sequence<res> best; parlay::sequence<float> kept;
  for (size_t i = 0; i < buckets.size(); i++) {
    const res* top = nullptr;
    for (const res& r : results) {
      if (r.recall < buckets[i]) continue;
      if (i + 1 < buckets.size() && r.recall > buckets[i + 1]) continue;
      if (!top || top->QPS < r.QPS) top = &r;
    }
    if (top) { top->print(); best.push_back(*top); kept.push_back(buckets[i]); }
  }
  return {best, kept};
}

}  // namespace parlayANN
