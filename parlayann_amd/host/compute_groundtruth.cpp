// compute_groundtruth.cpp -- the tool of data_tools/compute_groundtruth.cpp (flags :96-121) over the C-ABI: exact
// k nearest base points of every query by brute force on the device (pann_bruteforce_knn), written in the .ibin
// layout of :64-95 ([nq:i32][k:i32][nq*k ids:i32][nq*k distances:f32]).  Neighbours are ordered by (distance, id).
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "device_index.h"

using namespace parlayANN;

struct Args {
  std::map<std::string, std::string> kv;
  Args(int argc, char** argv) { for (int i = 1; i + 1 < argc; i += 2) kv[argv[i]] = argv[i + 1]; }
  const char* str(const char* k) const { auto it = kv.find(k); return it == kv.end() ? nullptr : it->second.c_str(); }
  long num(const char* k, long d) const { auto s = str(k); return s ? atol(s) : d; }
};

template <class Point>
int run(const Args& a, long k) {
  using PR = PointRange<Point>;
  PR Base(a.str("-base_path")), Queries(a.str("-query_path"));
  if (Base.dimension() != Queries.dimension()) { std::cout << "Error: base and query dimensions differ" << std::endl; abort(); }
  DeviceIndex<PR, unsigned int> DI(Base, nullptr, 1, (int)a.num("-device", 0));
  const size_t nq = Queries.size();
  std::vector<uint32_t> ids(nq * (size_t)k);
  std::vector<float> dists(nq * (size_t)k);
  pann_check(pann_bruteforce_knn(DI.h, Queries.data(), nq, Queries.get_aligned_bytes(), (uint32_t)k, ids.data(), dists.data()));
  std::cout << "Writing file with dimension " << k << std::endl;
  std::cout << "File contains groundtruth for " << nq << " query points" << std::endl;
  std::ofstream w(a.str("-gt_path"), std::ios::binary | std::ios::out);
  if (!w.is_open()) { std::cout << "Error: cannot open " << a.str("-gt_path") << std::endl; abort(); }
  const int32_t hdr[2] = {(int32_t)nq, (int32_t)k};
  w.write((const char*)hdr, 8);
  w.write((const char*)ids.data(), (std::streamsize)(ids.size() * 4));
  w.write((const char*)dists.data(), (std::streamsize)(dists.size() * 4));
  return 0;
}

int main(int argc, char** argv) {
  Args a(argc, argv);
  if (!a.str("-base_path") || !a.str("-query_path") || !a.str("-gt_path") || !a.str("-data_type") || !a.str("-dist_func")) {
    std::cout << "usage: compute_groundtruth -base_path <b> -query_path <q> -data_type <d> -k <k> -dist_func <d> -gt_path <outfile>" << std::endl;
    return 1;
  }
  const std::string df = a.str("-dist_func"), tp = a.str("-data_type");
  if (df != "Euclidian" && df != "mips") { std::cout << "Error: invalid distance type: specify Euclidian or mips" << std::endl; abort(); }
  if (tp != "uint8" && tp != "int8" && tp != "float") { std::cout << "Error: data type not specified correctly, specify int8, uint8, or float" << std::endl; abort(); }
  const long k = a.num("-k", 100);
  std::cout << "Computing the " << k << " nearest neighbors" << std::endl;
  const bool mips = df == "mips";
  if (tp == "uint8") return mips ? run<Mips_Point<uint8_t>>(a, k) : run<Euclidian_Point<uint8_t>>(a, k);
  if (tp == "int8") return mips ? run<Mips_Point<int8_t>>(a, k) : run<Euclidian_Point<int8_t>>(a, k);
  return mips ? run<Mips_Point<float>>(a, k) : run<Euclidian_Point<float>>(a, k);
}
