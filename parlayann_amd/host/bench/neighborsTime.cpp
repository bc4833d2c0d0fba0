// neighborsTime.cpp -- the `neighbors` driver of algorithms/bench/neighborsTime.C, built once per algorithm directory
// like upstream (vamana/neighbors, HCNNG/neighbors: the Makefile selects the plugin's neighbors.h):
//   main (:73-252) parses the reference's flags, loads base / query / ground truth / graph, applies -normalize and
//   -quantize_bits 8, then timeNeighbors<Point, PointRange, indexType> (:50-70) -> ANN<...>(G, k, BP, Query_Points, GT,
//   res_file, graph_built, Points) -> G.save(-graph_outfile).
// Flags as upstream: -base_path -query_path -gt_path -graph_path -graph_outfile -res_path -data_type {uint8,int8,float}
//   -dist_func {Euclidian,mips} -k -Q -R -L -alpha -num_passes -two_pass -mst_deg -num_clusters -cluster_size -delta
//   -quantize_bits {0,8} -quantize_mode {0,1} -verbose -normalize -self -range -radius -radius_2 -rerank_factor
//   (-quantize_bits 16 and -quantize_mode 2..5 are rejected: out of scope).
// Added here: -device <ordinal>, -seed <s>, -use_existing (self range search seeded with out-neighbours),
//   -host_tree (HCNNG cross-check path).
#include <chrono>
#include <cstring>
#include <string>

#ifdef PANN_ALG_HCNNG
#include "../HCNNG/neighbors.h"
#else
#include "../vamana/neighbors.h"
#endif
#include "../quantize.h"
#include "parse_command_line.h"

using namespace parlayANN;
using uint = unsigned int;

template <typename Point, typename PointRange, typename indexType>
void timeNeighbors(Graph<indexType>& G, PointRange& Query_Points, long k, BuildParams& BP, char* outFile, groundTruth<indexType> GT,
                   char* res_file, bool graph_built, PointRange& Points) {
  const auto t0 = std::chrono::steady_clock::now();                  // time_loop(1, 0, ...) of bench/time_loop.h: one timed round
  ANN<Point, PointRange, indexType>(G, k, BP, Query_Points, GT, res_file, graph_built, Points);
  std::cout << "ANN: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
  if (outFile != NULL) G.save(outFile);
}

template <typename Point>
void run_plain(char* iFile, char* qFile, char* gFile, char* oFile, char* rFile, long maxDeg, long k, BuildParams& BP,
               groundTruth<uint>& GT, bool graph_built, bool normalize, int quantize) {
  using PR = PointRange<Point>;
  PR Points(iFile);
  PR Query_Points(qFile);
  Graph<unsigned int> G;
  if (gFile == NULL) G = Graph<unsigned int>(maxDeg, Points.size());
  else G = Graph<unsigned int>(gFile);
  if constexpr (std::is_same<typename Point::T, float>::value) {
    if (normalize) {                                                   // :147-153
      std::cout << "normalizing data" << std::endl;
      normalize_range(Points);
      normalize_range(Query_Points);
    }
    if (quantize == 8) {                                               // :157-164, :190-197
      std::cout << "quantizing data to 1 byte" << std::endl;
      if constexpr (Point::metric == PANN_L2) {
        using QPoint = Euclidian_Point<uint8_t>;
        using QPR = PointRange<QPoint>;
        const euclid_u8_parameters pm = generate_parameters_u8(Points);
        QPR Points_ = quantize_u8(Points, pm);
        QPR Query_Points_ = quantize_u8(Query_Points, pm);
        timeNeighbors<QPoint, QPR, uint>(G, Query_Points_, k, BP, oFile, GT, rFile, graph_built, Points_);
      } else {
        using QPoint = Mips_Point<int8_t>;                             // Quantized_Mips_Point<8>: trim = false (:193)
        using QPR = PointRange<QPoint>;
        const float mv = generate_max_val_mips_i8(Points, false);
        QPR Points_ = quantize_mips_i8(Points, mv);
        QPR Query_Points_ = quantize_mips_i8(Query_Points, mv);
        timeNeighbors<QPoint, QPR, uint>(G, Query_Points_, k, BP, oFile, GT, rFile, graph_built, Points_);
      }
      return;
    }
  }
  timeNeighbors<Point, PR, uint>(G, Query_Points, k, BP, oFile, GT, rFile, graph_built, Points);
}

// ---- command line --------------------------------------------------------------------------------------------------
// The flag NAMES, their defaults and their accepted ranges are the interface of the reference's `neighbors` binary
// (bench/neighborsTime.C; tests/test_host_cpp_gpu.py pins them).  They live in ONE table; parsing, range checks and the
// usage text are generated from it.
namespace {

enum class Kind { text, integer, real, toggle };

struct FlagSpec {
  const char* name;
  Kind kind;
  double fallback;        // value when the flag is absent (numeric kinds)
  double lo, hi;          // accepted closed range (numeric kinds); outside -> usage + exit, like commandLine::badArgument
  bool required;
  const char* what;
};

constexpr double kAny = 9.0e18;

const FlagSpec kFlags[] = {
    // files and types
    {"-base_path", Kind::text, 0, 0, 0, true, "base vectors (.bin: n, d, rows)"},
    {"-query_path", Kind::text, 0, 0, 0, false, "query vectors"},
    {"-gt_path", Kind::text, 0, 0, 0, false, "ground truth (n, k, ids, dists)"},
    {"-graph_path", Kind::text, 0, 0, 0, false, "prebuilt graph to search"},
    {"-graph_outfile", Kind::text, 0, 0, 0, false, "where to save the built graph"},
    {"-res_path", Kind::text, 0, 0, 0, false, "csv of the search sweep"},
    {"-data_type", Kind::text, 0, 0, 0, true, "uint8 | int8 | float"},
    {"-dist_func", Kind::text, 0, 0, 0, true, "Euclidian | mips"},
    // Vamana
    {"-R", Kind::integer, 0, 0, kAny, false, "max degree"},
    {"-L", Kind::integer, 0, 0, kAny, false, "build beam"},
    {"-alpha", Kind::real, 1.0, -kAny, kAny, false, "prune slack"},
    {"-num_passes", Kind::integer, 1, -kAny, kAny, false, "passes over the points"},
    {"-two_pass", Kind::integer, 0, 0, 1, false, "1 = two passes"},
    {"-single_batch", Kind::integer, 0, -kAny, kAny, false, "random start degree, one batch per pass"},
    // HCNNG
    {"-mst_deg", Kind::integer, 0, 0, kAny, false, "MST degree bound"},
    {"-num_clusters", Kind::integer, 0, 0, kAny, false, "cluster trees"},
    {"-cluster_size", Kind::integer, 0, 0, kAny, false, "leaf size"},
    // search
    {"-k", Kind::integer, 0, 0, 1000, false, "neighbours reported"},
    {"-Q", Kind::integer, 0, -kAny, kAny, false, "single beam instead of the sweep"},
    {"-rerank_factor", Kind::integer, 100, -kAny, kAny, false, "candidates re-scored per k"},
    {"-delta", Kind::real, 0, 0, kAny, false, ""},
    {"-radius", Kind::real, 0, -kAny, kAny, false, "range search radius"},
    {"-radius_2", Kind::real, 0, -kAny, kAny, false, "expansion radius (default: -radius)"},
    {"-trim", Kind::real, 0, -kAny, kAny, false, ""},
    // transforms and modes
    {"-quantize_bits", Kind::integer, 0, -kAny, kAny, false, "0 | 8"},
    {"-quantize_mode", Kind::integer, 0, -kAny, kAny, false, "0 | 1"},
    {"-verbose", Kind::toggle, 0, 0, 0, false, ""},
    {"-normalize", Kind::toggle, 0, 0, 0, false, ""},
    {"-self", Kind::toggle, 0, 0, 0, false, ""},
    {"-range", Kind::toggle, 0, 0, 0, false, ""},
    // this build's additions
    {"-device", Kind::integer, 0, 0, 1023, false, "GPU ordinal"},
    {"-seed", Kind::integer, 1, -kAny, kAny, false, "build seed"},
    {"-use_existing", Kind::toggle, 0, 0, 0, false, "self range search seeded with out-neighbours"},
    {"-host_tree", Kind::toggle, 0, 0, 0, false, "HCNNG cross-check path"},
};

std::string usage_text() {
  std::string u;
  for (const FlagSpec& f : kFlags) {
    const char* arg = f.kind == Kind::text ? " <s>" : f.kind == Kind::integer ? " <n>" : f.kind == Kind::real ? " <x>" : "";
    u += f.required ? std::string(f.name) + arg + " " : "[" + std::string(f.name) + arg + "] ";
  }
  return u;
}

// the parsed command line: every table row has a value
class Flags {
 public:
  explicit Flags(commandLine& P) {
    for (size_t i = 0; i < sizeof(kFlags) / sizeof(kFlags[0]); i++) {
      const FlagSpec& f = kFlags[i];
      Slot& v = slots_[i];
      switch (f.kind) {
        case Kind::text: v.s = P.getOptionValue(f.name); v.given = v.s != NULL; break;
        case Kind::toggle: v.given = P.getOption(f.name); v.x = v.given ? 1 : 0; break;
        case Kind::integer: v.given = P.getOptionValue(f.name) != NULL; v.x = (double)P.getOptionLongValue(f.name, (long)f.fallback); break;
        case Kind::real: v.given = P.getOptionValue(f.name) != NULL; v.x = P.getOptionDoubleValue(f.name, f.fallback); break;
      }
      const bool numeric = f.kind == Kind::integer || f.kind == Kind::real;
      if ((f.required && !v.given) || (numeric && (v.x < f.lo || v.x > f.hi))) P.badArgument();
    }
  }
  char* text(const char* name) const { return at(name).s; }
  long integer(const char* name) const { return (long)at(name).x; }
  double real(const char* name) const { return at(name).x; }
  bool on(const char* name) const { return at(name).x != 0; }
  bool given(const char* name) const { return at(name).given; }

 private:
  struct Slot { char* s = NULL; double x = 0; bool given = false; };
  Slot slots_[sizeof(kFlags) / sizeof(kFlags[0])];
  const Slot& at(const char* name) const {
    for (size_t i = 0; i < sizeof(kFlags) / sizeof(kFlags[0]); i++) if (std::strcmp(kFlags[i].name, name) == 0) return slots_[i];
    std::cout << "internal error: flag " << name << " is not in the table" << std::endl;
    abort();
  }
};

[[noreturn]] void refuse(const std::string& why) {
  std::cout << "Error: " << why << std::endl;
  abort();
}

template <typename T>
void run_typed(bool euclidian, const Flags& F, BuildParams& BP, groundTruth<uint>& GT) {
  const bool is_float = std::is_same<T, float>::value;
  const bool normalize = is_float && F.on("-normalize");
  const int qbits = is_float ? (int)F.integer("-quantize_bits") : 0;
  char* gFile = F.text("-graph_path");
  if (euclidian)
    run_plain<Euclidian_Point<T>>(F.text("-base_path"), F.text("-query_path"), gFile, F.text("-graph_outfile"), F.text("-res_path"),
                                  BP.max_degree(), F.integer("-k"), BP, GT, gFile != NULL, normalize, qbits);
  else
    run_plain<Mips_Point<T>>(F.text("-base_path"), F.text("-query_path"), gFile, F.text("-graph_outfile"), F.text("-res_path"),
                             BP.max_degree(), F.integer("-k"), BP, GT, gFile != NULL, normalize, qbits);
}

}  // namespace

int main(int argc, char* argv[]) {
  commandLine P(argc, argv, usage_text());
  const Flags F(P);

  const std::string elem = F.text("-data_type"), metric = F.text("-dist_func");
  if (elem != "uint8" && elem != "int8" && elem != "float") refuse("vector type not specified correctly, specify int8, uint8, or float");
  if (metric != "Euclidian" && metric != "mips") refuse("specify distance type Euclidian or mips");
  const long qbits = F.integer("-quantize_bits");
  if (qbits != 0 && !(qbits == 8 && elem == "float")) refuse("-quantize_bits supports 8 with -data_type float (16 is not mirrored)");

  // BuildParams in the reference's constructor order (types.h:181-190)
  const int passes = F.integer("-two_pass") == 1 ? 2 : (int)F.integer("-num_passes");
  const double radius = F.real("-radius");
  BuildParams BP(F.integer("-R"), F.integer("-L"), F.real("-alpha"), passes, F.integer("-num_clusters"), F.integer("-cluster_size"),
                 F.integer("-mst_deg"), F.real("-delta"), F.on("-verbose"), (int)F.integer("-quantize_mode"), radius,
                 F.given("-radius_2") ? F.real("-radius_2") : radius, F.on("-self"), F.on("-range"), (int)F.integer("-single_batch"),
                 F.integer("-Q"), F.real("-trim"), (int)F.integer("-rerank_factor"));
  BP.seed = (uint64_t)F.integer("-seed");
  BP.use_existing = F.on("-use_existing");
  BP.host_tree = F.on("-host_tree");
  set_default_device((int)F.integer("-device"));
#ifdef PANN_ALG_HCNNG
  if (BP.alg_type != "HCNNG") refuse("HCNNG needs -num_clusters, -cluster_size and -mst_deg");
#else
  if (BP.alg_type != "Vamana") refuse("Vamana needs -R, -L and -alpha");
#endif

  groundTruth<uint> GT(F.text("-gt_path"));
  const bool euclidian = metric == "Euclidian";
  if (elem == "float") run_typed<float>(euclidian, F, BP, GT);
  else if (elem == "uint8") run_typed<uint8_t>(euclidian, F, BP, GT);
  else run_typed<int8_t>(euclidian, F, BP, GT);
  return 0;
}
