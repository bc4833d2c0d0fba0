// parlay_compat.h -- the two parlaylib names that appear in the SIGNATURES of the reference's hot-path surface
// (parlay::sequence in beamSearch.h:217-223,353-387,537-548; vamana/index.h:63-65,124-126,188-192).  When
// parlaylib is on the include path (a maintainer's tree) its own sequence is used; in this image parlaylib is
// absent (an un-vendored submodule), so a std::vector alias stands in.  Nothing here restates parlaylib code.
#pragma once
#if __has_include(<parlay/sequence.h>)
#include <parlay/sequence.h>
#else
#include <vector>
namespace parlay {
template <typename T>
using sequence = std::vector<T>;
}  // namespace parlay
#endif
