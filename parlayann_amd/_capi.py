"""ctypes binding of the C-ABI in include/pann.h (libpann.so, built in-tree by csrc/Makefile).

The library is the product's only compute path: if it is missing, or no HIP device is visible,
callers get an exception -- there is no CPU fallback (DESIGN.md "No fallback").
"""
import ctypes as C
import os

# torch bundles its own libamdhip64.so; importing it first makes the dynamic loader resolve
# libpann.so's NEEDED libamdhip64.so.7 to that same copy, so tensors allocated by torch and kernels
# launched by libpann.so share ONE HIP runtime per process.
try:  # pragma: no cover - ordering shim only
    import torch  # noqa: F401
except Exception:  # torch is plumbing, not a requirement of the C-ABI
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpann.so")
if os.environ.get("PANN_LIBRARY"):      # A/B runs of diagnostic builds (tools/): same ABI, another file
    LIB_PATH = os.environ["PANN_LIBRARY"]

PANN_U8, PANN_I8, PANN_F32, PANN_F16, PANN_BF16 = 0, 1, 2, 3, 4
PANN_L2, PANN_MIPS = 0, 1
PANN_OK = 0
PANN_ERR_OVERFLOW = 5
PANN_ABI_VERSION = 3
PANN_STATUS_VISITED_OVERFLOW, PANN_STATUS_DROPPED_OVERFLOW = 1, 2

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


class QueryParams(C.Structure):
    """pann_query_params == QueryParams (algorithms/utils/types.h:218-231)."""
    _fields_ = [("k", C.c_int64), ("beam", C.c_int64), ("cut", C.c_double), ("limit", C.c_int64),
                ("degree_limit", C.c_int64), ("rerank_factor", C.c_int32), ("pad", C.c_float)]


class SearchOut(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("dists", C.c_void_p), ("out_k", C.c_uint32),
                ("frontier_size", C.c_void_p), ("visited_count", C.c_void_p),
                ("dist_cmps", C.c_void_p), ("degree_sum", C.c_void_p),
                ("visited_ids", C.c_void_p), ("visited_dists", C.c_void_p),
                ("visited_cap", C.c_uint32), ("status", C.c_void_p)]


class BuildStats(C.Structure):
    _fields_ = [("t_search_s", C.c_double), ("t_prune_s", C.c_double), ("t_bidirect_s", C.c_double),
                ("t_reprune_s", C.c_double), ("search_dist_cmps", C.c_uint64),
                ("prune_dist_cmps", C.c_uint64), ("visited_total", C.c_uint64),
                ("per_point_visited", C.c_void_p), ("per_point_dist_cmps", C.c_void_p)]


# every symbol include/pann.h declares: (restype, argtypes)
SIGNATURES = {
    "pann_abi_version": (C.c_int, []),
    "pann_last_error": (C.c_char_p, []),
    "pann_device_count": (C.c_int, []),
    "pann_index_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint32, C.c_int,
                                    C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.c_int]),
    "pann_index_create_empty": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int]),
    "pann_index_upload_points": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]),
    "pann_index_destroy": (None, [C.c_void_p]),
    "pann_index_size": (C.c_uint64, [C.c_void_p]),
    "pann_index_dims": (C.c_uint32, [C.c_void_p]),
    "pann_index_max_degree": (C.c_uint32, [C.c_void_p]),
    "pann_index_device": (C.c_int, [C.c_void_p]),
    "pann_index_set_exact_float_order": (C.c_int, [C.c_void_p, C.c_int]),
    "pann_range_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32,
                                    C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pann_index_reserve_dropped": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pann_index_dropped_capacity": (C.c_uint32, [C.c_void_p]),
    "pann_index_set_graph": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pann_index_update_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pann_index_get_graph": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pann_index_clear_graph": (C.c_int, [C.c_void_p]),
    "pann_index_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pann_index_set_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "pann_index_get_option": (C.c_int64, [C.c_void_p, C.c_char_p]),
    "pann_batch_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.c_uint32, C.POINTER(QueryParams), C.POINTER(SearchOut)]),
    "pann_batch_search_per_query_starts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                     C.c_void_p, C.c_uint32, C.POINTER(QueryParams), C.POINTER(SearchOut)]),
    "pann_batch_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                        C.c_uint32, C.POINTER(QueryParams), C.POINTER(SearchOut), C.c_void_p]),
    "pann_pair_distances": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pann_query_distances": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                       C.c_void_p]),
    "pann_robust_prune_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "pann_vamana_insert_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_double, C.POINTER(BuildStats)]),
    "pann_vamana_build_single_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_uint32, C.c_uint64,
                                                 C.c_int, C.c_void_p]),
    "pann_vamana_build": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_uint64, C.c_int,
                                    C.POINTER(BuildStats)]),
    "pann_vamana_search_prune_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_double, C.c_void_p, C.POINTER(BuildStats)]),
    "pann_vamana_apply_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_double,
                                             C.POINTER(BuildStats)]),
    "pann_vamana_sort_neighbors": (C.c_int, [C.c_void_p]),
    "pann_build_permutation": (None, [C.c_uint64, C.c_uint64, C.c_void_p]),
    "pann_vamana_batch_schedule": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]),
    "pann_leaf_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pann_leaf_knn_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                      C.c_void_p]),
    "pann_rerank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                              C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "pann_pivot_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pann_hcnng_build": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "pann_merge_topk_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "pann_hcnng_build_trees_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                             C.c_void_p, C.c_uint32, C.c_void_p]),
    "pann_hcnng_assemble_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pann_bruteforce_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                      C.c_void_p]),
}

_lib = None


class PannError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pann error {code}: {msg}")
        self.code = code


def load():
    """Load libpann.so and attach signatures.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C parlayann_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != PANN_OK:
        raise PannError(rc, load().pann_last_error().decode("utf-8", "replace"))
