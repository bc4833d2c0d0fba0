"""Python mirror of python/wrapper.py: the same five entry points with the same argument orders,
string-dispatched on (metric, dtype)."""
import ctypes as C
import os

import numpy as np

from . import _capi, io, quantize
from .graph_index import (FloatEuclidianIndex, FloatMipsIndex, Int8EuclidianIndex, Int8MipsIndex,
                          UInt8EuclidianIndex, UInt8MipsIndex)
from .index import DeviceIndex

_DT = {"uint8": np.uint8, "int8": np.int8, "float": np.float32}


def _check(metric, dtype):
    if metric not in ("Euclidian", "mips"):
        raise Exception("Invalid metric " + str(metric))
    if dtype not in _DT:
        raise Exception("Invalid data type " + str(dtype))


def build_vamana_index(metric, dtype, data_dir, index_dir, R, L, alpha, two_pass, seed=1, device=0):
    """python/builder.cpp:36-95 build_vamana_index: load points, (MIPS: normalise and adjust alpha
    :45-55), build with BuildParams(R, L, alpha, two_pass ? 2 : 1), save the graph."""
    _check(metric, dtype)
    X = io.read_bin(data_dir, _DT[dtype])
    if metric == "mips":
        print("normalizing")
        if dtype == "float":
            X = quantize.normalize_rows(X)
        if X.shape[1] <= 200:
            alpha = 1.0 if X.shape[1] < 100 else .98
    ix = DeviceIndex(X, max_degree=R, metric=metric, device=device)
    ix.vamana_build(R, L, alpha, num_passes=2 if two_pass else 1, seed=seed, sort_neighbors=True)
    io.write_graph(index_dir, ix.get_graph())
    ix.close()


def build_hcnng_index(metric, dtype, data_dir, index_dir, mst_deg, num_clusters, cluster_size, seed=1, device=0):
    """python/builder.cpp:114-140: BuildParams(num_clusters, cluster_size, mst_deg), hcnng_index::build_index."""
    _check(metric, dtype)
    X = io.read_bin(data_dir, _DT[dtype])
    G = hcnng_build(X, metric, num_clusters, cluster_size, mst_deg, seed=seed, device=device)
    io.write_graph(index_dir, G)


def build_pynndescent_index(*a, **k):
    raise NotImplementedError("pyNNDescent is out of scope (SURVEY.md section 2 #16)")


def build_hnsw_index(*a, **k):
    raise NotImplementedError("HNSW is out of scope (SURVEY.md section 2 #17)")


def load_index(metric, dtype, data_dir, index_dir, hnsw=False):
    _check(metric, dtype)
    cls = {("Euclidian", "uint8"): UInt8EuclidianIndex, ("Euclidian", "int8"): Int8EuclidianIndex,
           ("Euclidian", "float"): FloatEuclidianIndex, ("mips", "uint8"): UInt8MipsIndex,
           ("mips", "int8"): Int8MipsIndex, ("mips", "float"): FloatMipsIndex}[(metric, dtype)]
    return cls(data_dir, index_dir, hnsw)


# ---- HCNNG through the C++ host mirror (parlayann_amd/host/hcnng_index.h), compiled into
# lib/libpann_host.so by parlayann_amd/host/Makefile ----
_host = None


def _host_lib():
    global _host
    if _host is None:
        _capi.load()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpann_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `make -C parlayann_amd/host`")
        _host = C.CDLL(path)
        _host.pann_host_hcnng_build.restype = C.c_int
    return _host


def hcnng_build(X, metric, num_clusters, cluster_size, mst_deg, seed=1, device=0, host_mirror=False):
    """HCNNG graph of X.  Default: the whole build on the device (pann_hcnng_build).  host_mirror=True runs the
    C++ host mirror instead (host tree + Kruskal around pann_pivot_split / pann_leaf_knn_batch); both follow
    the same seeding rules and give the same graph."""
    X = np.ascontiguousarray(X)
    if not host_mirror:
        ix = DeviceIndex(X, max_degree=num_clusters * mst_deg, metric=metric, device=device)
        hcnng_build.last_times = ix.hcnng_build(num_clusters, cluster_size, mst_deg, seed)
        G = ix.get_graph()
        ix.close()
        return G
    n, d = X.shape
    maxdeg = num_clusters * mst_deg                      # BuildParams::max_degree (types.h:210-214)
    G = np.zeros((n, maxdeg + 1), np.uint32)
    times = np.zeros(3, np.float64)
    from .index import _DT as DTC, _metric_code
    rc = _host_lib().pann_host_hcnng_build(X.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(d),
                                           C.c_int(DTC[X.dtype]), C.c_int(_metric_code(metric)), C.c_long(num_clusters),
                                           C.c_long(cluster_size), C.c_long(mst_deg), C.c_uint64(seed), C.c_int(device),
                                           G.ctypes.data_as(C.c_void_p), times.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("hcnng build failed")
    hcnng_build.last_times = {"tree_s": times[0], "leaf_knn_s": times[1], "mst_s": times[2]}
    return G
