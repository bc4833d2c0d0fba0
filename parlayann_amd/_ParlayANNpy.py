"""Drop-in for the reference's pybind module `_ParlayANNpy` (python/module.cpp:132-177): the same 6
index classes, the 24 `build_{vamana,hcnng,pynndescent,hnsw}_{float,uint8,int8}_{euclidian,mips}_index`
functions with the reference's argument orders, and the `defaults` submodule.  `python/wrapper.py`'s
`from _ParlayANNpy import *` works unchanged against this module (pyNNDescent / HNSW builders raise:
out of scope)."""
import types

from . import wrapper as _w
from .graph_index import (FloatEuclidianIndex, FloatMipsIndex, Int8EuclidianIndex, Int8MipsIndex,  # noqa: F401
                          UInt8EuclidianIndex, UInt8MipsIndex)

defaults = types.SimpleNamespace(METRIC="Euclidian", ALPHA=1.2, GRAPH_DEGREE=64, BEAMWIDTH=128)   # module.cpp:142-148

__all__ = ["FloatEuclidianIndex", "FloatMipsIndex", "UInt8EuclidianIndex", "UInt8MipsIndex", "Int8EuclidianIndex",
           "Int8MipsIndex", "defaults"]


def _mk_vamana(dtype, metric):
    def f(distance_metric, vector_bin_path, index_output_path, graph_degree, beam_width, alpha, two_pass):
        # builder.cpp:36-95; `distance_metric` is passed through as in the reference (module.cpp:62-63)
        return _w.build_vamana_index(metric, dtype, vector_bin_path, index_output_path, graph_degree, beam_width, alpha, two_pass)
    return f


def _mk_hcnng(dtype, metric):
    def f(distance_metric, vector_bin_path, index_output_path, mst_deg, num_clusters, cluster_size):
        return _w.build_hcnng_index(metric, dtype, vector_bin_path, index_output_path, mst_deg, num_clusters, cluster_size)
    return f


def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name}: out of scope of the MI355X hot path (SURVEY.md section 2)")
    return f


for _dt in ("float", "uint8", "int8"):
    for _mname, _metric in (("euclidian", "Euclidian"), ("mips", "mips")):
        for _alg, _mk in (("vamana", _mk_vamana), ("hcnng", _mk_hcnng)):
            _n = f"build_{_alg}_{_dt}_{_mname}_index"
            globals()[_n] = _mk(_dt, _metric); __all__.append(_n)
        for _alg in ("pynndescent", "hnsw"):
            _n = f"build_{_alg}_{_dt}_{_mname}_index"
            globals()[_n] = _unsupported(_n); __all__.append(_n)
