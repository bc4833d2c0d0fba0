"""On-disk formats of the reference:
  vectors  .bin  : [n:u32][d:u32][n*d*sizeof(T)]                 (point_range.h:74-117, python/_files.py:42-95)
  graph          : [n:u32][maxDeg:u32][deg[n]:u32][edges...:u32]  (graph.h:147-232)
  truth    .ibin : [n:i32][k:i32][ids n*k:u32][dists n*k:f32]     (types.h:48-73, compute_groundtruth.cpp:63-102)
"""
import numpy as np


def write_bin(path, x):
    x = np.ascontiguousarray(x)
    with open(path, "wb") as f:
        np.array(x.shape, dtype=np.uint32).tofile(f)
        x.tofile(f)


def read_bin(path, dtype):
    with open(path, "rb") as f:
        n, d = np.fromfile(f, dtype=np.uint32, count=2)
        return np.fromfile(f, dtype=dtype, count=int(n) * int(d)).reshape(int(n), int(d))


def write_graph(path, graph):
    """graph: n x (maxDeg+1) uint32 in the in-memory reference layout (slot 0 = degree)."""
    g = np.ascontiguousarray(graph, dtype=np.uint32)
    n, w = g.shape
    deg = g[:, 0]
    with open(path, "wb") as f:
        np.array([n, w - 1], dtype=np.uint32).tofile(f)
        deg.tofile(f)
        mask = np.arange(w - 1)[None, :] < deg[:, None]
        g[:, 1:][mask].tofile(f)


def read_graph(path):
    with open(path, "rb") as f:
        n, maxdeg = (int(v) for v in np.fromfile(f, dtype=np.uint32, count=2))
        deg = np.fromfile(f, dtype=np.uint32, count=n)
        edges = np.fromfile(f, dtype=np.uint32, count=int(deg.sum()))
    g = np.zeros((n, maxdeg + 1), dtype=np.uint32)
    g[:, 0] = deg
    mask = np.arange(maxdeg)[None, :] < deg[:, None]
    g[:, 1:][mask] = edges
    return g


def write_ibin(path, ids, dists):
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    with open(path, "wb") as f:
        np.array(ids.shape, dtype=np.int32).tofile(f)
        ids.tofile(f)
        dists.tofile(f)


def read_ibin(path):
    with open(path, "rb") as f:
        n, k = (int(v) for v in np.fromfile(f, dtype=np.int32, count=2))
        ids = np.fromfile(f, dtype=np.uint32, count=n * k).reshape(n, k)
        dists = np.fromfile(f, dtype=np.float32, count=n * k).reshape(n, k)
    return ids, dists
