#!/usr/bin/env python3
"""Experiment (round 3): how much of the 10 K-query launch's tail is the ORDER in which the queries are dispatched?
The bench workload (1M x 128 fp16, R = 64, beam 64, k = 10, 10 000 queries); the query rows are permuted on the host, so every
order runs the same kernel on the same queries:
  batch        the order of the batch (what bench.py times)
  longest      descending visited count of a previous run -- an oracle no one-shot batch has: the upper bound of any predictor
  shortest     ascending (the worst case)
  far / near   descending / ascending distance from the query to the start point (a predictor that costs one distance per query)
usage: tail_probe.py [n=1000000] [nq=10000]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parlayann_amd import DeviceIndex, datasets, _capi  # noqa: E402
from parlayann_amd._capi import QueryParams, SearchOut, check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dev = torch.device("cuda", 0)
X = datasets.sift1m_like(n, 128, seed=1234, dtype=np.float32).astype(np.float16)
Q = datasets.sift1m_like(nq, 128, seed=4321, dtype=np.float16)
ix = DeviceIndex(X, max_degree=64)
ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1, sort_neighbors=True)
lib = _capi.load()
k, beam = 10, 64
stream = torch.cuda.current_stream(dev)
d_starts = torch.zeros(1, dtype=torch.int32, device=dev)
d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_vis = torch.empty(nq, dtype=torch.int32, device=dev)
d_cmps = torch.empty(nq, dtype=torch.int32, device=dev)
d_status = torch.zeros(1, dtype=torch.int32, device=dev)
qp = QueryParams(k=k, beam=beam, cut=1.35, limit=n, degree_limit=64, rerank_factor=100, pad=1.0)
out = SearchOut(ids=d_ids.data_ptr(), dists=None, out_k=k, frontier_size=None, visited_count=d_vis.data_ptr(),
                dist_cmps=d_cmps.data_ptr(), degree_sum=None, visited_ids=None, visited_dists=None, visited_cap=0,
                status=d_status.data_ptr())


def run(Qp, reps=20):
    d_q = torch.from_numpy(Qp.view(np.uint8).reshape(nq, -1)).to(dev)

    def go():
        check(lib.pann_batch_search_dev(ix.handle, d_q.data_ptr(), None, nq, 256, d_starts.data_ptr(), 1, C.byref(qp), C.byref(out),
                                        C.c_void_p(stream.cuda_stream)))
    for _ in range(3):
        go()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(stream); go(); b.record(stream)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])), d_vis.cpu().numpy().astype(np.int64), d_cmps.cpu().numpy().astype(np.int64)


ms0, vis, cmps = run(Q)
dstart = ((Q.astype(np.float32) - X[0].astype(np.float32)) ** 2).sum(1)
print(f"visited: mean {vis.mean():.1f} min {vis.min()} max {vis.max()}; cmps mean {cmps.mean():.0f} max {cmps.max()}; "
      f"corr(cmps, dist to start) {np.corrcoef(cmps, dstart)[0, 1]:.3f}", flush=True)
# candidate predictors of a query's length, all from the query and a sample of the base
rng = np.random.default_rng(7)
P = X[rng.choice(n, 4096, replace=False)].astype(np.float32)
Qf = Q.astype(np.float32)
D = (Qf ** 2).sum(1)[:, None] + (P ** 2).sum(1)[None, :] - 2.0 * Qf @ P.T        # nq x 4096
Ds = np.sort(D, axis=1)
r0 = np.median(Ds[:, 0])
preds = {"nn_pivot_1024": -np.sort(D[:, :1024], axis=1)[:, 0], "nn_pivot_4096": -Ds[:, 0], "mean10_4096": -Ds[:, :10].mean(1),
         "ratio_1_10": Ds[:, 0] / Ds[:, 9], "count_r": (D < 1.5 * r0).sum(1).astype(np.float64), "norm": (Qf ** 2).sum(1),
         "gap_50_1": -(Ds[:, 49] - Ds[:, 0])}


def spearman(a, b):
    ra = np.argsort(np.argsort(a)); rb = np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1])


for name, v in preds.items():
    print(f"predictor {name:14s} spearman with cmps {spearman(v, cmps):+.3f}", flush=True)
best = max(preds, key=lambda k_: abs(spearman(preds[k_], cmps)))
sgn = 1.0 if spearman(preds[best], cmps) > 0 else -1.0
print("best predictor:", best)
orders = {"batch": np.arange(nq), "predicted": np.argsort(-sgn * preds[best], kind="stable"), "longest": np.argsort(-cmps, kind="stable"), "shortest": np.argsort(cmps, kind="stable"),
          "far": np.argsort(-dstart, kind="stable"), "near": np.argsort(dstart, kind="stable")}
for rep in range(2):
    for name, perm in orders.items():
        ms, v, c = run(np.ascontiguousarray(Q[perm]))
        assert c.sum() == cmps.sum()
        print(f"{name:9s} {ms:.4f} ms  ({nq / ms / 1e3:.2f} M QPS)", flush=True)
