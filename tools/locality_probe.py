#!/usr/bin/env python3
"""Experiment (round 3): does the ORDER of the queries inside one launch matter?  The searches of one batch_insert batch are
independent (vamana/index.h:247-270: snapshot semantics), so the launch may take them in any order; queries that run side by
side on one XCD and are close in space share candidate rows in that XCD's L2.  Orders compared on a C3-shaped table:
random (= the batch order), sorted by a locality key, sorted + contiguous ranges per XCD (block b -> sorted[(b%8)*m/8 + b/8])."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parlayann_amd import DeviceIndex, datasets, _capi  # noqa: E402
from parlayann_amd._capi import QueryParams, SearchOut, check  # noqa: E402

n = int(os.environ.get("N", 2_000_000))
m = int(os.environ.get("M", 40_000))
shape = os.environ.get("SHAPE", "c3")
beam = int(os.environ.get("BEAM", 128))
dev = torch.device("cuda", 0)
if shape == "c3":
    X = datasets.deep_like(n, 96, seed=1234)
else:
    X = datasets.sift_like_device(n, 128, 1234, dev, np.float16, noise_scale=10.0)
ix = DeviceIndex(X, max_degree=64)
t0 = time.time(); ix.vamana_build(64, 128, 1.05 if shape == "c3" else 1.15, num_passes=2, seed=1); print(f"build {time.time()-t0:.2f}s", flush=True)
lib = _capi.load()
rng = np.random.default_rng(5)
batch = rng.choice(n, m, replace=False).astype(np.uint32)

# locality key: nearest of 4096 pivots, pivots grouped by their nearest of 64 top pivots
NP = int(os.environ.get("PIV", 4096))
piv = rng.choice(n, NP, replace=False)
pix = DeviceIndex(X[piv], max_degree=4)
near, _ = pix.bruteforce_knn(X[batch], 1)
top = DeviceIndex(X[piv[:64]], max_degree=4)
ptop, _ = top.bruteforce_knn(X[piv], 1)
key = ptop[near[:, 0], 0].astype(np.int64) * NP + near[:, 0]
srt = batch[np.argsort(key, kind="stable")]
cpx = m // 8
b = np.arange(m)
swz = srt[np.minimum((b % 8) * cpx + b // 8, m - 1)] if m % 8 == 0 else srt


def run(ids, label):
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    d_starts = torch.zeros(1, dtype=torch.int32, device=dev)
    d_vis = torch.empty(m, dtype=torch.int32, device=dev); d_cmps = torch.empty(m, dtype=torch.int32, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    k = 0 if beam > 64 else 10
    d_out = torch.empty((m, max(k, 1)), dtype=torch.int32, device=dev)
    qp = QueryParams(k=k, beam=beam, cut=1.35, limit=n, degree_limit=64, rerank_factor=100, pad=1.0)
    out = SearchOut(ids=d_out.data_ptr() if k else None, out_k=k, visited_count=d_vis.data_ptr(), dist_cmps=d_cmps.data_ptr(), status=d_status.data_ptr())
    st = torch.cuda.current_stream(dev)
    def go():
        check(lib.pann_batch_search_dev(ix.handle, None, d_ids.data_ptr(), m, 0, d_starts.data_ptr(), 1, C.byref(qp), C.byref(out), C.c_void_p(st.cuda_stream)))
    go(); go(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(5): go()
    e.record(st); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 5
    cm = int(d_cmps.cpu().numpy().astype(np.int64).sum())
    rowb = X.shape[1] * X.itemsize
    print(json.dumps({"order": label, "ms": ms, "cmps": cm, "alg_TBps": cm * rowb / (ms / 1e3) / 1e12, "status": int(d_status.item())}), flush=True)


for rep in range(2):
    run(batch, "random"); run(srt, "sorted"); run(swz, "sorted+xcd")
