#!/usr/bin/env python3
"""Run the BASELINE.json configurations that fit one GPU (C1, C3, C5) end to end on the device and
print one JSON line each (build time, recall, QPS, counters).  Not a bench contract: the numbers go
to DESIGN.md.  usage: run_configs.py [c1] [c3[:n]] [c5[:n]]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parlayann_amd import DeviceIndex, datasets, quantize, wrapper  # noqa: E402
from parlayann_amd.recall import recall_at_k  # noqa: E402


def search_stats(ix, Q, k, beam, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = ix.batch_search(Q, k=k, beam=beam, cut=1.35)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return r, len(Q) / best


def c1():
    X = datasets.sift_like(10_000, 128, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(1_000, 128, seed=4321, dtype=np.uint8)
    ix = DeviceIndex(X, max_degree=32)
    t0 = time.time(); st = ix.vamana_build(32, 64, 1.2, num_passes=1, seed=1); tb = time.time() - t0
    gt, gd = ix.bruteforce_knn(Q, 100)
    out = {"config": "C1 SIFT-10K u8 d=128 Vamana R=32 L=64 a=1.2, 1K queries", "build_s": tb}
    for beam in (10, 16, 32, 64, 128):
        r, qps = search_stats(ix, Q, 10, beam)
        out[f"beam{beam}"] = {"recall": recall_at_k(r["ids"], gt, gd, 10), "visited": float(r["visited_count"].mean()),
                              "cmps": float(r["dist_cmps"].mean()), "qps_host_inclusive": qps}
    print(json.dumps(out), flush=True)


def c3(n):
    t0 = time.time()
    X = datasets.deep_like(n, 96, seed=1234); Q = datasets.deep_like(10_000, 96, seed=4321)
    tg = time.time() - t0
    ix = DeviceIndex(X, max_degree=64)
    t0 = time.time(); st = ix.vamana_build(64, 128, 1.05, num_passes=2, seed=1); tb = time.time() - t0
    gt, gd = ix.bruteforce_knn(Q, 100)
    out = {"config": f"C3 DEEP-shaped {n}x96 f32 Vamana R=64 L=128 a=1.05 x2, 10K queries", "datagen_s": tg, "build_s": tb,
           "build_phases_s": {"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s, "reprune": st.t_reprune_s},
           "build_dist_cmps": {"search": st.search_dist_cmps, "prune": st.prune_dist_cmps},
           "avg_visited_per_insert": st.visited_total / (2.0 * n)}
    G = ix.get_graph()
    out["avg_degree"] = float(G[:, 0].mean()); out["max_degree"] = int(G[:, 0].max())
    for beam in (32, 64, 128):
        r, qps = search_stats(ix, Q, 10, beam)
        out[f"beam{beam}"] = {"recall": recall_at_k(r["ids"], gt, gd, 10), "visited": float(r["visited_count"].mean()),
                              "cmps": float(r["dist_cmps"].mean()), "qps_host_inclusive": qps}
    print(json.dumps(out), flush=True)


def c5(n):
    t0 = time.time()
    X = datasets.t2i_like(n, 200, seed=1234); Q = datasets.t2i_like(10_000, 200, seed=4321)
    mv = quantize.mips_i8_max_val(X, trim=False)          # Quantized_Mips_Point<8> = <8,false,255> (neighborsTime.C:193)
    Xq = quantize.mips_i8_translate(X, mv); Qq = quantize.mips_i8_translate(Q, mv)
    tg = time.time() - t0
    t0 = time.time()
    G = wrapper.hcnng_build(Xq, "mips", 30, 1000, 3, seed=1); tb = time.time() - t0
    ix = DeviceIndex(Xq, G, metric="mips")
    gt, gd = ix.bruteforce_knn(Qq, 100)
    out = {"config": f"C5 T2I-shaped {n}x200 f32->int8 MIPS, HCNNG 30 trees leaf 1000 mst_deg 3, 10K queries", "datagen_s": tg,
           "build_s": tb, "build_phases_s": wrapper.hcnng_build.last_times, "avg_degree": float(G[:, 0].mean()),
           "max_degree": int(G[:, 0].max())}
    for beam in (32, 64, 128):
        r, qps = search_stats(ix, Qq, 10, beam)
        out[f"beam{beam}"] = {"recall_vs_int8_gt": recall_at_k(r["ids"], gt, gd, 10), "visited": float(r["visited_count"].mean()),
                              "cmps": float(r["dist_cmps"].mean()), "qps_host_inclusive": qps}
    print(json.dumps(out), flush=True)


def c4shard(n):
    """one of the 8 shards of C4 (synthetic 100M x 128 fp16): 12.5M points, own sub-graph, all 10K queries"""
    t0 = time.time()
    X = datasets.sift_like(n, 128, seed=1234, dtype=np.float16); Q = datasets.sift_like(10_000, 128, seed=4321, dtype=np.float16)
    tg = time.time() - t0
    ix = DeviceIndex(X, max_degree=64)
    t0 = time.time(); st = ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb = time.time() - t0
    gt, gd = ix.bruteforce_knn(Q, 100)
    out = {"config": f"C4 one shard: {n}x128 fp16 Vamana R=64 L=128 a=1.15 x2, 10K queries", "datagen_s": tg, "build_s": tb,
           "build_phases_s": {"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s, "reprune": st.t_reprune_s}}
    for beam in (32, 64, 128):
        r, qps = search_stats(ix, Q, 10, beam)
        out[f"beam{beam}"] = {"recall": recall_at_k(r["ids"], gt, gd, 10), "visited": float(r["visited_count"].mean()),
                              "cmps": float(r["dist_cmps"].mean()), "qps_host_inclusive": qps}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    for a in sys.argv[1:] or ["c1"]:
        name, _, arg = a.partition(":")
        if name == "c1":
            c1()
        elif name == "c3":
            c3(int(arg or 10_000_000))
        elif name == "c5":
            c5(int(arg or 1_000_000))
        elif name == "c4shard":
            c4shard(int(arg or 12_500_000))
