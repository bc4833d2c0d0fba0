#!/usr/bin/env python3
"""Run the BASELINE.json configurations that fit one GPU (C1, C3, C5) end to end on the device and
print one JSON line each (build time, recall, QPS, counters).  Not a bench contract: the numbers go
to DESIGN.md.  usage: run_configs.py [c1] [quickstart] [c3[:n]] [c5[:n]] [c4shard[:n]]"""
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


def device_qps(ix, Q, k, beam, cut=1.35, reps=5):
    """device-resident timing like bench.py: queries already in HBM, HIP events around pann_batch_search_dev"""
    import ctypes as C
    import torch
    from parlayann_amd import _capi
    from parlayann_amd._capi import QueryParams, SearchOut, check
    lib = _capi.load()
    dev = torch.device("cuda", 0)
    nq = len(Q)
    d_q = torch.from_numpy(np.ascontiguousarray(Q).view(np.uint8).reshape(nq, -1)).to(dev)
    d_s = torch.zeros(1, dtype=torch.int32, device=dev)
    d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_vis = torch.empty(nq, dtype=torch.int32, device=dev); d_cm = torch.empty(nq, dtype=torch.int32, device=dev)
    qp = QueryParams(k=k, beam=beam, cut=cut, limit=ix.n, degree_limit=ix.max_degree, rerank_factor=100, pad=1.0)
    out = SearchOut(ids=d_ids.data_ptr(), dists=None, out_k=k, frontier_size=None, visited_count=d_vis.data_ptr(),
                    dist_cmps=d_cm.data_ptr(), degree_sum=None, visited_ids=None, visited_dists=None, visited_cap=0)
    st = torch.cuda.current_stream(dev)
    best = None
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        check(lib.pann_batch_search_dev(ix.handle, d_q.data_ptr(), None, nq, Q.shape[1] * Q.itemsize, d_s.data_ptr(), 1,
                                        C.byref(qp), C.byref(out), C.c_void_p(st.cuda_stream)))
        b.record(st); torch.cuda.synchronize(dev)
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    return d_ids.cpu().numpy().view(np.uint32), float(d_vis.float().mean()), float(d_cm.float().mean()), nq / (best / 1e3)


def quickstart():
    """the shape of docs/quickstart.md:37-101 (the only published numbers): 100K x 128 f32, Vamana R=32 L=64 a=1.2
    one pass, 10K queries, 10@10 at the beam widths of the published table (synthetic SIFT-shaped data)"""
    X = datasets.sift1m_like(100_000, 128, seed=1234, dtype=np.float32)
    Q = datasets.sift1m_like(10_000, 128, seed=4321, dtype=np.float32)
    ix = DeviceIndex(X, max_degree=32)
    ix.vamana_build(32, 64, 1.2, num_passes=1, seed=2)            # warm-up build (allocations, code load)
    t0 = time.time(); st = ix.vamana_build(32, 64, 1.2, num_passes=1, seed=1); tb = time.time() - t0
    G = ix.get_graph()
    gt, gd = ix.bruteforce_knn(Q, 100)
    out = {"config": "quickstart shape: 100K x 128 f32 Vamana R=32 L=64 a=1.2 x1, 10K queries (published: build 0.8123 s on 72 cores)",
           "build_s": tb, "build_phases_s": {"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s,
                                             "reprune": st.t_reprune_s},
           "avg_visited_per_insert": st.visited_total / 1e5, "avg_degree": float(G[:, 0].mean()), "max_degree": int(G[:, 0].max())}
    for beam, cut in ((12, 1.35), (17, 1.35), (45, 1.35), (70, 1.35), (1000, 10.0)):
        ids, vis, cm, qps = device_qps(ix, Q, 10, beam, cut)
        out[f"Q{beam}"] = {"recall": recall_at_k(ids, gt, gd, 10), "visited": vis, "cmps": cm, "qps_device_resident": qps}
    print(json.dumps(out), flush=True)


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
           "build_s": tb, "build_phases_s": wrapper.hcnng_build.last_times,
           "build_device_s": float(sum(wrapper.hcnng_build.last_times.values())),      # build_s minus the PCIe copies of points and graph
           "avg_degree": float(G[:, 0].mean()),
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


def _fill_chunk(args):
    path, n_total, d, lo, hi, seed = args
    mm = np.memmap(path, dtype=np.float16, mode="r+", shape=(n_total, d))
    # every slice has its own 256 cluster centres (a larger corpus has more modes, not denser ones); queries use slice 0's
    x = datasets._mixture(hi - lo, d, seed, 256, 16, center_scale=22.0, basis_scale=9.0, noise_scale=12.0, centers_seed=seed)
    mm[lo:hi] = np.clip(np.rint(x + 100.0), 0, 255).astype(np.float16)
    mm.flush()
    return hi - lo


def c4full(n):
    """all of C4 (synthetic 100M x 128 fp16) on ONE MI355X: 25.6 GB of points + 25.6 GB of graph in its 288 GB of HBM;
    the base is generated by 8 worker processes into a shared memory map (seeds 1234..1241, one per 12.5M slice)"""
    import multiprocessing as mp
    d, parts = 128, 8
    path = "/dev/shm/pann_c4_base.f16"
    t0 = time.time()
    mm = np.memmap(path, dtype=np.float16, mode="w+", shape=(n, d)); del mm
    step = (n + parts - 1) // parts
    with mp.get_context("spawn").Pool(parts) as pool:
        pool.map(_fill_chunk, [(path, n, d, i * step, min(n, (i + 1) * step), 1234 + i) for i in range(parts)])
    X = np.memmap(path, dtype=np.float16, mode="r", shape=(n, d))
    qs = [datasets._mixture(10_000 // parts, d, 4321 + i, 256, 16, center_scale=22.0, basis_scale=9.0, noise_scale=12.0,
                            centers_seed=1234 + i) for i in range(parts)]       # queries from every slice's mixture
    Q = np.clip(np.rint(np.concatenate(qs) + 100.0), 0, 255).astype(np.float16)
    tg = time.time() - t0
    try:
        t0 = time.time(); ix = DeviceIndex(X, max_degree=64); tu = time.time() - t0
        t0 = time.time(); st = ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb = time.time() - t0
        t0 = time.time(); gt, gd = ix.bruteforce_knn(Q, 100); tgt = time.time() - t0
        out = {"config": f"C4 on one GPU: {n}x128 fp16 Vamana R=64 L=128 a=1.15 x2, 10K queries", "datagen_s": tg, "upload_s": tu,
               "build_s": tb, "groundtruth_s": tgt,
               "build_phases_s": {"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s, "reprune": st.t_reprune_s}}
        for beam in (32, 64, 128):
            ids, vis, cm, qps = device_qps(ix, Q, 10, beam)
            out[f"beam{beam}"] = {"recall": recall_at_k(ids, gt, gd, 10), "visited": vis, "cmps": cm, "qps_device_resident": qps}
        print(json.dumps(out), flush=True)
    finally:
        os.remove(path)


if __name__ == "__main__":
    for a in sys.argv[1:] or ["c1"]:
        name, _, arg = a.partition(":")
        if name == "c1":
            c1()
        elif name == "quickstart":
            quickstart()
        elif name == "c3":
            c3(int(arg or 10_000_000))
        elif name == "c5":
            c5(int(arg or 1_000_000))
        elif name == "c4shard":
            c4shard(int(arg or 12_500_000))
        elif name == "c4full":
            c4full(int(arg or 100_000_000))
