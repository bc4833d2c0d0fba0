#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: QPS @ recall@10 >= 0.95 on SIFT-1M-shaped data (d=128, beam=64)
with the achieved fraction of the HBM roofline, 1..8 GPUs (one process per GPU, no collective on
the data path: queries shard embarrassingly, the index is replicated -- SURVEY.md section 8e).

A "step" = one batched beam search (pann_batch_search_dev) over --nq queries already resident in
HBM, results left in HBM.  Weak scaling: every rank searches its own --nq queries per step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--beam", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--R", type=int, default=64)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--alpha", type=float, default=1.15)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["f16", "u8", "f32"],
                    help="element type of the device copy (default f16 = BASELINE config[1]); the others are for comparison runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data", default="sift1m_like", choices=["sift1m_like", "sift_like"],
                    help="synthetic generator: sift1m_like is calibrated at n = 1M (BASELINE config[1]); sift_like (easier) keeps a "
                         "recall above 0.9 at beam 64 for tables far larger than 1M (the HBM-resident profile runs)")
    ap.add_argument("--strict", action="store_true", help="exit with status 3 when recall@10 < 0.95 (always on for the default workload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from parlayann_amd import DeviceIndex, datasets, _capi
    from parlayann_amd._capi import QueryParams, SearchOut, check
    from parlayann_amd.recall import recall_at_k

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    # one process per GPU.  (Rehearsals of N ranks on a box with fewer GPUs: ranks wrap around the visible
    # devices and PANN_BENCH_BACKEND=gloo replaces RCCL, which cannot put two ranks on one device.)
    dev_ord = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_ord)
    dev = torch.device("cuda", dev_ord)
    backend = os.environ.get("PANN_BENCH_BACKEND", "nccl")
    if "RANK" in os.environ:   # launched by torch.distributed.run (also with one rank: same code path)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL; used only for barriers + max-over-ranks
        else:
            dist.init_process_group(backend)

    # ---- synthetic SIFT-1M-shaped data: integer-valued in [0,255] (see datasets.sift1m_like: difficulty calibrated at n = 1M) ----
    t0 = time.time()
    gen = getattr(datasets, args.data)
    Xf = gen(args.n, args.d, seed=1234, dtype=np.float32)                     # the reference's float points
    np_dt = {"f16": np.float16, "u8": np.uint8, "f32": np.float32}[args.dtype]
    X = Xf.astype(np_dt)                                                      # "fp32 -> fp16": exact here (integer-valued)
    Q = gen(args.nq, args.d, seed=4321 + rank, dtype=np_dt)
    log(f"[rank {rank}] data generated in {time.time() - t0:.1f}s")

    # ---- index: replicated on every GPU, built on the device by the product's own builder ----
    ix = DeviceIndex(X, max_degree=args.R, device=dev_ord)
    t0 = time.time()
    bst = ix.vamana_build(args.R, args.L, args.alpha, num_passes=args.passes, seed=1, sort_neighbors=True)
    build_s = time.time() - t0
    log(f"[rank {rank}] vamana build n={args.n} R={args.R} L={args.L} passes={args.passes}: {build_s:.1f}s "
        f"(search {bst.t_search_s:.1f}s prune {bst.t_prune_s:.1f}s bidirect {bst.t_bidirect_s:.1f}s "
        f"reprune {bst.t_reprune_s:.1f}s)")

    # ---- device-resident inputs / outputs ----
    lib = _capi.load()
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(args.nq, -1)).to(dev)   # raw bytes of the query rows
    d_starts = torch.zeros(1, dtype=torch.int32, device=dev)      # start point 0 (check_nn_recall.h:178)
    d_ids = torch.empty((args.nq, args.k), dtype=torch.int32, device=dev)
    d_dists = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
    d_vis = torch.empty(args.nq, dtype=torch.int32, device=dev)
    d_cmps = torch.empty(args.nq, dtype=torch.int32, device=dev)
    d_deg = torch.empty(args.nq, dtype=torch.int32, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)      # the kernel's status word, copied here on the launch stream
    qp = QueryParams(k=args.k, beam=args.beam, cut=1.35, limit=args.n, degree_limit=args.R, rerank_factor=100, pad=1.0)
    out = SearchOut(ids=d_ids.data_ptr(), dists=d_dists.data_ptr(), out_k=args.k, frontier_size=None,
                    visited_count=d_vis.data_ptr(), dist_cmps=d_cmps.data_ptr(), degree_sum=d_deg.data_ptr(),
                    visited_ids=None, visited_dists=None, visited_cap=0, status=d_status.data_ptr())
    stream = torch.cuda.current_stream(dev)

    def step():
        check(lib.pann_batch_search_dev(ix.handle, d_q.data_ptr(), None, args.nq, args.d * Q.itemsize, d_starts.data_ptr(), 1,
                                        C.byref(qp), C.byref(out), C.c_void_p(stream.cuda_stream)))

    # ---- timed region: exactly --steps steps (barrier + synchronize on both sides, MAX over ranks:
    # parlayann_amd.distributed.timed_steps); per-launch HIP events on the launch stream ----
    from parlayann_amd import distributed as D
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = iter(evs)
    warm = [args.warmup]

    def timed_step():
        if warm[0] > 0:
            warm[0] -= 1
            step()
            return
        a, b = next(it)
        a.record(stream); step(); b.record(stream)

    elapsed = D.timed_steps(timed_step, args.steps, args.warmup, sync=lambda: torch.cuda.synchronize(dev),
                            device=dev if backend == "nccl" else None)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    status = int(d_status.item())
    if status:        # pann_search_out::status: bit 2 = internal scratch overflow, the results would not be the reference's
        raise SystemExit(f"bench.py: the timed launches reported status {status} (include/pann.h PANN_STATUS_*): results invalid")
    ms_per_step = elapsed * 1e3 / args.steps
    qps = args.nq * world / (ms_per_step / 1e3)

    # ---- algorithmic bytes per launch from the reference's own counters ----
    vis = d_vis.cpu().numpy().astype(np.int64); cmps = d_cmps.cpu().numpy().astype(np.int64)
    deg = d_deg.cpu().numpy().astype(np.int64)
    esize = Q.itemsize
    bytes_q = cmps * args.d * esize + (vis + deg) * 4 + args.d * esize + args.k * 8
    alg_bytes = int(bytes_q.sum())
    achieved = alg_bytes / (kern_ms / 1e3) / 1e9

    if rank == 0:
        # recall against exact ground truth (device brute force; tie-aware like checkRecall)
        gt_ids, gt_d = ix.bruteforce_knn(Q, 100)
        rec = recall_at_k(d_ids.cpu().numpy().view(np.uint32), gt_ids, gt_d, args.k)
        # HBM-side traffic of ONE launch: not measurable from inside this process (rocprofv3 counters need their own
        # passes); taken from the committed profile of the SAME workload when there is one, and labelled as such
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                for ent in (tj if isinstance(tj, list) else [tj]):
                    key = {"n": args.n, "nq": args.nq, "beam": args.beam, "dtype": args.dtype, "d": args.d, "R": args.R, "data": args.data}
                    if all(ent.get(k) == v for k, v in key.items()):
                        traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source", "profiles/traffic_latest.json")
            except Exception:
                traffic = None
        default_workload = (args.n == 1_000_000 and args.nq == 10_000 and args.beam == 64 and args.dtype == "f16" and args.d == 128
                            and args.data == "sift1m_like")
        # The 256 MB table of the default workload is about the size of the Infinity Cache.  The committed profile of the same
        # kernel on a 3.2 GB table (bench.py --n 12500000 --data sift_like, one C4 shard) is quoted beside it, labelled as such.
        hbm_ref = None
        if default_workload and os.path.exists(tpath):
            try:
                for ent in json.load(open(tpath)):
                    if ent.get("n") == 12_500_000 and "frac_algorithmic" in ent:
                        hbm_ref = {"n": ent["n"], "data": ent["data"], "queries_per_s": ent["qps"], "recall_at_10": ent["recall_at_10"],
                                   "kernel_ms": ent["kernel_ms_profiled"], "frac_algorithmic": ent["frac_algorithmic"],
                                   "frac_counted": ent["frac_counted"], "source": ent["source"],
                                   "note": "committed rocprofv3 run of this kernel on a table 12x the Infinity Cache; not measured in this run"}
            except Exception:
                hbm_ref = None
        res = {
            "metric": "QPS @ recall@10>=0.95, SIFT-1M d=128 beam=64; achieved HBM GB/s vs roofline",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SIFT-1M-shaped batched beam search: {args.n}x{args.d} integer-valued fp32->{args.dtype} base, "
                                   f"{args.nq} queries/step/GPU, beam={args.beam} k={args.k} cut=1.35 start=0, prebuilt "
                                   f"Vamana R={args.R} L={args.L} alpha={args.alpha} x{args.passes} passes (built on device)",
                       "n": args.n, "d": args.d, "nq_per_gpu": args.nq, "beam": args.beam, "k": args.k,
                       "parallelism": f"query-sharded x{world}, index replicated, no collective"},
            "recall_at_10": rec, "recall_ok": bool(rec >= 0.95),
            "avg_visited": float(vis.mean()), "avg_dist_cmps": float(cmps.mean()),
            "build_s": build_s,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # achieved / frac: ALGORITHMIC bytes (from the kernel's own counters) over the measured
                         # launch time.  traffic: counter-measured HBM-side bytes of one launch of this workload, from the
                         # committed rocprofv3 passes named in traffic_source (not from this run); traffic_frac = traffic / time / peak
                         "traffic_source": traffic_src,
                         "traffic_frac": (traffic / (kern_ms / 1e3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "hbm_resident_table_reference": hbm_ref,
                         "kernel": "beam_search_b64_kernel" if args.beam <= 64 else ("beam_search_b128_kernel" if args.beam <= 128 else "beam_search_kernel"), "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(ix, Xf, Q.astype(np.float32), args)
        print(json.dumps(res), flush=True)
        if rec < 0.95 and (args.strict or default_workload):
            log(f"bench.py: recall@10 = {rec:.4f} < 0.95: the metric is quoted AT recall >= 0.95, this line is invalid")
            ix.close()
            raise SystemExit(3)
    ix.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def usable_cpus():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(ix, Xf, Qf, args):
    """The reference's CPU path for the same workload: the oracle (a port of filtered_beam_search,
    one task per query like qsearchAll's parallel_for, beamSearch.h:556) on the host cores, float32
    points as Euclidian_Point<float> would hold them, same graph (downloaded from the device).
    Protocol of checkRecall: time only the batched search, several repetitions, best taken."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    o = oracle_api.load()
    G = ix.get_graph()
    threads = usable_cpus()
    best = None
    reps = 0
    t_all = time.time()
    while reps < 5 and time.time() - t_all < 25.0:
        t0 = time.perf_counter()
        o.batch_search(Xf, G, queries=Qf, k=args.k, beam=args.beam, cut=1.35, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    return {"value": len(Qf) / best, "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"all {len(Qf)} queries of the step, {reps} repetitions, best; float32 base (reference type), "
                      f"same device-built graph, {threads} std::threads"}


if __name__ == "__main__":
    main()
