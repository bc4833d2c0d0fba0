#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: QPS @ recall@10 >= 0.95 on SIFT-1M-shaped data (d=128, beam=64)
with the achieved fraction of the HBM roofline, 1..8 GPUs, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode query|c4|c3build|c5build]

`--gpus N` with N > 1 and no RANK in the environment: this process only LAUNCHES -- it starts N fresh child
processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything here has touched the GPU,
waits, relays rank 0's JSON line and exits non-zero if any child did (the reference's driver forks its own workers
too: bench/neighborsTime.C:50-70 -> parlay::parallel_for, utils/beamSearch.h:556).  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks are already there.

Modes (a "step" is one pass of the mode's hot path, inputs resident in HBM):
  query   (default) replicated index, queries sharded: one batched beam search (pann_batch_search_dev) over --nq queries
          per rank; no collective on the data path.  Weak scaling.  BASELINE config[1]; the only mode with the headline metric.
  c4      sharded index: rank r owns --n base points (default 12.5M: 100M over 8) with its own sub-graph; every query
          goes to every rank; ONE all-gather of the packed top-k rows + device merge per step.  Weak scaling (corpus grows).
  c3build Vamana build (DEEP-shaped --n x 96 f32, default 10M) with every batch of batch_insert split over the ranks
          (vamana/index.h:188-316 is what the ranks run between the collectives); one step = one whole build.  Strong scaling.
  c5build HCNNG build (T2I-shaped --n x 200 int8 MIPS, default 10M, 30 trees) with the trees split over the ranks; one
          step = one whole build.  Strong scaling.
  hbm_leg the second leg of the default run by itself (12.5M x 128 fp16 table, same kernel / beam / k): what the rocprofv3 passes
          of profiles/r03_bench12m_* run, so that its launches are not mixed with the 1M-table launches of the same grid size.
  launchcheck  no GPU work at all: the launcher, the process group (gloo when no GPU is visible) and the timing contract
          with an empty step; prints n_gpus / ranks_seen.  What tests/test_bench_launcher_cpu.py runs.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MODES = ("query", "c4", "c3build", "c5build", "hbm_leg", "launchcheck")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 2 for the build modes)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 1 for the build modes)")
    ap.add_argument("--mode", default="query", choices=MODES)
    ap.add_argument("--n", type=int, default=None, help="base points (per rank in c4); default per mode")
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--beam", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--R", type=int, default=64)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--alpha", type=float, default=1.15)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--trees", type=int, default=30)
    ap.add_argument("--dtype", default="f16", choices=["f16", "u8", "f32"],
                    help="element type of the device copy (default f16 = BASELINE config[1]); the others are for comparison runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true",
                    help="skip the second, labelled leg of the default run (the same kernel on a 12.5M-point, 3.2 GB table)")
    ap.add_argument("--hbm-n", type=int, default=12_500_000)
    ap.add_argument("--hbm-noise", type=float, default=None, help="noise scale of the HBM leg's generator (calibration runs)")
    ap.add_argument("--data", default="sift1m_like", choices=["sift1m_like", "sift_like"],
                    help="synthetic generator: sift1m_like is calibrated at n = 1M (BASELINE config[1]); sift_like (easier) keeps a "
                         "recall above 0.9 at beam 64 for tables far larger than 1M")
    ap.add_argument("--strict", action="store_true", help="exit with status 3 when recall@10 < 0.95 (always on for the default workload)")
    args = ap.parse_args(argv)
    build = args.mode in ("c3build", "c5build")
    if args.steps is None:
        args.steps = 2 if build else 20
    if args.warmup is None:
        args.warmup = 1 if build else 3
    if args.n is None:
        args.n = {"query": 1_000_000, "c4": 12_500_000, "c3build": 10_000_000, "c5build": 10_000_000, "hbm_leg": 12_500_000,
                  "launchcheck": 0}[args.mode]
    return args


# ---------------------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  (N > 1, not under torch.distributed.run)
# ---------------------------------------------------------------------------------------------------------------------

def launch_ranks(n_ranks, argv):
    """Start n_ranks fresh children of this script, one per GPU, BEFORE this process has imported torch or made any HIP call
    (a process that has initialised the GPU must not fork or exec GPU work); relay rank 0's stdout; return the exit status."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PANN_BENCH_LAUNCHER=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []

    def relay():
        for raw in procs[0].stdout:
            line = raw.decode("utf-8", "replace")
            if line.lstrip().startswith("{"):
                lines.append(line)                      # rank 0's JSON line(s): the launcher's own stdout
            else:
                sys.stderr.write(line)                  # anything else a library printed on stdout (gloo's connection notes)
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    live = set(range(n_ranks))
    while live:
        time.sleep(0.2)
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                log(f"bench.py launcher: rank {r} exited with {code}; stopping the other ranks")
                for o in live:
                    procs[o].terminate()          # exactly the children started above, by handle
    t.join(timeout=10)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------

class Rank:
    """process-group plumbing of one rank: device, backend, barrier, ranks_seen"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("PANN_BENCH_BACKEND", "nccl")
        self.gpu = torch.cuda.is_available()
        if not self.gpu and args.mode != "launchcheck":
            raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
        if self.gpu:
            # one process per GPU.  (Rehearsals of N ranks on a box with fewer GPUs: ranks wrap around the visible
            # devices and PANN_BENCH_BACKEND=gloo replaces RCCL, which cannot put two ranks on one device.)
            self.dev_ord = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(self.dev_ord)
            self.dev = torch.device("cuda", self.dev_ord)
        else:                                   # launchcheck on a box without a GPU: process-group plumbing only
            self.dev_ord, self.dev, self.backend = -1, None, "gloo"
        if "RANK" in os.environ:   # launched by this file's launcher or by torch.distributed.run (also with one rank: same code path)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)   # RCCL
            else:
                dist.init_process_group(self.backend)
            if dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        elif args.gpus != 1:
            raise SystemExit("bench.py: --gpus N > 1 without a process group (internal error: the launcher should have run)")
        self.cdev = self.dev if self.backend == "nccl" else None      # where collective payloads live
        # every rank reports in: ranks_seen == [0 .. N-1] in the line proves N processes joined the group
        self.ranks_seen = [0]
        if dist.is_initialized():
            t = torch.tensor([self.rank], dtype=torch.int64, device=self.cdev or "cpu")
            out = torch.empty(self.world, dtype=torch.int64, device=self.cdev or "cpu")
            dist.all_gather_into_tensor(out, t)
            self.ranks_seen = [int(v) for v in out.cpu()]
            devs = torch.tensor([self.dev_ord], dtype=torch.int64, device=self.cdev or "cpu")
            dout = torch.empty(self.world, dtype=torch.int64, device=self.cdev or "cpu")
            dist.all_gather_into_tensor(dout, devs)
            self.devices_seen = [int(v) for v in dout.cpu()]
        else:
            self.devices_seen = [self.dev_ord]

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize(self.dev)

    def timed(self, step, steps, warmup):
        from parlayann_amd import distributed as D
        return D.timed_steps(step, steps, warmup, sync=self.sync, device=self.cdev)

    def sum_over_ranks(self, values):
        """element-wise sum of a short list of floats over the ranks (reporting only, outside the timed region)"""
        if not self.dist.is_initialized():
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.cdev or "cpu")
        self.dist.all_reduce(t)
        return [float(v) for v in t.cpu()]

    def finish(self):
        if self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()


def kernel_name(beam):
    return "beam_search_b64_kernel" if beam <= 64 else ("beam_search_b128_kernel" if beam <= 128 else "beam_search_kernel")


def search_leg(rk, ix, Q, args, beam, steps, warmup):
    """`steps` timed launches of pann_batch_search_dev over the rows of Q (already resident), per-launch HIP events on the
    launch stream; returns the numbers both the headline and the HBM-resident leg report"""
    import numpy as np
    torch = rk.torch
    from parlayann_amd import _capi
    from parlayann_amd._capi import QueryParams, SearchOut, check
    lib = _capi.load()
    dev, nq, k = rk.dev, len(Q), args.k
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(nq, -1)).to(dev)   # raw bytes of the query rows
    d_starts = torch.zeros(1, dtype=torch.int32, device=dev)      # start point 0 (check_nn_recall.h:178)
    d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
    d_vis = torch.empty(nq, dtype=torch.int32, device=dev)
    d_cmps = torch.empty(nq, dtype=torch.int32, device=dev)
    d_deg = torch.empty(nq, dtype=torch.int32, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)      # the kernel's status word, copied here on the launch stream
    qp = QueryParams(k=k, beam=beam, cut=1.35, limit=ix.n, degree_limit=ix.max_degree, rerank_factor=100, pad=1.0)
    out = SearchOut(ids=d_ids.data_ptr(), dists=d_dists.data_ptr(), out_k=k, frontier_size=None,
                    visited_count=d_vis.data_ptr(), dist_cmps=d_cmps.data_ptr(), degree_sum=d_deg.data_ptr(),
                    visited_ids=None, visited_dists=None, visited_cap=0, status=d_status.data_ptr())
    stream = torch.cuda.current_stream(dev)
    row_bytes = Q.shape[1] * Q.itemsize

    def step():
        check(lib.pann_batch_search_dev(ix.handle, d_q.data_ptr(), None, nq, row_bytes, d_starts.data_ptr(), 1,
                                        C.byref(qp), C.byref(out), C.c_void_p(stream.cuda_stream)))

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    it = iter(evs)
    warm = [warmup]

    def timed_step():
        if warm[0] > 0:
            warm[0] -= 1
            step()
            return
        a, b = next(it)
        a.record(stream); step(); b.record(stream)

    # timed region: exactly `steps` steps (barrier + synchronize on both sides, MAX over ranks: distributed.timed_steps)
    elapsed = rk.timed(timed_step, steps, warmup)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    status = int(d_status.item())
    if status:        # pann_search_out::status: bit 2 = internal scratch overflow, the results would not be the reference's
        raise SystemExit(f"bench.py: the timed launches reported status {status} (include/pann.h PANN_STATUS_*): results invalid")
    # algorithmic bytes per launch from the reference's own counters
    vis = d_vis.cpu().numpy().astype(np.int64); cmps = d_cmps.cpu().numpy().astype(np.int64)
    deg = d_deg.cpu().numpy().astype(np.int64)
    bytes_q = cmps * row_bytes + (vis + deg) * 4 + row_bytes + k * 8
    alg_bytes = int(bytes_q.sum())
    return {"elapsed_s": elapsed, "ms_per_step": elapsed * 1e3 / steps, "kernel_ms": kern_ms, "alg_bytes": alg_bytes,
            "achieved_gbps": alg_bytes / (kern_ms / 1e3) / 1e9, "avg_visited": float(vis.mean()), "avg_cmps": float(cmps.mean()),
            "ids": d_ids}


def committed_traffic(args, key=None):
    """HBM-side bytes of ONE launch are not measurable from inside this process (rocprofv3 counters need their own passes):
    taken from the committed profile of the SAME workload when there is one, and labelled as such"""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        tj = json.load(open(tpath))
        if key is None:
            key = {"n": args.n, "nq": args.nq, "beam": args.beam, "dtype": args.dtype, "d": args.d, "R": args.R, "data": args.data}
        for ent in (tj if isinstance(tj, list) else [tj]):
            if all(ent.get(k) == v for k, v in key.items()):
                return ent.get("hbm_bytes_per_launch"), ent.get("source", "profiles/traffic_latest.json")
    except Exception:
        pass
    return None, None


# The second leg's generator: datasets.sift_like's geometry drawn on the device; the noise scale is calibrated
# (tools/calibrate_hbm_leg.sh) so that recall@10 >= 0.95 at beam 64 on 12.5M points, i.e. the leg is quoted
# at the metric's recall.
HBM_LEG_NOISE = 10.0


def hbm_resident_leg(rk, args):
    """The headline's 256 MB table is about the size of the 256 MiB Infinity Cache, so its rate is not an HBM number.  This leg
    runs the SAME kernel, beam and k on a table 12x the cache (12.5M x 128 fp16 = 3.2 GB = one C4 shard), built on the device,
    and is reported beside the headline as roofline.hbm_resident -- measured in this run, never `value`."""
    import numpy as np
    from parlayann_amd import DeviceIndex, datasets
    from parlayann_amd.recall import recall_at_k
    noise = HBM_LEG_NOISE if args.hbm_noise is None else args.hbm_noise
    t0 = time.time()
    X = datasets.sift_like_device(args.hbm_n, 128, 1234, rk.dev, np.float16, noise_scale=noise)
    Q = datasets.sift_like_device(args.nq, 128, 4321, rk.dev, np.float16, noise_scale=noise)
    tgen = time.time() - t0
    ix = DeviceIndex(X, max_degree=64, device=rk.dev_ord)
    t0 = time.time()
    ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1, sort_neighbors=True)
    build_s = time.time() - t0
    log(f"[hbm leg] {args.hbm_n}x128 f16 generated in {tgen:.1f}s, built in {build_s:.1f}s")
    r = search_leg(rk, ix, Q, args, 64, args.steps, args.warmup)
    gt_ids, gt_d = ix.bruteforce_knn(Q, 100)
    rec = recall_at_k(r["ids"].cpu().numpy().view(np.uint32), gt_ids, gt_d, args.k)
    ix.close()
    traffic, traffic_src = (None, None)
    if args.hbm_noise is None:
        traffic, traffic_src = committed_traffic(args, {"n": args.hbm_n, "nq": args.nq, "beam": 64, "dtype": "f16", "d": 128, "R": 64, "data": "hbm_leg"})
    return {"traffic": traffic, "traffic_source": traffic_src,
            "traffic_frac": (traffic / (r["kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            "workload": f"{args.hbm_n}x128 integer-valued f16 (sift_like geometry, noise {noise}, generated on the device), "
                        f"Vamana R=64 L=128 alpha=1.15 x2 built on the device, {args.nq} queries/step, beam=64 k={args.k}",
            "table_bytes": int(args.hbm_n) * 256, "queries_per_s": args.nq / (r["ms_per_step"] / 1e3), "ms_per_step": r["ms_per_step"],
            "kernel": kernel_name(64), "kernel_ms": r["kernel_ms"], "algorithmic_bytes_per_launch": r["alg_bytes"],
            "achieved": r["achieved_gbps"], "unit": "GB/s", "frac": r["achieved_gbps"] / HBM_PEAK_GBPS,
            "recall_at_10": rec, "recall_ok": bool(rec >= 0.95), "avg_visited": r["avg_visited"], "avg_dist_cmps": r["avg_cmps"],
            "steps": args.steps, "warmup": args.warmup, "build_s": build_s, "measured_in_this_run": True}


def mode_query(rk, args):
    import numpy as np
    from parlayann_amd import DeviceIndex, datasets
    from parlayann_amd.recall import recall_at_k
    rank, world = rk.rank, rk.world
    # ---- synthetic SIFT-1M-shaped data: integer-valued in [0,255] (see datasets.sift1m_like: difficulty calibrated at n = 1M) ----
    t0 = time.time()
    gen = getattr(datasets, args.data)
    Xf = gen(args.n, args.d, seed=1234, dtype=np.float32)                     # the reference's float points
    np_dt = {"f16": np.float16, "u8": np.uint8, "f32": np.float32}[args.dtype]
    X = Xf.astype(np_dt)                                                      # "fp32 -> fp16": exact here (integer-valued)
    Q = gen(args.nq, args.d, seed=4321 + rank, dtype=np_dt)
    log(f"[rank {rank}] data generated in {time.time() - t0:.1f}s")

    # ---- index: replicated on every GPU, built on the device by the product's own builder ----
    ix = DeviceIndex(X, max_degree=args.R, device=rk.dev_ord)
    t0 = time.time()
    bst = ix.vamana_build(args.R, args.L, args.alpha, num_passes=args.passes, seed=1, sort_neighbors=True)
    build_s = time.time() - t0
    log(f"[rank {rank}] vamana build n={args.n} R={args.R} L={args.L} passes={args.passes}: {build_s:.1f}s "
        f"(search {bst.t_search_s:.1f}s prune {bst.t_prune_s:.1f}s bidirect {bst.t_bidirect_s:.1f}s "
        f"reprune {bst.t_reprune_s:.1f}s)")

    r = search_leg(rk, ix, Q, args, args.beam, args.steps, args.warmup)
    ms_per_step = r["ms_per_step"]
    qps = args.nq * world / (ms_per_step / 1e3)
    res = None
    default_workload = (args.n == 1_000_000 and args.nq == 10_000 and args.beam == 64 and args.dtype == "f16" and args.d == 128
                        and args.data == "sift1m_like")
    if rank == 0:
        # recall against exact ground truth (device brute force; tie-aware like checkRecall)
        gt_ids, gt_d = ix.bruteforce_knn(Q, 100)
        rec = recall_at_k(r["ids"].cpu().numpy().view(np.uint32), gt_ids, gt_d, args.k)
        traffic, traffic_src = committed_traffic(args)
        kern_ms, achieved = r["kernel_ms"], r["achieved_gbps"]
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
            "avg_visited": r["avg_visited"], "avg_dist_cmps": r["avg_cmps"],
            "build_s": build_s,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # achieved / frac: ALGORITHMIC bytes (from the kernel's own counters) over the measured
                         # launch time.  traffic: counter-measured HBM-side bytes of one launch of this workload, from the
                         # committed rocprofv3 passes named in traffic_source (not from this run); traffic_frac = traffic / time / peak
                         "traffic_source": traffic_src,
                         "traffic_frac": (traffic / (kern_ms / 1e3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "kernel": kernel_name(args.beam), "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": r["alg_bytes"]},
            "ranks_seen": rk.ranks_seen, "devices_seen": rk.devices_seen, "backend": rk.backend if world > 1 or "RANK" in os.environ else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(ix, Xf, Q.astype(np.float32), args)
    del Xf, X
    ix.close()
    if rank == 0:
        if world == 1 and default_workload and not args.no_hbm_leg:
            res["roofline"]["hbm_resident"] = hbm_resident_leg(rk, args)
        print(json.dumps(res), flush=True)
        if res["recall_at_10"] < 0.95 and (args.strict or default_workload):
            log(f"bench.py: recall@10 = {res['recall_at_10']:.4f} < 0.95: the metric is quoted AT recall >= 0.95, this line is invalid")
            return 3
    return 0


def mode_c4(rk, args):
    """Sharded index (BASELINE config[3]: 100M x 128 fp16 over 8 shards -> 12.5M per rank): distributed.DeviceShardedIndex"""
    import numpy as np
    torch = rk.torch
    from parlayann_amd import datasets, distributed as D
    from parlayann_amd.recall import recall_at_k
    rank, world = rk.rank, rk.world
    n_total = args.n * world
    lo, hi = D.shard_range(n_total, rank, world)
    t0 = time.time()
    # slice r of the corpus has its own 256 cluster centres (tools/run_configs.py c4full uses the same rule); the queries are
    # drawn from every slice's distribution in turn, identical on every rank
    X = datasets.sift_like_device(hi - lo, 128, 1234 + rank, rk.dev, np.float16, centers_seed=1234 + rank)
    per = max(1, args.nq // world)
    Q = np.concatenate([datasets.sift_like_device(per, 128, 4321 + i, rk.dev, np.float16, centers_seed=1234 + i) for i in range(world)])
    tgen = time.time() - t0
    tb = [0.0]

    def build(ix):
        t0 = time.time(); ix.vamana_build(args.R, args.L, args.alpha, num_passes=args.passes, seed=1); tb[0] = time.time() - t0

    sh = D.DeviceShardedIndex(X, args.R, build, device_ordinal=rk.dev_ord, n_total=n_total)
    log(f"[rank {rank}] shard [{lo}, {hi}) generated in {tgen:.1f}s, built in {tb[0]:.1f}s")
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(len(Q), -1)).to(rk.dev)
    last = [None]

    def step():
        last[0] = sh.search(d_q, args.k, args.beam, counters=True)

    c0 = sh.collectives
    elapsed = rk.timed(step, args.steps, args.warmup)
    colls = sh.collectives - c0
    ms_per_step = elapsed * 1e3 / args.steps
    vis, cmps, deg = (t.cpu().numpy().astype(np.int64) for t in sh.last_counters)
    alg_local = int((cmps * 256 + (vis + deg) * 4 + 256 + args.k * 8).sum())
    alg_all = rk.sum_over_ranks([alg_local])[0]
    gt_i, gt_d = sh.bruteforce(Q, 100)                                  # exact ground truth over all shards, same exchange
    if rank == 0:
        oi, _ = last[0]
        rec = recall_at_k(oi.cpu().numpy().view(np.uint32), gt_i.cpu().numpy().view(np.uint32), gt_d.cpu().numpy(), args.k)
        achieved = alg_all / (ms_per_step / 1e3) / 1e9
        print(json.dumps({
            "metric": "sharded-index QPS (every query on every shard, one all-gather + device merge per step)",
            "value": len(Q) / (ms_per_step / 1e3), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"C4 sharded index: {n_total}x128 fp16 in {world} shards of {args.n} (own Vamana R={args.R} L={args.L} "
                                   f"x{args.passes} per shard, built on device), {len(Q)} queries/step to every shard, beam={args.beam} k={args.k}",
                       "n_total": n_total, "n_per_gpu": args.n, "nq": len(Q), "beam": args.beam, "k": args.k,
                       "parallelism": f"index-sharded x{world}; ONE all-gather of {args.k * 8} B per query per rank + device merge"},
            "recall_at_10": rec, "build_s_rank0": tb[0], "collectives_per_step": colls / (args.steps + args.warmup),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBPS * world), "traffic": None, "kernel": kernel_name(args.beam),
                         "note": "algorithmic bytes of all ranks' local searches over the whole step (search + status read + "
                                 "all-gather + merge), not over the kernel alone"},
            "ranks_seen": rk.ranks_seen, "devices_seen": rk.devices_seen, "backend": rk.backend}), flush=True)
    sh.close()


def graph_checksum(G):
    import numpy as np
    return int(np.bitwise_xor.reduce(G.ravel().astype(np.uint64) * np.arange(1, G.size + 1, dtype=np.uint64)))


def mode_c3build(rk, args):
    """Vamana build, every batch split over the ranks (BASELINE config[2] shape: DEEP-like n x 96 f32, R=64 L=128 alpha 1.05 x2)"""
    import numpy as np
    from parlayann_amd import DeviceIndex, datasets, distributed as D
    from parlayann_amd.recall import recall_at_k
    rank, world = rk.rank, rk.world
    t0 = time.time()
    X = datasets.deep_like(args.n, 96, seed=1234); Q = datasets.deep_like(args.nq, 96, seed=4321)
    log(f"[rank {rank}] data generated in {time.time() - t0:.1f}s")
    ix = DeviceIndex(X, max_degree=args.R, device=rk.dev_ord)
    last = [None]

    def step():
        ix.clear_graph()
        last[0] = D.device_vamana_build_sharded(ix, args.R, args.L, 1.05, num_passes=args.passes, seed=1)

    elapsed = rk.timed(step, args.steps, args.warmup)
    s_per_step = elapsed / args.steps
    info = last[0]; st = info["stats"]
    cs = graph_checksum(ix.get_graph()) if args.n <= 20_000_000 else None
    if rank == 0:
        r = ix.batch_search(Q, k=args.k, beam=args.beam)
        gt, gd = ix.bruteforce_knn(Q, 100)
        rec = recall_at_k(r["ids"], gt, gd, args.k)
        # phase A of this rank: search_dist_cmps comparisons x the 384-byte row each reads, over the time spent in the searches
        alg = float(st.search_dist_cmps) * 96 * 4
        print(json.dumps({
            "metric": "Vamana build throughput (batch_insert split over the ranks, one all-gather of new rows per batch)",
            "value": args.n / s_per_step, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": s_per_step * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C3 Vamana build: DEEP-shaped {args.n}x96 f32, R={args.R} L={args.L} alpha=1.05 x{args.passes} passes, "
                                   f"points + graph replicated, inserts of every batch split over {world} ranks",
                       "n": args.n, "parallelism": f"batch-sharded x{world}; one all-gather of m x R x 4 B per batch"},
            "build_s": s_per_step, "collectives_per_build": info["collectives"], "bytes_gathered_per_build": info["bytes_gathered"],
            "rank0_phases_s_last_build": {"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s, "reprune": st.t_reprune_s},
            "recall_at_10": rec, "graph_checksum": cs,
            "roofline": {"bound": "hbm", "achieved": alg / max(st.t_search_s, 1e-9) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": alg / max(st.t_search_s, 1e-9) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "kernel": "beam_search_b128_kernel",
                         "note": "rank 0's share of the build-time searches: comparisons x 384 B over its search-phase seconds"},
            "ranks_seen": rk.ranks_seen, "devices_seen": rk.devices_seen, "backend": rk.backend}), flush=True)
    ix.close()


def mode_c5build(rk, args):
    """HCNNG build, trees split over the ranks (BASELINE config[4] shape: T2I-like n x 200 f32 -> int8, MIPS, 30 trees x 1000 x 3)"""
    import numpy as np
    from parlayann_amd import DeviceIndex, datasets, quantize, distributed as D
    from parlayann_amd.recall import recall_at_k
    rank, world = rk.rank, rk.world
    t0 = time.time()
    Xf = datasets.t2i_like(args.n, 200, seed=1234); Qf = datasets.t2i_like(args.nq, 200, seed=4321)
    mv = quantize.mips_i8_max_val(Xf, trim=False)
    X, Q = quantize.mips_i8_translate(Xf, mv), quantize.mips_i8_translate(Qf, mv); del Xf, Qf
    log(f"[rank {rank}] data generated + quantised in {time.time() - t0:.1f}s")
    mst = 3
    ix = DeviceIndex(X, max_degree=args.trees * mst, metric="mips", device=rk.dev_ord)       # ONE resident index per rank
    last = [None]

    def step():
        ix.clear_graph()
        last[0] = D.device_hcnng_build_tree_parallel(ix, args.trees, 1000, mst, seed=1)

    elapsed = rk.timed(step, args.steps, args.warmup)
    s_per_step = elapsed / args.steps
    info = last[0]
    G = ix.get_graph()
    if rank == 0:
        r = ix.batch_search(Q, k=args.k, beam=args.beam)
        gt, gd = ix.bruteforce_knn(Q, 100)
        rec = recall_at_k(r["ids"], gt, gd, args.k)
        print(json.dumps({
            "metric": "HCNNG build throughput (cluster trees split over the ranks, one all-gather of the edge slabs)",
            "value": args.n / s_per_step, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": s_per_step * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": f"C5 HCNNG build: T2I-shaped {args.n}x200 f32->int8 MIPS, {args.trees} trees x leaf 1000 x MST degree {mst}, "
                                   f"points replicated, trees split over {world} ranks",
                       "n": args.n, "parallelism": f"tree-sharded x{world}; one all-gather of the per-rank edge slabs"},
            "build_s": s_per_step, "bytes_gathered_per_build": info["bytes_gathered"],
            "rank0_phases_s_last_build": {k: float(info[k]) for k in ("tree_s", "leaf_knn_s", "mst_s")},
            "avg_degree": float(G[:, 0].mean()), "recall_at_10": rec, "graph_checksum": graph_checksum(G),
            "roofline": None,
            "ranks_seen": rk.ranks_seen, "devices_seen": rk.devices_seen, "backend": rk.backend}), flush=True)
    ix.close()


def mode_hbm_leg(rk, args):
    args.hbm_n = args.n
    leg = hbm_resident_leg(rk, args)
    if rk.rank == 0:
        print(json.dumps({
            "metric": "QPS @ recall@10>=0.95 on the HBM-resident table (second leg of the default run, by itself)",
            "value": leg["queries_per_s"], "unit": "queries/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic", "config": {"workload": leg["workload"], "n": args.n, "d": 128, "nq_per_gpu": args.nq, "beam": 64, "k": args.k},
            "recall_at_10": leg["recall_at_10"], "recall_ok": leg["recall_ok"], "avg_visited": leg["avg_visited"],
            "avg_dist_cmps": leg["avg_dist_cmps"], "build_s": leg["build_s"],
            "roofline": {"bound": "hbm", "achieved": leg["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": leg["frac"],
                         "traffic": None, "kernel": leg["kernel"], "kernel_ms": leg["kernel_ms"],
                         "algorithmic_bytes_per_launch": leg["algorithmic_bytes_per_launch"]}}), flush=True)
    return 0 if leg["recall_ok"] else 3


def mode_launchcheck(rk, args):
    elapsed = rk.timed(lambda: None, args.steps, args.warmup)
    if rk.rank == 0:
        print(json.dumps({"metric": "launch check (no GPU work)", "value": 0.0, "unit": "none", "n_gpus": rk.world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / max(args.steps, 1), "ranks_seen": rk.ranks_seen,
                          "devices_seen": rk.devices_seen, "backend": rk.backend,
                          "launcher_pid": os.environ.get("PANN_BENCH_LAUNCHER")}), flush=True)
    return 3 if os.environ.get("PANN_BENCH_FAIL_RANK") == str(rk.rank) else 0      # test hook: a failing rank fails the launch


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


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # nothing above has imported torch or touched the GPU
    rk = Rank(args)
    rc = {"query": mode_query, "c4": mode_c4, "c3build": mode_c3build, "c5build": mode_c5build, "hbm_leg": mode_hbm_leg,
          "launchcheck": mode_launchcheck}[args.mode](rk, args)
    rk.finish()
    sys.exit(rc or 0)


if __name__ == "__main__":
    main()
