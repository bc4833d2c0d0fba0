#!/usr/bin/env python3
"""Multi-GPU paths of SURVEY.md section 8(e) as a runnable driver, one process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
      tools/dist_run.py c4 --points 100000000          # sharded index: rank r owns ids [r*n/8, (r+1)*n/8)
  ... tools/dist_run.py c5 --points 10000000           # HCNNG: rank r builds trees r, r+W, ...; one all-gather of the slabs

c4: every rank generates and builds ONLY its own slice (own sub-graph, local ids), every query goes to every rank,
    the per-rank top-k lists (k*8 bytes per query) are all-gathered and merged by (dist,id).  No other collective.
c5: points are replicated; the 30 cluster trees are split over the ranks; ONE all-gather of the per-tree edge slabs,
    then every rank assembles the identical graph (tree order) and uploads it.
PANN_DIST_BACKEND=gloo rehearses on a box with fewer GPUs than ranks (ranks wrap around the visible devices)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["c4", "c5"])
    ap.add_argument("--points", type=int, default=0)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--beam", type=int, default=64)
    ap.add_argument("--trees", type=int, default=30)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from parlayann_amd import DeviceIndex, datasets, quantize, distributed as D
    from parlayann_amd.recall import recall_at_k
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev_ord = int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()
    torch.cuda.set_device(dev_ord)
    backend = os.environ.get("PANN_DIST_BACKEND", "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_ord))
        cdev = torch.device("cuda", dev_ord)
    else:
        dist.init_process_group(backend)
        cdev = None
    out = {"mode": args.mode, "world": world, "backend": backend}
    if args.mode == "c4":
        n = args.points or 100_000_000
        lo, hi = D.shard_range(n, rank, world)
        t0 = time.time()
        # slice r of the corpus: its own 256 cluster centres (tools/run_configs.py c4full uses the same rule)
        x = datasets._mixture(hi - lo, 128, 1234 + rank, 256, 16, center_scale=22.0, basis_scale=9.0, noise_scale=12.0,
                              centers_seed=1234 + rank)
        X = np.clip(np.rint(x + 100.0), 0, 255).astype(np.float16); del x
        qs = [datasets._mixture(args.queries // world, 128, 4321 + i, 256, 16, center_scale=22.0, basis_scale=9.0, noise_scale=12.0,
                                centers_seed=1234 + i) for i in range(world)]
        Q = np.clip(np.rint(np.concatenate(qs) + 100.0), 0, 255).astype(np.float16)
        tgen = time.time() - t0
        ix = DeviceIndex(X, max_degree=64, device=dev_ord)
        t0 = time.time(); ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb = time.time() - t0
        D.barrier()

        def local_search(k, beam):
            r = ix.batch_search(Q, k=k, beam=beam)
            return r["ids"], r["dists"]

        def sharded(k, beam, local=local_search):
            lid, d = local(k, beam)
            gid = (lid.astype(np.int64) + lo).astype(np.uint32); gid[lid == 0xFFFFFFFF] = 0xFFFFFFFF
            return D.merge_topk(D.all_gather_array(gid, cdev), D.all_gather_array(np.ascontiguousarray(d, np.float32), cdev), k)

        gt, gd = sharded(100, 0, local=lambda k, beam: ix.bruteforce_knn(Q, k))        # exact ground truth, same exchange
        D.barrier(); t0 = time.time(); ids, dists = sharded(10, args.beam); D.barrier(); ts = time.time() - t0
        out.update(n=n, shard=hi - lo, datagen_s=tgen, build_s=tb, search_s_host_inclusive=ts, qps_host_inclusive=len(Q) / ts,
                   recall_at_10=recall_at_k(ids, gt, gd, 10))
        ix.close()
    else:
        n = args.points or 10_000_000
        Xf = datasets.t2i_like(n, 200, seed=1234); Qf = datasets.t2i_like(args.queries, 200, seed=4321)
        mv = quantize.mips_i8_max_val(Xf, trim=False)
        X, Q = quantize.mips_i8_translate(Xf, mv), quantize.mips_i8_translate(Qf, mv); del Xf, Qf
        mst = 3

        def build_tree(t):          # tree t alone, seeded like tree t of the single-process build (tree_index offsets the seed)
            it = DeviceIndex(X, max_degree=mst, metric="mips", device=dev_ord)
            it.hcnng_build(1, 1000, mst, seed=1 + t)
            g = it.get_graph(); it.close()
            return g

        D.barrier(); t0 = time.time()
        G = D.hcnng_build_tree_parallel(build_tree, n, args.trees, mst, device=cdev)
        D.barrier(); tb = time.time() - t0
        ix = DeviceIndex(X, G, metric="mips", device=dev_ord)
        r = ix.batch_search(Q, k=10, beam=args.beam)
        gt, gd = ix.bruteforce_knn(Q, 100)
        out.update(n=n, trees=args.trees, build_s=tb, avg_degree=float(G[:, 0].mean()), recall_at_10=recall_at_k(r["ids"], gt, gd, 10),
                   graph_checksum=int(np.bitwise_xor.reduce(G.ravel().astype(np.uint64) * np.arange(1, G.size + 1, dtype=np.uint64))))
        ix.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    D.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
