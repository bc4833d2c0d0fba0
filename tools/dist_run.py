#!/usr/bin/env python3
"""Multi-GPU paths of SURVEY.md section 8(e) as a runnable driver, one process per GPU, nothing on the host between the
kernels and the collectives (device tensors; under RCCL they never leave the GPUs):

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
      tools/dist_run.py c4 --points 100000000          # sharded index: rank r owns ids [r*n/8, (r+1)*n/8)
  ... tools/dist_run.py c3 --points 10000000           # Vamana build: every batch split over the ranks, one all-gather per batch
  ... tools/dist_run.py c5 --points 10000000           # HCNNG: rank r builds trees r, r+W, ...; one all-gather of the slabs

c4: every rank generates and builds ONLY its own slice (own sub-graph, local ids); every query goes to every rank; the per-rank
    top-k lists (k*8 bytes per query) are all-gathered and merged on the device (pann_merge_topk_dev).  No other collective.
c3: points and graph replicated; distributed.device_vamana_build_sharded; the graph equals the single-GPU build (checksum printed).
c5: points replicated on ONE resident index per rank; distributed.device_hcnng_build_tree_parallel; same graph as one GPU.
PANN_DIST_BACKEND=gloo rehearses on a box with fewer GPUs than ranks (ranks wrap around the visible devices; device tensors
are then staged through the host inside all_gather_tensor)."""
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
    ap.add_argument("mode", choices=["c3", "c4", "c5"])
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
    def checksum(G):
        return int(np.bitwise_xor.reduce(G.ravel().astype(np.uint64) * np.arange(1, G.size + 1, dtype=np.uint64)))

    dev = torch.device("cuda", dev_ord)
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
        tb = [0.0]

        def build(ix):
            t0 = time.time(); ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb[0] = time.time() - t0

        sh = D.DeviceShardedIndex(X, 64, build, device_ordinal=dev_ord, n_total=n)
        D.barrier()
        d_q = torch.from_numpy(Q.view(np.uint8).reshape(len(Q), -1)).to(dev)
        gt_i, gt_d = sh.bruteforce(Q, 100)                                  # exact ground truth, same exchange
        sh.search(d_q, 10, args.beam); torch.cuda.synchronize(dev)          # warm-up (workspace allocation)
        D.barrier(); t0 = time.time()
        oi, od = sh.search(d_q, 10, args.beam)
        torch.cuda.synchronize(dev); D.barrier(); ts = time.time() - t0
        out.update(n=n, shard=hi - lo, datagen_s=tgen, build_s=tb[0], search_s_device_resident=ts, qps_device_resident=len(Q) / ts,
                   recall_at_10=recall_at_k(oi.cpu().numpy().view(np.uint32), gt_i.cpu().numpy().view(np.uint32), gt_d.cpu().numpy(), 10))
        sh.close()
    elif args.mode == "c3":
        n = args.points or 10_000_000
        X = datasets.deep_like(n, 96, seed=1234); Q = datasets.deep_like(args.queries, 96, seed=4321)
        ix = DeviceIndex(X, max_degree=64, device=dev_ord)
        D.barrier(); t0 = time.time()
        info = D.device_vamana_build_sharded(ix, 64, 128, 1.05, num_passes=2, seed=1)
        D.barrier(); tb = time.time() - t0
        st = info["stats"]
        r = ix.batch_search(Q, k=10, beam=args.beam)
        gt, gd = ix.bruteforce_knn(Q, 100)
        out.update(n=n, build_s=tb, collectives=info["collectives"], bytes_gathered=info["bytes_gathered"],
                   rank0_phases_s={"search": st.t_search_s, "prune": st.t_prune_s, "bidirect": st.t_bidirect_s, "reprune": st.t_reprune_s},
                   recall_at_10=recall_at_k(r["ids"], gt, gd, 10), graph_checksum=checksum(ix.get_graph()))
        ix.close()
    else:
        n = args.points or 10_000_000
        Xf = datasets.t2i_like(n, 200, seed=1234); Qf = datasets.t2i_like(args.queries, 200, seed=4321)
        mv = quantize.mips_i8_max_val(Xf, trim=False)
        X, Q = quantize.mips_i8_translate(Xf, mv), quantize.mips_i8_translate(Qf, mv); del Xf, Qf
        mst = 3
        ix = DeviceIndex(X, max_degree=args.trees * mst, metric="mips", device=dev_ord)       # ONE resident index per rank
        D.barrier(); t0 = time.time()
        info = D.device_hcnng_build_tree_parallel(ix, args.trees, 1000, mst, seed=1)
        D.barrier(); tb = time.time() - t0
        G = ix.get_graph()
        r = ix.batch_search(Q, k=10, beam=args.beam)
        gt, gd = ix.bruteforce_knn(Q, 100)
        out.update(n=n, trees=args.trees, build_s=tb, bytes_gathered=info["bytes_gathered"], avg_degree=float(G[:, 0].mean()),
                   recall_at_10=recall_at_k(r["ids"], gt, gd, 10), graph_checksum=checksum(G))
        ix.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    D.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
