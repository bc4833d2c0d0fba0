#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_summary_*.txt (tools/profile_bench.sh <tag> ...) into the committed profiles/<tag>_kernel_stats.txt,
profiles/<tag>_pmc.txt and an entry of profiles/traffic_latest.json (the counter-measured HBM bytes of one launch that bench.py
quotes, labelled with its source, for exactly this workload).
usage: collect_profiles.py <tag> [--n N] [--nq NQ] [--dtype f16] [--d 128] [--R 64] [--beam 64] [--data sift1m_like] [--note TEXT]"""
import argparse
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--dtype", default="f16"); ap.add_argument("--d", type=int, default=128); ap.add_argument("--R", type=int, default=64)
ap.add_argument("--beam", type=int, default=64); ap.add_argument("--data", default="sift1m_like"); ap.add_argument("--note", default="")
ap.add_argument("--bench-args", default=None, help="the bench.py arguments of the profiled runs, for the header lines (default: derived from the key)")
a = ap.parse_args()
tag, grid = a.tag, f"grid={a.nq * 64}"
bargs = a.bench_args or f"--n {a.n} --nq {a.nq} --dtype {a.dtype} --d {a.d} --R {a.R} --beam {a.beam} --data {a.data}"
s = open(f"gpurun_out/{tag}_summary_stats.txt").read().splitlines()
out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline {bargs} --steps 20   (MI355X, {tag}) {a.note}",
       "# whole process: includes the Vamana build of the index (its searches run beam_search_b128_kernel, L=128) and the brute-force ground truth"]
out += [l[:200] for l in s[:16] if not l.startswith("columns")]
out += ["", f"# dispatches grouped by grid size; {grid} ({a.nq} waves x 64) are the TIMED query steps (3 warmup + 20)"]
out += [l for l in s if grid in l]
open(f"profiles/{tag}_kernel_stats.txt", "w").write("\n".join(out) + "\n")
pm = [f"# rocprofv3 --pmc <counters> (one pass per group; never combined with --kernel-trace/--stats) -- python3 bench.py --no-cpu-baseline {bargs} --steps 5",
      f"# per dispatch of the query step ({grid}), mean over dispatches; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them"]
vals = {}
for f in ("fetch", "write", "sq", "tcc"):
    path = f"gpurun_out/{tag}_summary_{f}.txt"
    if not os.path.exists(path):
        continue
    take = False
    for l in open(path).read().splitlines():
        if "grid=" in l:
            take = grid in l
        if take:
            pm.append(l)
            m = re.match(r"\s+(\S+)\s+n=\d+ mean=(\S+)", l)
            if m:
                vals[m.group(1)] = float(m.group(2))
open(f"profiles/{tag}_pmc.txt", "w").write("\n".join(pm) + "\n")
bench = json.loads(open(f"gpurun_out/{tag}_stats.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/{tag}.json", "w"))
hbm = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
ent = {"n": a.n, "nq": a.nq, "beam": a.beam, "dtype": a.dtype, "d": a.d, "R": a.R, "data": a.data, "hbm_bytes_per_launch": hbm,
       "source": f"profiles/{tag}_pmc.txt",
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes: (2*FETCH_SIZE + WRITE_SIZE)*1024; the factor 2 is the "
              "gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B); FETCH_SIZE counts Infinity-Cache hits too"}
if "TCC_HIT_sum" in vals:
    ent["l2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
tp = "profiles/traffic_latest.json"
cur = json.load(open(tp)) if os.path.exists(tp) else []
cur = [e for e in (cur if isinstance(cur, list) else [cur])
       if not all(e.get(k) == ent[k] for k in ("n", "nq", "beam", "dtype", "d", "R", "data"))] + [ent]
json.dump(cur, open(tp, "w"), indent=1)
alg = bench["roofline"]["algorithmic_bytes_per_launch"]; ms = bench["roofline"]["kernel_ms"]
print(f"{tag}: hbm bytes per launch {hbm} = {hbm / alg:.3f} x algorithmic; kernel {ms:.3f} ms under the profiler; algorithmic "
      f"{alg / ms / 1e6:.0f} GB/s, HBM-side {hbm / ms / 1e6:.0f} GB/s", vals)
