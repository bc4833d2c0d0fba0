#!/usr/bin/env python3
"""Turn gpurun_out/summary_*.txt (tools/profile_bench.sh) into the committed profiles/<tag>_* files and
profiles/traffic_latest.json.  usage: collect_profiles.py <tag> [nq=10000]"""
import json
import re
import sys

tag = sys.argv[1]
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
grid = f"grid={nq * 64}"
s = open("gpurun_out/summary_stats.txt").read().splitlines()
out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --no-cpu-baseline   (MI355X, {tag})",
       "# whole process: includes the 1M-point Vamana build (its searches run beam_search_b128_kernel, L=128) and the brute-force ground truth"]
out += [l[:200] for l in s[:16] if not l.startswith("columns")]
out += ["", f"# dispatches grouped by grid size; {grid} ({nq} waves x 64) are the TIMED query steps (3 warmup + 20)"]
out += [l for l in s if grid in l]
open(f"profiles/{tag}_bench1m_kernel_stats.txt", "w").write("\n".join(out) + "\n")
pm = ["# rocprofv3 --pmc <counters> (one pass per group; never combined with --kernel-trace/--stats) -- python3 bench.py --steps 5 --no-cpu-baseline",
      f"# per dispatch of the query step ({grid}), mean over dispatches; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them"]
vals = {}
for f in ("fetch", "write", "sq", "tcc"):
    take = False
    for l in open(f"gpurun_out/summary_{f}.txt").read().splitlines():
        if "grid=" in l:
            take = grid in l
        if take:
            pm.append(l)
            m = re.match(r"\s+(\S+)\s+n=\d+ mean=(\S+)", l)
            if m:
                vals[m.group(1)] = float(m.group(2))
open(f"profiles/{tag}_bench1m_pmc.txt", "w").write("\n".join(pm) + "\n")
hbm = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
json.dump({"n": 1000000, "nq": nq, "beam": 64, "hbm_bytes_per_launch": hbm,
           "how": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{tag}_bench1m_pmc.txt): "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024; the factor 2 is the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE "
                  f"tallies 128-B requests at 64 B); cross-check (TCC_HIT+TCC_MISS)*128 B = {(vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']) * 128 / 1e9:.2f} GB"},
          open("profiles/traffic_latest.json", "w"))
print("hbm bytes per launch", hbm, vals)
