#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats, or --pmc passes) into a small text summary
that can be committed under profiles/.  Usage: prof_summary.py <rocprof_out_dir> [kernel_substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "beam_search_kernel"
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"== {os.path.relpath(f, d)} (top kernels by total time)")
        rows = list(csv.DictReader(open(f)))
        for r in rows[:14]:
            name = r.get("Name", "")[:110]
            print(f"{name:110s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} "
                  f"pct={r.get('Percentage')}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        durs = defaultdict(list)
        meta = {}
        rd = csv.DictReader(open(f))
        print("columns:", rd.fieldnames)
        for r in rd:
            n = r.get("Kernel_Name", "")
            if pat in n:
                n = n[:70] + " grid=" + str(r.get("Grid_Size_X", r.get("Grid_Size")))
                durs[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[n] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                           r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
        print(f"== {os.path.relpath(f, d)} dispatches matching '{pat}' grouped by grid size")
        for n, v in durs.items():
            v2 = sorted(v)
            print(f"{n[:100]} n={len(v)} avg_us={sum(v) / len(v) / 1e3:.1f} med_us={v2[len(v2) // 2] / 1e3:.1f} "
                  f"min_us={v2[0] / 1e3:.1f} max_us={v2[-1] / 1e3:.1f} vgpr/agpr/sgpr/lds/grid/wg={meta[n]}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        rd = csv.DictReader(open(f))
        print("columns:", rd.fieldnames)
        for r in rd:
            n = r.get("Kernel_Name", "")
            if pat in n:
                n = n[:70] + " grid=" + str(r.get("Grid_Size_X", r.get("Grid_Size")))
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== {os.path.relpath(f, d)} counters for '{pat}' (per dispatch: mean over dispatches)")
        for n, cs in acc.items():
            print(n[:100])
            for c, v in sorted(cs.items()):
                v2 = sorted(v)
                print(f"   {c:28s} n={len(v)} mean={sum(v) / len(v):.6g} median={v2[len(v2) // 2]:.6g} max={v2[-1]:.6g}")


if __name__ == "__main__":
    main()
