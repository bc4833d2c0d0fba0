#!/usr/bin/env python3
"""Build-time guard for the kernels that issue loads / LDS reads by hand: the register that receives a hand-issued
load is invisible to hipcc's s_waitcnt pass, so it must never be spilled or copied before the hand-written wait -- the kernels
concerned must have NO scratch and no VGPR spills, and the beam-64 kernel must stay within its 72-VGPR launch bound.
Reads the AMDGPU metadata notes of the gfx950 code objects inside the built objects (fast: no recompilation).
usage: tools/check_kernel_regs.py [build dir]      (default parlayann_amd/csrc/build); exit status 1 on a violation"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = {   # object -> [(kernel name pattern, max vgprs or None)]
    "beam_search.o": [(r"beam_search_b64_kernel", 72), (r"beam_search_b128_kernel", None)],
    "leaf_knn.o": [(r"leaf_knn_kernel", None)],
    "dense.o": [(r"dense_gt_mfma_kernel", None)],
}


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}", "--unbundle"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
        name = g("name")
        if name:
            out.append({"name": name.group(1), "vgpr": int(g("vgpr_count").group(1)), "scratch": int(g("private_segment_fixed_size").group(1)),
                        "vgpr_spill": int(g("vgpr_spill_count").group(1))})
    return out


def main():
    bdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "parlayann_amd", "csrc", "build")
    bad = checked = 0
    for obj, rules in RULES.items():
        ks = kernels_of(os.path.join(bdir, obj))
        for pat, vmax in rules:
            for k in ks:
                if not re.search(pat, k["name"]):
                    continue
                checked += 1
                if k["scratch"] or k["vgpr_spill"] or (vmax and k["vgpr"] > vmax):
                    bad += 1
                    print(f"VIOLATION {k['name']}: vgpr {k['vgpr']} (max {vmax}) scratch {k['scratch']} spills {k['vgpr_spill']}")
    print(f"check_kernel_regs: {checked} kernels checked, {bad} violations")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main())
