"""CPU-only: the C-ABI library loads and exports every symbol include/pann.h declares, the ctypes
table covers exactly those symbols, and -- with no GPU -- compute entry points fail loudly
instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pann.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pann_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from parlayann_amd import _capi
    names = _declared()
    assert len(names) >= 20
    lib = C.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pann.h but not exported by libpann.so"
    assert sorted(_capi.SIGNATURES) == names
    assert _capi.load().pann_abi_version() == 3


def test_struct_layouts_match_header():
    from parlayann_amd import _capi
    assert C.sizeof(_capi.QueryParams) == 48          # 4x int64/double + int64 + int32 + float
    assert _capi.QueryParams.limit.offset == 24 and _capi.QueryParams.rerank_factor.offset == 40
    assert C.sizeof(_capi.SearchOut) == 88
    assert C.sizeof(_capi.BuildStats) == 72


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from parlayann_amd import DeviceIndex, PannError
    X = np.zeros((16, 8), np.uint8)
    with pytest.raises(PannError) as e:
        DeviceIndex(X, max_degree=4)
    assert e.value.code == 3 and "no HIP device" in str(e.value)     # PANN_ERR_NO_DEVICE


def test_streaming_loader_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from parlayann_amd import DeviceIndex, PannError, io
    io.write_bin(tmp_path / "b.bin", np.zeros((16, 8), np.uint8))
    with pytest.raises(PannError) as e:
        DeviceIndex.from_files(tmp_path / "b.bin", np.uint8, max_degree=4)
    assert e.value.code == 3
    with pytest.raises(ValueError):
        DeviceIndex.from_files(tmp_path / "b.bin", np.uint8, max_degree=4, rows=(4, 40))     # beyond the file: checked on the host


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "parlayann_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if "oracle_api" in txt or "libpann_oracle" in txt or "pann_oracle_" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    b = open(os.path.join(ROOT, "bench.py")).read()
    assert b.count("import oracle_api") == 1 and b.index("import oracle_api") > b.index("def cpu_baseline")


def test_host_mirror_compiles_without_a_gpu(tmp_path):
    import subprocess
    host = os.path.join(ROOT, "parlayann_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    assert os.path.exists(os.path.join(host, "vamana", "neighbors")) and os.path.exists(os.path.join(host, "HCNNG", "neighbors"))


def test_reference_signature_tu_compiles_without_a_gpu():
    """tests/host_api_check.cpp calls the host mirror with the reference's argument lists; it must at least compile and link
    against the C-ABI library here (it runs under -m gpu, tests/test_host_api_gpu.py)"""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host_api_check")
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-pthread", "-Wall", "-Wno-sign-compare", "-o", exe,
                           os.path.join(ROOT, "tests", "host_api_check.cpp"), "-L" + os.path.join(ROOT, "parlayann_amd", "lib"),
                           "-lpann", "-Wl,-rpath," + os.path.join(ROOT, "parlayann_amd", "lib")])
    assert os.path.exists(exe)


def test_kernels_with_hand_issued_loads_do_not_spill():
    """a register that receives a hand-issued load (beam-64 row prefetch, leaf-kNN LDS pipeline) must never be spilled
    or copied before its hand-written wait; the built code objects are checked for scratch / VGPR spills / the 72-VGPR bound"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernel_regs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
