"""Synthetic SIFT-/DEEP-shaped data (the real files are not available offline).

Gaussian mixture with per-cluster low-rank structure.  `sift_like` emits INTEGER-valued
coordinates in [0,255], so that u8, f32 and f16 copies of the same data give bit-identical
distances in any summation order (all partial sums are integers < 2**24).
"""
import numpy as np

from .bf16 import bfloat16, to_bf16


def _cast(x, dtype):
    """astype that also knows the bfloat16 stand-in dtype"""
    return to_bf16(x) if np.dtype(dtype) == bfloat16 else x.astype(dtype)


def _mixture(n, d, seed, n_centers, rank, center_scale, basis_scale, noise_scale, centers_seed=1234):
    crng = np.random.default_rng(centers_seed)          # cluster geometry shared by base and queries
    centers = crng.normal(0.0, center_scale, size=(n_centers, d)).astype(np.float32)
    bases = crng.normal(0.0, basis_scale, size=(n_centers, rank, d)).astype(np.float32)
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    chunk = 1 << 16
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        cid = rng.integers(0, n_centers, size=m)
        coef = rng.normal(0.0, 1.0, size=(m, rank)).astype(np.float32)
        x = centers[cid] + np.einsum("mr,mrd->md", coef, bases[cid])
        x += rng.normal(0.0, noise_scale, size=(m, d)).astype(np.float32)
        out[s:s + m] = x
    return out


def sift_like(n, d=128, seed=1234, dtype=np.uint8, n_centers=256, rank=16):
    """Integer-valued SIFT-shaped vectors in [0,255] as uint8 / float32 / float16."""
    x = _mixture(n, d, seed, n_centers, rank, center_scale=22.0, basis_scale=9.0, noise_scale=12.0)
    x = np.clip(np.rint(x + 100.0), 0, 255)
    return _cast(x, dtype)


def sift1m_like(n, d=128, seed=1234, dtype=np.float16):
    """The bench workload: integer-valued SIFT-shaped vectors whose difficulty is calibrated at n = 1M
    (tools/calibrate_sift.py) so that Vamana R=64 L=128 gives recall@10 in
    [0.95, 0.99] at beam 64 with SIFT-like work per query (~75 visited, ~3.4K distance comparisons);
    `sift_like` (rank 16) is easier (recall 0.9997) and stays the generator of the unit tests."""
    x = _mixture(n, d, seed, 256, 32, center_scale=22.0, basis_scale=9.0, noise_scale=14.0)
    x = np.clip(np.rint(x + 100.0), 0, 255)
    return _cast(x, dtype)


def deep_like(n, d=96, seed=1234, n_centers=256, rank=16):
    """Real-valued unit-norm float32 vectors (DEEP-shaped)."""
    x = _mixture(n, d, seed, n_centers, rank, center_scale=1.0, basis_scale=0.5, noise_scale=0.45)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return x.astype(np.float32)


def t2i_like(n, d=200, seed=1234, n_centers=256, rank=16):
    """Real-valued float32 vectors for MIPS (Text2Image-shaped)."""
    return _mixture(n, d, seed, n_centers, rank, center_scale=0.3, basis_scale=0.15, noise_scale=0.12)


def quantize_mips_int8(x, max_val=None):
    """Quantized_Mips_Point<8>::translate_point (mips_point.h:416-430): scale = 127/max_val,
    round, clamp to +-127.  max_val = max |x| over the data (generate_parameters, trim == 0)."""
    if max_val is None:
        max_val = float(np.max(np.abs(x)))
    scale = np.float32(127.0) / np.float32(max_val)
    q = np.rint(x.astype(np.float32) * scale)
    return np.clip(q, -127, 127).astype(np.int8), max_val
