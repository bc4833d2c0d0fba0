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


def sift_like_device(n, d, seed, device, dtype=np.float16, n_centers=256, rank=16, noise_scale=12.0, centers_seed=1234,
                     chunk=1 << 18):
    """`sift_like`'s mixture drawn ON THE GPU with torch's generator (plumbing: the 12.5M-point tables of bench.py's HBM-resident
    leg and sharded-index mode would take minutes of numpy on the host) and copied back as a numpy array of `dtype`.  Same
    geometry rule (cluster centres and low-rank bases from numpy's centers_seed stream, shared by base and queries), but NOT
    the same points as `sift_like` for equal seeds.  Integer-valued in [0, 255]."""
    import torch
    crng = np.random.default_rng(centers_seed)
    centers = torch.from_numpy(crng.normal(0.0, 22.0, size=(n_centers, d)).astype(np.float32)).to(device)
    bases = torch.from_numpy(crng.normal(0.0, 9.0, size=(n_centers, rank, d)).astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    tdt = {np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
    out = np.empty((n, d), dtype=dtype)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        cid = torch.randint(0, n_centers, (m,), generator=g, device=device)
        coef = torch.randn((m, rank), generator=g, device=device)
        x = centers[cid] + torch.bmm(coef.unsqueeze(1), bases[cid]).squeeze(1)
        x += torch.randn((m, d), generator=g, device=device) * noise_scale
        x = torch.clamp(torch.round(x + 100.0), 0, 255).to(tdt)
        out[s:s + m] = x.cpu().numpy()
    return out
