"""bfloat16 for numpy, which has no such type: a one-field structured dtype over uint16.  An array of `bfloat16` is two bytes per
element like the device rows, is told apart from a plain uint16 array by its dtype (so the C-ABI gets PANN_BF16), and converts
with `to_bf16` / `from_bf16`.  (PANN_BF16 is this build's extension; the reference has no two-byte float point type.)"""
import numpy as np

bfloat16 = np.dtype([("bf16", np.uint16)])


def to_bf16(x):
    """float array -> bfloat16 array, round to nearest even (integers of magnitude <= 256 are exact)"""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) >> 16).astype(np.uint16)
    nan = np.isnan(f)
    if nan.any():
        out[nan] = 0x7FC0
    return out.view(bfloat16).reshape(f.shape)


def from_bf16(a):
    """bfloat16 array -> float32 (exact)"""
    u = np.ascontiguousarray(a).view(np.uint16).astype(np.uint32) << 16
    return u.view(np.float32).reshape(np.shape(a))
