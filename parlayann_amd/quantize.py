"""Scalar quantisation of float points, as the reference's translating PointRange constructor does
(point_range.h:54-72) -- host-side preprocessing, numpy float32 arithmetic in the reference's order.

  Euclidian_Point<uint8_t>  generate_parameters  euclidian_point.h:211-235
                            translate_point      euclidian_point.h:182-209
  Quantized_Mips_Point<8>   generate_parameters  mips_point.h:433-486  (trim: 1e-4 quantiles)
                            translate_point      mips_point.h:416-430
  Mips_Point<T>::normalize                       mips_point.h:113-122
"""
import numpy as np

F = np.float32


class EuclidParams:
    """Euclidian_Point<uint8_t>::parameters (euclidian_point.h:100-110): slope = range/(max-min),
    offset = (int32) round(min*slope)."""

    def __init__(self, min_val, max_val, dims, rng=255):
        self.range = rng
        self.slope = F(rng) / (F(max_val) - F(min_val))
        self.offset = np.int32(_round_half_away(F(min_val) * self.slope))
        self.dims = dims

    @property
    def identity(self):
        return self.slope == F(1.0) and self.offset == 0


def euclid_u8_params(x):
    x = np.asarray(x, dtype=F)
    min_val = F(min(0.0, float(x.min())))          # mins/maxs start at 0 (:217-218)
    max_val = F(max(0.0, float(x.max())))
    all_ints = bool(np.all(x >= 0) and np.all(x == np.trunc(x)))
    if all_ints:
        if max_val < 256:
            max_val = F(255)
        min_val = F(0)
    return EuclidParams(min_val, max_val, x.shape[1])


def _round_half_away(v):
    """std::round: halves away from zero (numpy's round is half-to-even)."""
    v = np.asarray(v, dtype=F)
    return np.where(v >= 0, np.floor(v + F(0.5)), np.ceil(v - F(0.5))).astype(F)


def euclid_u8_translate(x, p):
    x = np.asarray(x, dtype=F)
    if p.identity:
        return x.astype(np.uint8)
    r = _round_half_away(x * p.slope).astype(np.int64) - np.int64(p.offset)
    return np.clip(r, 0, p.range).astype(np.uint8)


def normalize_rows(x):
    """Mips_Point::normalize: norm accumulated in double over float products, inv_norm in float."""
    x = np.asarray(x, dtype=F)
    norm = np.sqrt(np.sum((x * x).astype(np.float64), axis=1))
    norm[norm == 0] = 1.0
    inv = (1.0 / norm).astype(F)
    return (x * inv[:, None]).astype(F)


def mips_i8_max_val(x, trim=True):
    x = np.asarray(x, dtype=F).ravel()
    n = x.size
    if trim:
        lo_i = int(F(0.0001) * F(n))                        # (long)(cutoff * len), float arithmetic
        hi_i = int((1.0 - float(F(0.0001))) * (n - 1))      # (long)((1.0 - cutoff) * (len-1)), double
        part = np.partition(x, [lo_i, hi_i])
        min_val, max_val = part[lo_i], part[hi_i]
    else:
        min_val, max_val = x.min(), x.max()
    return F(max(float(max_val), -float(min_val)))


def mips_i8_translate(x, max_val, rng=255):
    x = np.asarray(x, dtype=F)
    half = rng // 2                                         # integer 127
    scale = F(half) / F(max_val)
    v = _round_half_away(x * scale)
    v = np.where(x < -F(max_val), -half, np.where(x > F(max_val), half, v))
    return v.astype(np.int8)
