"""numpy quantisation (parlayann_amd/quantize.py) against the oracle's C++ restatement of
euclidian_point.h:182-235 and mips_point.h:113-122,416-486 -- CPU only."""
import numpy as np

from parlayann_amd import datasets, quantize


def test_euclid_u8_real_valued(oracle):
    X = datasets.deep_like(3000, 96, seed=1) * 3.0 - 0.7
    p = quantize.euclid_u8_params(X)
    slope, offset = oracle.euclid_u8_params(X)
    assert p.slope == slope and int(p.offset) == offset and not p.identity
    np.testing.assert_array_equal(quantize.euclid_u8_translate(X, p), oracle.euclid_u8_translate(X, slope, offset))


def test_euclid_u8_integer_valued_is_plain_cast(oracle):
    X = datasets.sift_like(2000, 64, seed=1, dtype=np.float32)
    p = quantize.euclid_u8_params(X)
    slope, offset = oracle.euclid_u8_params(X)
    assert p.identity and slope == 1.0 and offset == 0          # max < 256, all non-negative ints (:226-230)
    np.testing.assert_array_equal(quantize.euclid_u8_translate(X, p), X.astype(np.uint8))
    np.testing.assert_array_equal(oracle.euclid_u8_translate(X, slope, offset), X.astype(np.uint8))


def test_mips_int8(oracle):
    X = datasets.t2i_like(4000, 200, seed=1)
    Xn = quantize.normalize_rows(X)
    np.testing.assert_array_equal(Xn, oracle.normalize(X))
    for trim in (True, False):
        mv = quantize.mips_i8_max_val(Xn, trim=trim)
        assert mv == oracle.mips_i8_maxval(Xn, trim=trim)
        np.testing.assert_array_equal(quantize.mips_i8_translate(Xn, mv), oracle.mips_i8_translate(Xn, mv))
    # halves round away from zero (std::round), values beyond +-max_val saturate at +-127
    mv = np.float32(1.0)
    x = np.array([[0.5 / 127, -0.5 / 127, 2.0, -2.0, 1.0, 2.5 / 127]], np.float32)
    np.testing.assert_array_equal(quantize.mips_i8_translate(x, mv), oracle.mips_i8_translate(x, mv))
