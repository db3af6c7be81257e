"""Shared helpers for the parity tests (seeded synthetic inputs, comparisons)."""
import numpy as np


def tiles(seed, n, h, w, c=1):
    return np.random.default_rng(seed).standard_normal((n, h, w, c)).astype(np.float32)


def rand_weights(seed, shape, scale=None):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal(shape).astype(np.float32)
    if scale is None:
        scale = 1.0 / np.sqrt(max(1, int(np.prod(shape[:-1]))))
    return (w * np.float32(scale)).astype(np.float32)


def assert_bit_exact(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    same = a.view(np.uint32 if a.dtype == np.float32 else a.dtype) == \
        b.view(np.uint32 if b.dtype == np.float32 else b.dtype)
    if not same.all():
        # +0.0 / -0.0 compare equal as floats but differ in bits: tolerate only that
        if a.dtype == np.float32 and np.array_equal(a, b):
            return
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d elements differ; first at %s: %r vs %r (max abs %g)" % (
            what, len(bad), a.size, i, a[i], b[i], float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))))
