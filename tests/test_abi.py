"""CPU: the C-ABI library loads and exports every symbol include/sequitr_hip.h declares
(no compute calls -- there is no GPU here), and the host-side checks fail loudly."""
import os
import re

import numpy as np
import pytest
import torch

from sequitr_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sequitr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sq_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_bound_and_exported():
    names = _declared_symbols()
    assert len(names) >= 13
    lib = _lib.load()
    for n in names:
        assert n in _lib.SIGNATURES, "header declares %s but _lib.SIGNATURES does not bind it" % n
        assert hasattr(lib, n), "libsequitr_hip.so does not export %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.sq_version() >= 100


def test_host_side_validation_needs_no_gpu():
    lib = _lib.load()
    # NULL pointers are refused before any launch, with a message
    rc = lib.sq_conv2d_nhwc_fwd_f32(None, None, None, None, 1, 16, 16, 16, 16, 3, 1.0, 1, None)
    assert rc == -1 and b"null" in lib.sq_last_error()
    rc = lib.sq_maxpool2x2_fwd_f32(16, 32, 1, 15, 16, 16, None)
    assert rc == -1 and b"even" in lib.sq_last_error()
    rc = lib.sq_convT2x2s2_nhwc_fwd_f32(16, 16, None, None, 16, 1, 4, 4, 12, 16, 0, None)
    assert rc == -1 and b"multiple of 16" in lib.sq_last_error()
    assert lib.sq_wsoftmax_ce_partials(10 ** 9) == 2048 and lib.sq_wsoftmax_ce_partials(100) == 1


def test_no_cpu_fallback():
    x = torch.zeros(1, 16, 16, 16)
    w = torch.zeros(3, 3, 16, 16)
    with pytest.raises(_lib.SequitrHipError):
        ops.conv2d(x, w)


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsequitr_hip.so")
    with pytest.raises(_lib.SequitrHipError):
        _lib.load()
