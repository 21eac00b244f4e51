"""CPU-side checks of the shared library: it loads without a GPU, exports every symbol that
include/mww.h declares, and the host-only sampler primitives reproduce CPython / numpy streams."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

from microwakeword_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nl():
    if not os.path.isfile(native.DEFAULT_LIB):
        import __graft_entry__ as g
        g.build()
    return native.NativeLib.get()


def test_every_declared_symbol_is_exported(nl):
    hdr = open(os.path.join(ROOT, "include", "mww.h")).read()
    declared = set(re.findall(r"\b(mww_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "header parse failed"
    missing = [s for s in sorted(declared) if not hasattr(nl.lib, s)]
    assert not missing, missing
    assert set(native.EXPORTS) == declared


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(native.NativeError):
        native.NativeLib(str(tmp_path / "nope.so"))


def test_no_device_is_an_error_not_a_fallback(nl):
    if nl.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(native.NativeError):
        native.Engine(194, 32, 3, 1, [48, 48, 48, 48], [5, 9, 13, 21], 8, lib=nl)


def test_mt19937_streams_match_cpython_and_numpy(nl):
    random.seed(12345)
    st = np.array(random.getstate()[1], np.uint32)
    out = np.zeros(50, np.float64)
    nl.lib.mww_rng_selftest(st.ctypes.data_as(C.c_void_p), 0, 50, out.ctypes.data_as(C.c_void_p), None, 0)
    assert out.tolist() == [random.random() for _ in range(50)]
    for bound in (1, 2, 7, 40, 195, 4096, 100000):
        random.seed(bound)
        st = np.array(random.getstate()[1], np.uint32)
        oi = np.zeros(64, np.uint32)
        nl.lib.mww_rng_selftest(st.ctypes.data_as(C.c_void_p), 1, 64, None, oi.ctypes.data_as(C.c_void_p), bound)
        assert oi.tolist() == [random.randrange(bound) for _ in range(64)]
    for high in (2, 3, 17, 207, 5000):
        np.random.seed(high)
        s = np.random.get_state()
        st = np.empty(625, np.uint32)
        st[:624], st[624] = s[1], s[2]
        oi = np.zeros(64, np.uint32)
        nl.lib.mww_rng_selftest(st.ctypes.data_as(C.c_void_p), 2, 64, None, oi.ctypes.data_as(C.c_void_p), high - 1)
        assert oi.tolist() == [int(np.random.randint(0, high)) for _ in range(64)]
    np.random.seed(9)
    s = np.random.get_state()
    st = np.empty(625, np.uint32)
    st[:624], st[624] = s[1], s[2]
    out = np.zeros(20, np.float64)
    nl.lib.mww_rng_selftest(st.ctypes.data_as(C.c_void_p), 0, 20, out.ctypes.data_as(C.c_void_p), None, 0)
    assert [int(5 * v) for v in out] == [int(np.random.uniform(0, 5)) for _ in range(20)]
