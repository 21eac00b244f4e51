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


def _toy_sampler_desc(rng):
    """Two providers (truncate_start / random) over ragged lengths, as numpy arrays + the ctypes descriptor."""
    lens = [rng.integers(150, 400, size=37).astype(np.int32), rng.integers(150, 400, size=53).astype(np.int32)]
    arrs = dict(sw=np.array([2.0, 10.0]), strat=np.array([native.STRATEGIES["truncate_start"], native.STRATEGIES["random"]], np.int32),
                offs=np.array([0, 37, 90], np.int64), st=np.concatenate([np.zeros(37, np.int32), np.ones(53, np.int32)]),
                src=np.concatenate([np.cumsum(lens[0]) - lens[0], np.cumsum(lens[1]) - lens[1]]).astype(np.int64) * 40,
                ln=np.concatenate(lens), coffs=np.zeros(3, np.int32), cuts=np.zeros(1, np.int32))
    d = native.SamplerDesc()
    d.n_providers = 2
    d.sampling_weight = arrs["sw"].ctypes.data_as(C.POINTER(C.c_double))
    d.strategy = arrs["strat"].ctypes.data_as(C.POINTER(C.c_int32))
    d.set_offsets = arrs["offs"].ctypes.data_as(C.POINTER(C.c_int64))
    d.set_store = arrs["st"].ctypes.data_as(C.POINTER(C.c_int32))
    d.set_src_elem = arrs["src"].ctypes.data_as(C.POINTER(C.c_int64))
    d.set_len = arrs["ln"].ctypes.data_as(C.POINTER(C.c_int32))
    d.cutoff_offsets = arrs["coffs"].ctypes.data_as(C.POINTER(C.c_int32))
    d.cutoffs = arrs["cuts"].ctypes.data_as(C.POINTER(C.c_int32))
    return d, arrs


def _mt_states(seed):
    random.seed(seed)
    np.random.seed(seed)
    py = np.array(random.getstate()[1], np.uint32)
    s = np.random.get_state()
    npst = np.empty(625, np.uint32)
    npst[:624], npst[624] = s[1], s[2]
    return py, npst


def test_prefetched_batches_are_the_synchronous_sampler_s_batches(nl):
    """The worker thread of mww_prefetcher hands out exactly the sequence of batches mww_sample_training_batch draws from
    the same stream states (host-only: no device involved), reports the stream positions of every batch boundary, and a
    second prefetcher started from such a boundary continues the sequence."""
    d, arrs = _toy_sampler_desc(np.random.default_rng(5))
    B, T, pol = 48, 194, (5, 2, 5, 2)
    nm = pol[1] + pol[3]
    labels, weights = np.array([1.0, 0.0], np.float32), np.array([0.5, 2.0], np.float32)

    py, npst = _mt_states(11)
    sync = []
    win, masks = np.zeros(B, native.WINDOW_DTYPE), np.zeros((B, nm, 2), np.int32)
    prov, samp, order = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    for _ in range(7):
        nl.check(nl.lib.mww_sample_training_batch(C.byref(d), py.ctypes.data_as(C.c_void_p), npst.ctypes.data_as(C.c_void_p), B, T, pol[0], pol[1],
                                                  pol[2], pol[3], -1, 1, win.ctypes.data_as(C.c_void_p), masks.ctypes.data_as(C.c_void_p),
                                                  prov.ctypes.data_as(C.c_void_p), samp.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p)))
        sync.append(dict(windows=win.copy(), masks=masks.copy(), provider=prov.copy(), sample=samp.copy(), py=py.copy(), np=npst.copy()))

    py0, np0 = _mt_states(11)
    pf = native.Prefetcher(nl, d, labels, weights, py0, np0, B, T, pol[0], pol[1], pol[2], pol[3], -1, depth=3)
    assert pf.rng_state()[2] == 0 and np.array_equal(pf.rng_state()[0], py0)
    for k in range(4):
        got = pf.acquire()
        for key in ("windows", "masks", "provider", "sample"):
            assert np.array_equal(got[key], sync[k][key]), (k, key)
        assert np.array_equal(got["labels"], labels[sync[k]["provider"]]) and np.array_equal(got["weights"], weights[sync[k]["provider"]])
        p1, n1, handed = pf.rng_state()
        assert handed == k + 1 and np.array_equal(p1, sync[k]["py"]) and np.array_equal(n1, sync[k]["np"])
    p1, n1, _ = pf.rng_state()   # the worker is up to three batches past this point; those draws are discarded with it
    pf.close()
    pf2 = native.Prefetcher(nl, d, labels, weights, p1, n1, B, T, pol[0], pol[1], pol[2], pol[3], -1, depth=1)
    for k in range(4, 7):
        got = pf2.acquire()
        for key in ("windows", "masks", "provider", "sample"):
            assert np.array_equal(got[key], sync[k][key]), (k, key)
    pf2.close()


def test_prefetcher_reports_a_sampler_error_instead_of_a_batch(nl):
    d, arrs = _toy_sampler_desc(np.random.default_rng(6))
    arrs["strat"][:] = native.STRATEGIES["none"]     # cannot form fixed-length windows from longer samples
    py, npst = _mt_states(3)
    pf = native.Prefetcher(nl, d, np.zeros(2, np.float32), np.ones(2, np.float32), py, npst, 8, 100, 0, 0, 0, 0, -1, depth=2)
    with pytest.raises(native.NativeError):
        pf.acquire()
    with pytest.raises(native.NativeError):
        pf.acquire()
    pf.close()
