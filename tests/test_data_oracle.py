"""oracle/data_oracle.py pinned against (a) golden vectors produced by the reference's own
data.py and (b) the reference module itself when /root/reference is present."""
import os
import random

import numpy as np
import pytest

from oracle import data_oracle as do

T = 194
POLICY = dict(freq_mix_prob=0.0, time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)
SCALE = np.float32(0.0390625)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "data_golden.npz"))


def stores_from_golden(gold, tag):
    def grab(prov, mode):
        out, i = [], 0
        while "%s/in/%s/%s/%d" % (tag, prov, mode, i) in gold:
            out.append(gold["%s/in/%s/%s/%d" % (tag, prov, mode, i)])
            i += 1
        return [out] if out else []

    return {p: {m: grab(p, m) for m in do.MODES} for p in ("pos", "neg", "cut")}


def build_providers(gold, tag):
    st = stores_from_golden(gold, tag)
    random.seed(3)
    np.random.seed(3)
    return [
        do.index_provider(st["pos"], True, 2.0, 1.0, "truncate_start", 1, 0.01),
        do.index_provider(st["neg"], False, 10.0, 1.5, "random", 1, 0.01),
        do.index_provider(st["cut"], False, 3.0, 0.5, "fixed_right_cutoff", 1, 0.01, [0, 5, 11]),
    ]


@pytest.mark.parametrize("tag", ["u16", "f32"])
def test_get_data_matches_reference_golden(gold, tag):
    provs = build_providers(gold, tag)
    for call in range(2):
        x, y, w, descs, order = do.get_data(provs, "training", 16, T, "default", POLICY)
        assert x.dtype == np.float32 and x.shape == (16, T, 40)
        np.testing.assert_array_equal(x, gold["%s/train%d/xc" % (tag, call)].astype(np.float32) * SCALE)
        np.testing.assert_array_equal(y, gold["%s/train%d/y" % (tag, call)])
        np.testing.assert_array_equal(w, gold["%s/train%d/w" % (tag, call)])
    x, y, w, _, _ = do.get_data(provs, "validation", 16, T, "truncate_start")
    np.testing.assert_array_equal(x, gold[tag + "/val/xc"].astype(np.float32) * SCALE)
    np.testing.assert_array_equal(y, gold[tag + "/val/y"])
    np.testing.assert_array_equal(w, gold[tag + "/val/w"])
    x, y, w, _, _ = do.get_data(provs, "validation_ambient", 16, T, "split")
    np.testing.assert_array_equal(x, gold[tag + "/amb/xc"].astype(np.float32) * SCALE)
    np.testing.assert_array_equal(y, gold[tag + "/amb/y"])
    sizes = [sum(p.mode_size(m) for p in provs) for m in ("training", "validation", "validation_ambient")]
    np.testing.assert_array_equal(sizes, gold[tag + "/sizes"])
    dur = [sum(p.stats[m]["total_duration"] for p in provs) for m in ("training", "validation", "validation_ambient")]
    np.testing.assert_allclose(dur, gold[tag + "/durations"], rtol=0, atol=0)


def test_known_answer_spec_augment(gold):
    random.seed(0)
    np.random.seed(0)
    tm, fm = do.draw_masks(T, 40, 5, 2, 5, 2)
    d = do.WindowDesc(0, 0, 0, 0, T, 0, tm, fm)
    out = do.materialise(np.ones((T, 40), np.float32), d, T)
    np.testing.assert_array_equal(out, gold["ka/spec_augment_ones"])
    # rows {98,99,107,108,109}, columns {2,3,4,16,17}
    assert sorted(np.where(out.sum(1) == 0)[0]) == [98, 99, 107, 108, 109]
    assert sorted(np.where(out.sum(0) == 0)[0]) == [2, 3, 4, 16, 17]


def test_fixed_length_strategies(gold):
    base = np.arange(300 * 40, dtype=np.float32).reshape(300, 40)
    np.random.seed(11)
    for strat in ("random", "truncate_start", "truncate_end", "fixed_right_cutoff", "none"):
        off, cp, pad = do.window_offset(300, T, strat, 7)
        got = do.materialise(base, do.WindowDesc(0, 0, 0, off, cp, pad), T)
        np.testing.assert_array_equal(got, gold["ka/fls_long_" + strat])
    off, cp, pad = do.window_offset(100, T, "random", 0)
    assert (off, cp, pad) == (0, 100, 94)  # left pad, no RNG draw
    np.testing.assert_array_equal(do.materialise(base[:100], do.WindowDesc(0, 0, 0, off, cp, pad), T), gold["ka/fls_short"])
    off, cp, pad = do.window_offset(T, T, "random", 0)
    np.testing.assert_array_equal(do.materialise(base[:T], do.WindowDesc(0, 0, 0, off, cp, pad), T), gold["ka/fls_equal"])


def test_random_offset_never_picks_last_window():
    # np.random.randint(0, L-T) has an exclusive high
    np.random.seed(0)
    offs = {do.window_offset(T + 3, T, "random")[0] for _ in range(200)}
    assert offs == {0, 1, 2}
    st = np.random.get_state()[2]
    assert do.window_offset(T + 1, T, "random")[0] == 0  # rng==0: numpy draws nothing
    assert np.random.get_state()[2] == st


def test_fixed_right_cutoff_too_large_raises():
    with pytest.raises(ValueError):
        do.window_offset(T + 3, T, "fixed_right_cutoff", 5)


@pytest.mark.reference
def test_live_reference_agrees_on_benchmark_store():
    """Same seeds, same §8(d) synthetic store -> identical batches from the reference module."""
    from oracle import ref_data_shim as shim

    if not shim.available():
        pytest.skip("reference tree not present")
    ref = shim.load_reference_data_module()
    import tempfile
    from microwakeword_amd.ragged import write_ragged_store

    pos, neg = do.synthetic_stores(64, 1234)
    with tempfile.TemporaryDirectory() as tmp:
        write_ragged_store(os.path.join(tmp, "pos", "training", "a_mmap"), pos)
        write_ragged_store(os.path.join(tmp, "neg", "training", "a_mmap"), neg)
        config = {"stride": 1, "window_step_ms": 10, "features": [
            dict(type="mmap", features_dir=os.path.join(tmp, "pos"), truth=True, sampling_weight=2.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
            dict(type="mmap", features_dir=os.path.join(tmp, "neg"), truth=False, sampling_weight=10.0, penalty_weight=1.0, truncation_strategy="random")]}
        random.seed(0); np.random.seed(0)
        fh = ref.FeatureHandler(config)
        rx, ry, rw = fh.get_data("training", 64, T, "default", POLICY)
    random.seed(0); np.random.seed(0)
    provs = [do.index_provider({"training": [pos]}, True, 2.0, 1.0, "truncate_start", 1, 0.01),
             do.index_provider({"training": [neg]}, False, 10.0, 1.0, "random", 1, 0.01)]
    x, y, w, _, _ = do.get_data(provs, "training", 64, T, "default", POLICY)
    np.testing.assert_array_equal(x, rx)
    np.testing.assert_array_equal(y, ry)
    np.testing.assert_array_equal(w, rw)


@pytest.mark.reference
def test_live_reference_fuzz():
    """Random feature sets, policies and truncation strategies (the cases of engine_checks.random_data_case, which the
    HIP path is tested on): the oracle and the reference's own data.py produce identical batches and leave both RNG
    streams in the same place.  Container only (needs /root/reference)."""
    from oracle import ref_data_shim as shim

    if not shim.available():
        pytest.skip("reference tree not present")
    import tempfile

    import engine_checks as ec
    from microwakeword_amd.ragged import write_ragged_store
    ref = shim.load_reference_data_module()
    for case in range(40):
        Tc, provs, policy, B, strategy = ec.random_data_case(case)
        with tempfile.TemporaryDirectory() as tmp:
            feats = []
            for i, p in enumerate(provs):
                write_ragged_store(os.path.join(tmp, "p%d" % i, "training", "a_mmap"), p["store"])
                feats.append(dict(type="mmap", features_dir=os.path.join(tmp, "p%d" % i), truth=p["truth"], sampling_weight=p["sampling_weight"],
                                  penalty_weight=p["penalty_weight"], truncation_strategy=p["truncation_strategy"],
                                  fixed_right_cutoffs=p["fixed_right_cutoffs"]))
            random.seed(case); np.random.seed(case)
            fh = ref.FeatureHandler({"stride": 1, "window_step_ms": 10, "features": feats})
            want = [fh.get_data("training", B, Tc, strategy, policy) for _ in range(2)]
            tail = (random.random(), np.random.random())
        random.seed(case); np.random.seed(case)
        op = [do.index_provider({"training": [p["store"]]}, p["truth"], p["sampling_weight"], p["penalty_weight"],
                                p["truncation_strategy"], 1, 0.01, p["fixed_right_cutoffs"]) for p in provs]
        for rx, ry, rw in want:
            x, y, w, _, _ = do.get_data(op, "training", B, Tc, strategy, policy)
            np.testing.assert_array_equal(x, rx, err_msg="case %d" % case)
            np.testing.assert_array_equal(y, ry)
            np.testing.assert_array_equal(w, rw)
        assert tail == (random.random(), np.random.random()), case
