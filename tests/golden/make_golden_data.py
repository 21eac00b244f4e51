"""Generates tests/golden/data_golden.npz by running the REFERENCE's own microwakeword/data.py
(imported via oracle/ref_data_shim.py) on small seeded ragged stores.  Run in the build container
(where /root/reference exists):   python tests/golden/make_golden_data.py
The .npz holds the inputs (store samples) and the reference outputs, so the tests need neither
the reference tree nor this script at run time."""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from microwakeword_amd.ragged import write_ragged_store  # noqa: E402
from oracle.ref_data_shim import load_reference_data_module  # noqa: E402

T = 194
POLICY = dict(freq_mix_prob=0.0, time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)


def make_samples(rng, n, lo, hi, dtype):
    lens = rng.integers(lo, hi + 1, size=n)
    if dtype == np.uint16:
        return [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16) for l in lens]
    # float32 stores hold already-scaled features (multiples of 1/25.6), which also keeps the .npz small
    return [rng.integers(0, 667, size=(int(l), 40)).astype(np.float32) * np.float32(0.0390625) for l in lens]


def codes(x):
    """Reference outputs are exact multiples of 1/25.6 = 0.0390625; store them as uint16 codes
    (checked to round-trip bit-exactly) so the fixture stays small."""
    c = np.rint(np.asarray(x, np.float64) / 0.0390625).astype(np.uint16)
    assert np.array_equal(c.astype(np.float32) * np.float32(0.0390625), np.asarray(x, np.float32))
    return c


def main():
    ref = load_reference_data_module()
    out = {}
    for tag, dtype in (("u16", np.uint16), ("f32", np.float32)):
        rng = np.random.default_rng(7 if tag == "u16" else 8)
        with tempfile.TemporaryDirectory() as tmp:
            spec = {
                "pos": dict(training=make_samples(rng, 12, 150, 260, dtype), validation=make_samples(rng, 5, 150, 260, dtype)),
                "neg": dict(training=make_samples(rng, 12, 150, 260, dtype), validation=make_samples(rng, 5, 150, 260, dtype),
                            validation_ambient=make_samples(rng, 2, 300, 380, dtype)),
                "cut": dict(training=make_samples(rng, 6, 230, 300, dtype)),
            }
            for prov, modes in spec.items():
                for mode, samples in modes.items():
                    write_ragged_store(os.path.join(tmp, prov, mode, "s_mmap"), samples)
                    for i, s in enumerate(samples):
                        out["%s/in/%s/%s/%d" % (tag, prov, mode, i)] = s
            config = {
                "stride": 1, "window_step_ms": 10,
                "features": [
                    dict(type="mmap", features_dir=os.path.join(tmp, "pos"), truth=True, sampling_weight=2.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
                    dict(type="mmap", features_dir=os.path.join(tmp, "neg"), truth=False, sampling_weight=10.0, penalty_weight=1.5, truncation_strategy="random"),
                    dict(type="mmap", features_dir=os.path.join(tmp, "cut"), truth=False, sampling_weight=3.0, penalty_weight=0.5, truncation_strategy="fixed_right_cutoff", fixed_right_cutoffs=[0, 5, 11]),
                ],
            }
            random.seed(3)
            np.random.seed(3)
            fh = ref.FeatureHandler(config)
            for call in range(2):
                x, y, w = fh.get_data("training", 16, T, "default", POLICY)
                out["%s/train%d/xc" % (tag, call)] = codes(x)
                out["%s/train%d/y" % (tag, call)] = y
                out["%s/train%d/w" % (tag, call)] = w
            x, y, w = fh.get_data("validation", 16, T, "truncate_start")
            out[tag + "/val/xc"], out[tag + "/val/y"], out[tag + "/val/w"] = codes(x), y, w
            x, y, w = fh.get_data("validation_ambient", 16, T, "split")
            out[tag + "/amb/xc"], out[tag + "/amb/y"], out[tag + "/amb/w"] = codes(x), y, w
            out[tag + "/sizes"] = np.array([fh.get_mode_size(m) for m in ("training", "validation", "validation_ambient")])
            out[tag + "/durations"] = np.array([fh.get_mode_duration(m) for m in ("training", "validation", "validation_ambient")])
    # the survey's first known-answer vector
    random.seed(0)
    np.random.seed(0)
    out["ka/spec_augment_ones"] = ref.spec_augment(np.ones((T, 40), np.float32), 5, 2, 5, 2)
    # fixed_length_spectrogram on every strategy
    base = np.arange(300 * 40, dtype=np.float32).reshape(300, 40)
    np.random.seed(11)
    for strat in ("random", "truncate_start", "truncate_end", "fixed_right_cutoff", "none"):
        out["ka/fls_long_" + strat] = ref.fixed_length_spectrogram(base, T, strat, 7)
    out["ka/fls_short"] = ref.fixed_length_spectrogram(base[:100], T, "random", 0)
    out["ka/fls_equal"] = ref.fixed_length_spectrogram(base[:T], T, "random", 0)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "data_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
