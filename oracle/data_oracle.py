"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference batch-assembly path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``microwakeword_amd``) never does and fails loudly when the HIP
library is missing.

What is restated (reference file:line):
  * ``spec_augment``                     microwakeword/data.py:32-71
  * ``fixed_length_spectrogram``         microwakeword/data.py:74-118
  * ``MmapFeatureGenerator.__init__``    microwakeword/data.py:148-211  (index + per-mode shuffle)
  * ``get_random_spectrogram``           microwakeword/data.py:235-271
  * ``get_feature_generator``            microwakeword/data.py:273-321
  * ``FeatureHandler.get_data``          microwakeword/data.py:497-597

It is written in *descriptor* form: every output window is first reduced to integers
(``store, sample, src_row, copy_rows, pad_rows`` + mask rectangles) drawn from the two host RNGs
(Python ``random`` and ``numpy.random`` legacy global state) in exactly the reference's call
order, and only then materialised.  The descriptor is the very thing the HIP
``assemble`` kernel consumes, so descriptor equality == "SpecAugment mask indices bit-exact".

Pinning: ``tests/test_data_oracle.py`` checks this file against golden vectors produced by the
reference's own ``data.py`` (``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` exists) and, when the reference tree is present, against the reference
module directly under identical seeds.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

U16_SCALE = np.float32(0.0390625)  # data.py:268-269  (== 1/25.6)
MODES = ("testing", "training", "validation", "testing_ambient", "validation_ambient")  # data.py:171-177


@dataclass
class WindowDesc:
    """One output window, as integers."""

    provider: int
    store: int
    sample: int
    src_row: int  # first source frame copied
    copy_rows: int  # frames copied
    pad_rows: int  # zero frames in front (left pad, data.py:107-113)
    time_masks: List[Tuple[int, int]] = field(default_factory=list)  # (t0, t)
    freq_masks: List[Tuple[int, int]] = field(default_factory=list)  # (f0, f)


@dataclass
class OracleProvider:
    """State of one ``MmapFeatureGenerator`` (data.py:121-211)."""

    stores: List[Sequence[np.ndarray]]
    feature_sets: Dict[str, List[Tuple[int, int]]]
    stats: Dict[str, Dict[str, float]]
    label: float
    sampling_weight: float
    penalty_weight: float
    truncation_strategy: str
    fixed_right_cutoffs: List[int]
    stride: int
    step: float

    def mode_size(self, mode):
        return self.stats[mode]["spectrogram_count"]


def index_provider(stores_by_mode, label, sampling_weight, penalty_weight, truncation_strategy,
                   stride, step, fixed_right_cutoffs=(0,)) -> OracleProvider:
    """``stores_by_mode``: {mode: [store, ...]} in the order the reference's glob would visit
    them.  Consumes ``random.shuffle`` once per mode in the reference's mode order, exactly as
    data.py:179-211 does."""
    stores: List[Sequence[np.ndarray]] = []
    sets: Dict[str, List[Tuple[int, int]]] = {m: [] for m in MODES}
    stats = {}
    for mode in MODES:
        duration = 0.0
        count = 0
        for st in stores_by_mode.get(mode, []):
            stores.append(st)
            si = len(stores) - 1
            for i in range(len(st)):
                sets[mode].append((si, i))
                duration += step * st[i].shape[0]
                count += 1
        random.shuffle(sets[mode])
        stats[mode] = {"spectrogram_count": count, "total_duration": duration}
    return OracleProvider(stores, sets, stats, float(label), sampling_weight, penalty_weight,
                          truncation_strategy, list(fixed_right_cutoffs), stride, step)


def window_offset(length: int, features_length: int, strategy: str, right_cutoff: int = 0):
    """data.py:93-118 reduced to integers -> (src_row, copy_rows, pad_rows).  Draws
    ``np.random.randint`` only for strategy "random" with length > features_length."""
    if length > features_length:
        if strategy == "random":
            off = int(np.random.randint(0, length - features_length))
        elif strategy == "truncate_start":
            off = length - features_length
        elif strategy == "truncate_end":
            off = 0
        elif strategy == "fixed_right_cutoff":
            off = length - features_length - right_cutoff
            if off < 0:
                # the reference would slice with a negative start and yield a short/empty window
                raise ValueError("fixed_right_cutoff %d larger than the %d spare frames" % (right_cutoff, length - features_length))
        elif strategy == "none":
            return 0, length, 0
        else:
            off = 0  # data.py leaves features_offset at 0 for unknown strategies
        return off, features_length, 0
    return 0, length, features_length - length


def draw_masks(time_frames, freq_bins, tmax, tcount, fmax, fcount):
    """data.py:61-69: width from np.random.uniform (truncated), start from random.randint."""
    tm, fm = [], []
    for _ in range(tcount):
        t = int(np.random.uniform(0, tmax))
        t0 = random.randint(0, time_frames - t)
        tm.append((t0, t))
    for _ in range(fcount):
        f = int(np.random.uniform(0, fmax))
        f0 = random.randint(0, freq_bins - f)
        fm.append((f0, f))
    return tm, fm


def materialise(sample: np.ndarray, d: WindowDesc, features_length: int) -> np.ndarray:
    rows = d.pad_rows + d.copy_rows
    out_dtype = np.float32 if sample.dtype == np.uint16 else sample.dtype
    out = np.zeros((rows, sample.shape[1]), out_dtype)
    src = sample[d.src_row:d.src_row + d.copy_rows]
    if sample.dtype == np.uint16:
        src = src.astype(np.float32) * U16_SCALE
    out[d.pad_rows:] = src
    for t0, t in d.time_masks:
        out[t0:t0 + t, :] = 0
    for f0, f in d.freq_masks:
        out[:, f0:f0 + f] = 0
    return out


def draw_training_descs(providers: List[OracleProvider], batch_size, features_length,
                        truncation_strategy="default", policy=None) -> List[WindowDesc]:
    """RNG call order of one ``get_data("training")`` up to (not including) the final shuffle."""
    policy = policy or {}
    tmax = policy.get("time_mask_max_size", 0)
    tc = policy.get("time_mask_count", 0)
    fmax = policy.get("freq_mask_max_size", 0)
    fc = policy.get("freq_mask_count", 0)
    live = [i for i, p in enumerate(providers) if p.mode_size("training")]
    chosen = random.choices(live, [providers[i].sampling_weight for i in live], k=batch_size)
    descs = []
    for pi in chosen:
        p = providers[pi]
        strat = p.truncation_strategy if truncation_strategy == "default" else truncation_strategy
        cutoff = random.choice(p.fixed_right_cutoffs) if strat == "fixed_right_cutoff" else 0
        si, sub = random.choice(p.feature_sets["training"])
        length = p.stores[si][sub].shape[0]
        off, cp, pad = window_offset(length, features_length, strat, cutoff)
        tm, fm = draw_masks(pad + cp, 40, tmax, tc, fmax, fc)
        descs.append(WindowDesc(pi, si, sub, off, cp, pad, tm, fm))
    return descs


def eval_descs(providers: List[OracleProvider], mode, features_length, truncation_strategy="default"):
    """Window list of the non-training branch (data.py:571-579 + 273-321), no RNG."""
    descs = []
    for pi, p in enumerate(providers):
        strat = p.truncation_strategy if truncation_strategy == "default" else truncation_strategy
        for si, sub in p.feature_sets[mode]:
            length = p.stores[si][sub].shape[0]
            if strat == "split":
                hop = int(1000 * p.step * p.stride)
                for s0 in range(0, length - features_length, hop):
                    descs.append(WindowDesc(pi, si, sub, s0, features_length, 0))
            else:
                for cutoff in p.fixed_right_cutoffs:
                    if strat == "random":
                        raise ValueError("strategy 'random' is not deterministic in evaluation")
                    off, cp, pad = window_offset(length, features_length, strat, cutoff)
                    descs.append(WindowDesc(pi, si, sub, off, cp, pad))
    return descs


def get_data(providers: List[OracleProvider], mode, batch_size, features_length,
             truncation_strategy="default", augmentation_policy=None):
    """Same contract as ``FeatureHandler.get_data`` (data.py:497-597) for mmap providers.
    Returns ``(x, y, w, descs, order)``: ``descs`` in draw order, ``order`` the final shuffle."""
    if mode == "training":
        descs = draw_training_descs(providers, batch_size, features_length, truncation_strategy,
                                    augmentation_policy)
    else:
        descs = eval_descs(providers, mode, features_length, truncation_strategy)
    data = [materialise(providers[d.provider].stores[d.store][d.sample], d, features_length) for d in descs]
    labels = np.array([float(providers[d.provider].label) for d in descs])
    weights = np.array([float(providers[d.provider].penalty_weight) for d in descs])
    if truncation_strategy == "none":
        return data, labels, weights, descs, np.arange(len(descs))
    x = np.array(data)
    order = np.arange(labels.shape[0])
    np.random.shuffle(order)  # data.py:593-595 — the guard is always true
    return x[order], labels[order], weights[order], descs, order


# ---- synthetic feature stores of SURVEY §8(d) -------------------------------------------------

def synthetic_stores(n_samples=4096, seed=1234, dtype=np.uint16, min_len=150, max_len=400):
    """Two providers' worth of ragged samples: lengths U{min..max}, values U{0..666} (uint16) or
    U[0,26) (float32), ``numpy.random.default_rng(seed)`` — does not touch the global RNGs."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        lens = rng.integers(min_len, max_len + 1, size=n_samples)
        if dtype == np.uint16:
            st = [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16) for l in lens]
        else:
            st = [(rng.random((int(l), 40), dtype=np.float32) * np.float32(26.0)) for l in lens]
        out.append(st)
    return out


def synthetic_providers(n_samples=4096, seed=1234, dtype=np.uint16, stride=1, step=0.01,
                        n_val=0, n_ambient=0, **kw):
    """The §8(d) benchmark feature set: provider 0 = label 1, sampling_weight 2, truncate_start;
    provider 1 = label 0, sampling_weight 10, random; penalty weights 1/1."""
    pos, neg = synthetic_stores(n_samples, seed, dtype, **kw)
    rng = np.random.default_rng(seed + 1)

    def extra(n, lo, hi):
        return [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16)
                for l in rng.integers(lo, hi + 1, size=n)] if n else None

    pos_modes = {"training": [pos]}
    neg_modes = {"training": [neg]}
    if n_val:
        pos_modes["validation"] = [extra(n_val, 150, 400)]
        neg_modes["validation"] = [extra(n_val, 150, 400)]
    if n_ambient:
        neg_modes["validation_ambient"] = [extra(n_ambient, 600, 1500)]
    p0 = index_provider(pos_modes, True, 2.0, 1.0, "truncate_start", stride, step)
    p1 = index_provider(neg_modes, False, 10.0, 1.0, "random", stride, step)
    return [p0, p1]
