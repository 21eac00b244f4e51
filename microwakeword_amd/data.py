"""Host-side mirror of the reference loader API with the feature stores resident in HBM.

Same names, arguments and error behaviour as the reference for the mmap-provider path:
  * ``FeatureHandler(config)``                          microwakeword/data.py:405-466
  * ``FeatureHandler.get_data(mode, batch_size, features_length, truncation_strategy,
    augmentation_policy)``                               microwakeword/data.py:497-597
  * ``get_mode_size`` / ``get_mode_duration``            microwakeword/data.py:468-495
  * provider dict keys ``type, features_dir, truth, sampling_weight, penalty_weight,
    truncation_strategy, fixed_right_cutoffs``           microwakeword/data.py:420-433
  * directory convention ``<features_dir>/<mode>/**/*_mmap/``   microwakeword/data.py:171-187

What differs is *where* the work happens: the per-sample Python loop (data.py:555-569) is
replaced by (1) ``mww_sample_training_batch`` — a C++ sampler that continues the very same two
Mersenne-Twister streams (Python ``random`` and ``numpy.random``) in the reference's call order,
so windows and SpecAugment masks are bit-identical under identical seeds — and (2) the HIP
``assemble`` kernel that gathers, pads, scales and masks straight from the HBM-resident store.

``"clips"`` providers (online audio augmentation, data.py:324-402) are out of scope and raise.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import native
from .ragged import FEATURE_BINS, RaggedStoreReader, find_store_dirs

MODES = ("testing", "training", "validation", "testing_ambient", "validation_ambient")  # data.py:171-177
DEFAULT_POLICY = {"freq_mix_prob": 0.0, "time_mask_max_size": 0, "time_mask_count": 0, "freq_mask_max_size": 0,
                  "freq_mask_count": 0}


class MmapFeatureProvider:
    """Index of one provider's ragged stores (the state of the reference's ``MmapFeatureGenerator``,
    data.py:148-211): per mode a shuffled list of (store, sample) plus lengths."""

    def __init__(self, path, label, sampling_weight, penalty_weight, truncation_strategy, stride, step,
                 fixed_right_cutoffs=(0,), stores=None):
        self.label = float(label)
        self.sampling_weight = sampling_weight
        self.penalty_weight = penalty_weight
        self.truncation_strategy = truncation_strategy
        self.fixed_right_cutoffs = list(fixed_right_cutoffs)
        self.stride = stride
        self.step = step
        self.stats: Dict[str, Dict[str, float]] = {}
        self.feature_sets: Dict[str, List[Tuple[int, int]]] = {m: [] for m in MODES}
        self.loaded_features: List[Sequence[np.ndarray]] = []
        for mode in MODES:
            duration, count = 0.0, 0
            if stores is not None:
                mode_stores = list(stores.get(mode, []))
            else:
                mode_stores = [RaggedStoreReader(p) for p in find_store_dirs(path, mode)]
            for st in mode_stores:
                self.loaded_features.append(st)
                fi = len(self.loaded_features) - 1
                for i in range(len(st)):
                    self.feature_sets[mode].append((fi, i))
                    duration += step * st[i].shape[0]
                    count += 1
            random.shuffle(self.feature_sets[mode])  # data.py:206 — consumes the global RNG like the reference
            self.stats[mode] = {"spectrogram_count": count, "total_duration": duration}
        self.store_id: Dict[str, int] = {}
        self.build_flat()

    def build_flat(self, keep=None):
        """The flat HBM image: one store per dtype, every sample's [T_i, 40] frames concatenated.  ``keep`` (a set of
        (store, sample) pairs): only those samples go in - a data-parallel rank keeps its shard of the training samples
        (FeatureHandler.shard_stores, SURVEY 8e "each rank uploads its shard to its own HBM"); the others get no offset."""
        self.flat: Dict[str, np.ndarray] = {}
        self.sample_start: List[List[int]] = []   # [loaded_feature][sub] -> element offset in its flat store (-1: not resident)
        self.sample_len: List[List[int]] = []
        self.feature_dtype: List[str] = []
        chunks: Dict[str, List[np.ndarray]] = {}
        cursor: Dict[str, int] = {}
        for fi, st in enumerate(self.loaded_features):
            starts, lens = [], []
            dt = None
            for i in range(len(st)):
                a = np.asarray(st[i])
                if a.ndim != 2 or a.shape[1] != FEATURE_BINS:
                    raise ValueError("spectrogram %d has shape %s, expected [T,%d]" % (i, a.shape, FEATURE_BINS))
                key = "u16" if a.dtype == np.uint16 else "f32"
                if a.dtype not in (np.uint16, np.float32):
                    a = a.astype(np.float32)
                if dt is None:
                    dt = key
                elif dt != key:
                    raise ValueError("mixed dtypes inside one ragged store")
                lens.append(a.shape[0])
                if keep is not None and (fi, i) not in keep:
                    starts.append(-1)
                    continue
                chunks.setdefault(key, []).append(np.ascontiguousarray(a).reshape(-1))
                starts.append(cursor.get(key, 0))
                cursor[key] = cursor.get(key, 0) + a.size
            self.sample_start.append(starts)
            self.sample_len.append(lens)
            self.feature_dtype.append(dt or "u16")
        for key, parts in chunks.items():
            self.flat[key] = np.concatenate(parts)

    def get_mode_duration(self, mode):
        return self.stats[mode]["total_duration"]

    def get_mode_size(self, mode):
        return self.stats[mode]["spectrogram_count"]

    def strategy(self, requested):
        return self.truncation_strategy if requested == "default" else requested


class FeatureHandler:
    """Drop-in for ``microwakeword.data.FeatureHandler`` (mmap providers)."""

    def __init__(self, config: dict, engine: Optional[native.Engine] = None, shard: Optional[Tuple[int, int]] = None):
        """``shard`` = (rank, world) of a data-parallel job: the rank keeps (and uploads) training samples rank, rank + W, ...
        of every provider (parallel.shard_feature_handler does the same to a handler built without it, at the price of
        one full upload first)."""
        self.feature_providers: List[MmapFeatureProvider] = []
        for feature_set in config["features"]:
            if feature_set["type"] == "mmap":
                self.feature_providers.append(MmapFeatureProvider(
                    feature_set.get("features_dir"), feature_set["truth"], feature_set["sampling_weight"],
                    feature_set["penalty_weight"], feature_set["truncation_strategy"], stride=config["stride"],
                    step=config["window_step_ms"] / 1000.0,
                    fixed_right_cutoffs=feature_set.get("fixed_right_cutoffs", [0]), stores=feature_set.get("stores")))
            elif feature_set["type"] == "clips":
                raise NotImplementedError("'clips' providers (online audio augmentation, reference data.py:324-402) are "
                                          "outside the MI355X hot path; generate RaggedMmap features offline")
            else:
                continue  # the reference silently ignores unknown provider types (data.py:419-452)
        self.engine: Optional[native.Engine] = None
        self._sampler = None
        self._private_rng = None
        self._prefetch_depth = 0
        self._pf = None
        self.eval_shard = (0, 1)   # (rank, world): evaluate_on_device scores windows rank, rank + W, ... (parallel.shard_feature_handler)
        self.uploaded_bytes = 0    # bytes of feature stores this handler sent to its engine's HBM (all uploads)
        self.resident_bytes = 0    # ... and what is resident now
        if shard is not None and int(shard[1]) > 1:
            self.shard_training_lists(int(shard[0]), int(shard[1]))
        if engine is not None:
            self.attach(engine)

    # ---- reference API
    def get_mode_duration(self, mode: str):
        return sum(p.get_mode_duration(mode) for p in self.feature_providers)

    def get_mode_size(self, mode: str):
        return sum(p.get_mode_size(mode) for p in self.feature_providers)

    # ---- device residency
    def attach(self, engine: native.Engine):
        """Uploads every provider's flat store into the engine's HBM (once; again after shard_training_lists)."""
        self.engine = engine
        next_id = 0
        self.resident_bytes = 0
        for p in self.feature_providers:
            for key, flat in p.flat.items():
                if next_id >= native.MWW_MAX_STORES:
                    raise ValueError("too many feature stores")
                engine.upload_store(next_id, flat)
                self.uploaded_bytes += flat.nbytes
                self.resident_bytes += flat.nbytes
                p.store_id[key] = next_id
                next_id += 1
        self._sampler = None
        self._eval_cache = {}

    def shard_training_lists(self, rank: int, world: int):
        """per provider, training sample i of the CANONICAL (store, sample) order goes to rank i mod W - a partition
        whatever per-rank shuffle produced the list - and only those samples (plus every validation / testing sample: their
        windows are sharded by index at evaluation time) stay in the provider's flat HBM image, which is uploaded again if
        the handler is attached.  Idempotent for the same (rank, world)."""
        if getattr(self, "_sharded_for", None) == (int(rank), int(world)):
            return
        if getattr(self, "_sharded_for", None) is not None:
            raise ValueError("feature handler already sharded for rank/world %r" % (self._sharded_for,))
        if int(world) == 1:   # a world of one keeps everything (and the image it has): only the canonical order of the lists
            for p in self.feature_providers:
                p.feature_sets["training"] = sorted(p.feature_sets["training"])
            self._sharded_for = (int(rank), 1)
            self._sampler = None
            return
        for p in self.feature_providers:
            p.feature_sets["training"] = sorted(p.feature_sets["training"])[rank::world]
            if p.stats["training"]["spectrogram_count"] and not p.feature_sets["training"]:
                raise ValueError("provider has fewer training samples than ranks")
            keep = set(p.feature_sets["training"])
            for mode in MODES:
                if mode != "training":
                    keep.update(p.feature_sets[mode])
            p.build_flat(keep)
        self._sharded_for = (int(rank), int(world))
        if getattr(self, "_pf", None) is not None:
            self._drop_prefetcher(keep_streams=True)   # its sampler description points into the old image
        if self.engine is not None:
            self.attach(self.engine)

    def _need_engine(self):
        if self.engine is None:
            # the reference's CLI builds ``FeatureHandler(config)`` BEFORE the model (model_train_eval.py:402-407) and hands both to
            # train.train: a handler built without an engine attaches, at its first use, to the engine created last in this process
            ref = native.Engine.latest
            eng = ref() if ref is not None else None
            if eng is not None and getattr(eng, "h", None):
                self.attach(eng)
                return
        if self.engine is None:
            raise RuntimeError("FeatureHandler is not attached to an MI355X engine (FeatureHandler.attach); "
                               "there is no CPU batch-assembly path")

    def _build_sampler(self):
        live = [p for p in self.feature_providers if p.get_mode_size("training")]
        if not live:
            raise ValueError("no provider has training spectrograms")
        sw = np.array([p.sampling_weight for p in live], np.float64)
        strat = np.array([native.STRATEGIES.get(p.truncation_strategy, -1) for p in live], np.int32)
        offs, st, src, ln, coffs, cuts = [0], [], [], [], [0], []
        for p in live:
            for fi, sub in p.feature_sets["training"]:
                st.append(p.store_id[p.feature_dtype[fi]])
                src.append(p.sample_start[fi][sub])
                ln.append(p.sample_len[fi][sub])
            offs.append(len(st))
            cuts += [int(c) for c in p.fixed_right_cutoffs]
            coffs.append(len(cuts))
        arrs = dict(labels=np.array([p.label for p in live], np.float64),
                    penalty=np.array([float(p.penalty_weight) for p in live], np.float64), sw=sw, strat=strat, offs=np.array(offs, np.int64), st=np.array(st, np.int32),
                    src=np.array(src, np.int64), ln=np.array(ln, np.int32), coffs=np.array(coffs, np.int32),
                    cuts=np.array(cuts if cuts else [0], np.int32))
        d = native.SamplerDesc()
        d.n_providers = len(live)
        d.sampling_weight = arrs["sw"].ctypes.data_as(C.POINTER(C.c_double))
        d.strategy = arrs["strat"].ctypes.data_as(C.POINTER(C.c_int32))
        d.set_offsets = arrs["offs"].ctypes.data_as(C.POINTER(C.c_int64))
        d.set_store = arrs["st"].ctypes.data_as(C.POINTER(C.c_int32))
        d.set_src_elem = arrs["src"].ctypes.data_as(C.POINTER(C.c_int64))
        d.set_len = arrs["ln"].ctypes.data_as(C.POINTER(C.c_int32))
        d.cutoff_offsets = arrs["coffs"].ctypes.data_as(C.POINTER(C.c_int32))
        d.cutoffs = arrs["cuts"].ctypes.data_as(C.POINTER(C.c_int32))
        self._sampler = (d, arrs, live)

    # ---- RNG plumbing: the sampler continues Python's and numpy's global MT19937 streams
    @staticmethod
    def _export_global_rng():
        ps = random.getstate()
        ns = np.random.get_state()
        py = np.array(ps[1], dtype=np.uint32)
        npst = np.empty(625, np.uint32)
        npst[:624] = ns[1]
        npst[624] = ns[2]
        return py, npst, ps, ns

    @staticmethod
    def _import_global_rng(py, npst, ps, ns):
        random.setstate((ps[0], tuple(int(v) for v in py), ps[2]))
        np.random.set_state((ns[0], npst[:624].copy(), int(npst[624]), ns[3], ns[4]))

    def use_private_rng(self, prefetch: int = 4):
        """Snapshot the global RNG states now and keep advancing private copies from here on
        (same streams, no per-batch get/setstate cost).  The global generators are left untouched.
        ``prefetch`` > 0 additionally lets ``next_training_batch_on_device`` take its batches from a worker thread that
        draws that many batches ahead from those private streams (``native.Prefetcher``): same batches in the same order,
        off the launching thread."""
        self._drop_prefetcher(keep_streams=False)
        py, npst, _, _ = self._export_global_rng()
        self._private_rng = (py, npst)
        self._prefetch_depth = int(prefetch)

    def release_private_rng(self):
        """Hand the private streams back: the global ``random`` / ``numpy.random`` generators continue from where the last
        batch HANDED OUT left the private copies, and later draws use the global generators again.  The train loop
        brackets every validation pass with this and ``use_private_rng``: the reference's validation shuffles
        (data.py:593-595, on the global numpy stream) then advance the same stream the next training draws continue
        from, exactly as they do without private streams."""
        self._drop_prefetcher()
        if self._private_rng is not None:
            py, npst = self._private_rng
            _, _, ps, ns = self._export_global_rng()
            self._import_global_rng(py, npst, ps, ns)
            self._private_rng = None

    def _drop_prefetcher(self, keep_streams=True):
        """Stop the worker; the private streams continue from the last batch it HANDED OUT (what it drew ahead is discarded)."""
        pf = self.__dict__.get("_pf")
        if pf is not None:
            if keep_streams and self._private_rng is not None:
                py, npst, _ = pf[1].rng_state()
                self._private_rng = (py, npst)
            pf[1].close()
            self._pf = None

    def _policy(self, augmentation_policy, truncation_strategy):
        pol = dict(DEFAULT_POLICY)
        pol.update(augmentation_policy or {})
        tmax, tc = int(pol["time_mask_max_size"]), int(pol["time_mask_count"])
        fmax, fc = int(pol["freq_mask_max_size"]), int(pol["freq_mask_count"])
        if tc + fc > native.MAX_MASKS:
            raise ValueError("at most %d SpecAugment masks per window" % native.MAX_MASKS)
        if truncation_strategy == "default":
            dstrat = -1
            if any(s < 0 or s == native.STRATEGIES["none"] for s in self._sampler[1]["strat"]):
                raise ValueError("a provider's truncation strategy cannot form a fixed-length training batch")
        else:
            if truncation_strategy not in native.STRATEGIES or truncation_strategy == "none":
                raise ValueError("truncation strategy %r cannot form a fixed-length training batch" % truncation_strategy)
            dstrat = native.STRATEGIES[truncation_strategy]
        return tmax, tc, fmax, fc, dstrat

    def _sample(self, B, features_length, truncation_strategy, augmentation_policy, apply_order):
        self._need_engine()
        self._drop_prefetcher()   # a synchronous draw continues the streams where the last handed-out batch left them
        if self._sampler is None:
            self._build_sampler()
        d, arrs, live = self._sampler
        tmax, tc, fmax, fc, dstrat = self._policy(augmentation_policy, truncation_strategy)
        nm = tc + fc
        buf = self._bufs.get((B, nm)) if hasattr(self, "_bufs") else None
        if buf is None:
            if not hasattr(self, "_bufs"):
                self._bufs = {}
            buf = dict(win=np.zeros(B, native.WINDOW_DTYPE), masks=np.zeros((B, max(nm, 1), 2), np.int32),
                       prov=np.zeros(B, np.int32), samp=np.zeros(B, np.int32), order=np.zeros(B, np.int32))
            buf["ptrs"] = [buf[k].ctypes.data_as(C.c_void_p) for k in ("win", "masks", "prov", "samp", "order")]
            self._bufs[(B, nm)] = buf
        if self._private_rng is not None:
            py, npst = self._private_rng
            saved = None
        else:
            py, npst, ps, ns = self._export_global_rng()
            saved = (ps, ns)
        lib = self.engine.nl
        lib.check(lib.lib.mww_sample_training_batch(
            C.byref(d), py.ctypes.data_as(C.c_void_p), npst.ctypes.data_as(C.c_void_p), B, int(features_length), tmax, tc,
            fmax, fc, dstrat, int(apply_order), *buf["ptrs"]))
        if saved is not None:
            self._import_global_rng(py, npst, *saved)
        return buf, tc, fc, arrs

    def draw_training_batch(self, batch_size, features_length, truncation_strategy="default", augmentation_policy=None):
        """RNG half of ``get_data("training")``: returns (windows, masks, labels, weights) in the
        final (shuffled) order plus the draw-order details for tests."""
        buf, tc, fc, arrs = self._sample(int(batch_size), features_length, truncation_strategy, augmentation_policy, 0)
        nm = tc + fc
        order = buf["order"].copy()
        prov = buf["prov"].copy()
        labels = arrs["labels"][prov]
        weights = arrs["penalty"][prov]
        return dict(windows=buf["win"][order], masks=buf["masks"][order][:, :nm], labels=labels[order], weights=weights[order],
                    n_time=tc, n_freq=fc, order=order, provider=prov, sample=buf["samp"].copy(),
                    draw_windows=buf["win"].copy(), draw_masks=buf["masks"].copy())

    def next_training_batch_on_device(self, batch_size, features_length, truncation_strategy="default",
                                      augmentation_policy=None, class_weights=(1.0, 1.0), weight_broadcast=None,
                                      want_targets=False):
        """Fast path of the train loop: leaves x in the engine's batch buffer (no host copy of the
        spectrograms) and the labels / per-sample weights (penalty x class weight, train.py:288-293) next
        to it — they travel in the same mailbox as the window descriptors, so no separate copy is
        enqueued.  ``class_weights`` = (negative, positive); ``weight_broadcast``: model.combine_weights.  Returns
        ``(labels, penalty_weights)``."""
        neg, pos = class_weights
        if weight_broadcast is None:
            from .model import DEFAULT_WEIGHT_BROADCAST
            weight_broadcast = DEFAULT_WEIGHT_BROADCAST
        if self._prefetch_depth > 0 and self._private_rng is not None:
            # batches drawn ahead by the worker thread: one native call per step on this thread
            self._need_engine()
            pol = dict(DEFAULT_POLICY)
            pol.update(augmentation_policy or {})
            key = (int(batch_size), int(features_length), truncation_strategy, float(neg), float(pos), weight_broadcast,
                   tuple(int(pol[k]) for k in ("time_mask_max_size", "time_mask_count", "freq_mask_max_size", "freq_mask_count")))
            if self._pf is None or self._pf[0] != key or self._sampler is None:
                self._drop_prefetcher()
                if self._sampler is None:
                    self._build_sampler()
                d, arrs, live = self._sampler
                tmax, tc, fmax, fc, dstrat = self._policy(augmentation_policy, truncation_strategy)
                cw = np.where(arrs["labels"] != 0, float(pos), float(neg))   # train.py:288-293: class weight of each provider's label
                py, npst = self._private_rng
                # penalty and class weights travel apart: the worker combines them per batch as `weight_broadcast` says
                # (model.combine_weights; every reading is the per-sample product while the class weights are uniform)
                self._pf = (key, native.Prefetcher(self.engine.nl, d, arrs["labels"], arrs["penalty"], py, npst, int(batch_size),
                                                   int(features_length), tmax, tc, fmax, fc, dstrat, self._prefetch_depth,
                                                   class_weights=cw, broadcast=weight_broadcast))
            return self.engine.assemble_prefetched(self._pf[1], want_targets)
        buf, tc, fc, arrs = self._sample(int(batch_size), features_length, truncation_strategy, augmentation_policy, 1)
        y = arrs["labels"][buf["prov"]]
        w = arrs["penalty"][buf["prov"]]
        if neg == 1.0 and pos == 1.0:
            self.engine.set_targets(y, w)
        else:
            from .model import combine_weights
            self.engine.set_targets(y, combine_weights(w, y, neg, pos, weight_broadcast))
        self.engine.assemble(buf["win"], buf["masks"], tc, fc)
        return y, w

    def _eval_windows(self, mode, features_length, truncation_strategy):
        """Window descriptors, labels and weights of a whole evaluation mode.  Deterministic strategies are
        indexed once and cached (validation runs every eval_step_interval steps on the same windows)."""
        key = (mode, int(features_length), truncation_strategy, self.eval_shard[1] > 1)
        cache = self.__dict__.setdefault("_eval_cache", {})
        if key in cache:
            return cache[key]
        out = self._index_eval_windows(mode, features_length, truncation_strategy)
        if all(p.strategy(truncation_strategy) != "random" for p in self.feature_providers):
            cache[key] = out
        return out

    def _index_eval_windows(self, mode, features_length, truncation_strategy):
        win, labels, weights, split_blocks = [], [], [], []
        for p in self.feature_providers:
            strat = p.strategy(truncation_strategy)
            # data-parallel ranks index the samples in canonical (store, sample) order: the per-mode shuffle of
            # MmapFeatureProvider (global ``random`` stream) need not agree between processes, the shards must
            samples = sorted(p.feature_sets[mode]) if self.eval_shard[1] > 1 else p.feature_sets[mode]
            for fi, sub in samples:
                length = p.sample_len[fi][sub]
                base = p.sample_start[fi][sub]
                sid = p.store_id[p.feature_dtype[fi]]
                if strat == "split":  # data.py:301-311
                    hop = int(1000 * p.step * p.stride)
                    starts = np.arange(0, length - features_length, hop, dtype=np.int64)
                    if starts.size:
                        blk = np.zeros(starts.size, native.WINDOW_DTYPE)
                        blk["store"], blk["copy_rows"], blk["src_elem"] = sid, features_length, base + starts * FEATURE_BINS
                        split_blocks.append((len(win), blk))
                        win.extend([None] * starts.size)
                        labels.extend([p.label] * starts.size)
                        weights.extend([p.penalty_weight] * starts.size)
                    continue
                for cutoff in p.fixed_right_cutoffs:  # data.py:312-321
                    if length > features_length:
                        if strat == "truncate_start":
                            off = length - features_length
                        elif strat == "truncate_end":
                            off = 0
                        elif strat == "fixed_right_cutoff":
                            off = length - features_length - cutoff
                            if off < 0:
                                raise ValueError("fixed_right_cutoff larger than the spare frames")
                        elif strat == "random":
                            off = int(np.random.randint(0, length - features_length))
                        else:
                            raise ValueError("truncation strategy %r does not give fixed-length windows" % strat)
                        win.append((sid, 0, features_length, 0, base + off * FEATURE_BINS))
                    else:
                        win.append((sid, features_length - length, length, 0, base))
                    labels.append(p.label)
                    weights.append(p.penalty_weight)
        if not win:
            return np.zeros(0, native.WINDOW_DTYPE), np.array(labels), np.array(weights)
        out = np.zeros(len(win), native.WINDOW_DTYPE)
        covered = np.zeros(len(win), bool)
        for at, blk in split_blocks:   # the ambient sets' 100 ms-stride windows, built as arrays (tens of thousands of them)
            out[at:at + blk.size] = blk
            covered[at:at + blk.size] = True
        rest = [w for w in win if w is not None]
        if rest:
            out[~covered] = np.array(rest, native.WINDOW_DTYPE)
        return out, np.array(labels), np.array(weights)

    def evaluate_on_device(self, model, mode: str, features_length: int, truncation_strategy: str = "default",
                           batch_size: int = 1024):
        """``get_data(mode, ...)`` + ``model.evaluate(x, y, batch_size)`` (train.py:42-58,75-96) without the
        host round trip of the spectrograms: the windows are gathered from the HBM-resident stores
        straight into the engine's batch buffer, the inference-mode forward runs on them and the
        threshold counters accumulate on the device.  Consumes the same ``np.random.shuffle`` draw as
        ``get_data`` so the global RNG stream stays where the reference would leave it (single process; a data-parallel
        rank shuffles its own shard of the windows, ``eval_shard``, and ``labels`` are that shard's).  Like Keras'
        ``evaluate`` it starts by calling ``model.reset_metrics()`` (looked up at call time, so the
        reference's no-op swap still works).  Returns ``(n_windows, labels, model metric results)``."""
        self._need_engine()
        if model.engine is not self.engine:
            raise ValueError("model and FeatureHandler must share one engine")
        if truncation_strategy == "none":
            raise NotImplementedError("variable-length ('none') evaluation is outside the MI355X path")
        win, labels, _ = self._eval_windows(mode, features_length, truncation_strategy)
        n = win.shape[0]
        rank, world = self.eval_shard
        if world > 1:
            # data-parallel validation: the window list is the same on every rank, rank r scores windows
            # r, r + W, ...; the counters are summed over the ranks when the results are read (Model.evaluation_results)
            mine = np.arange(rank, n, world)
            win, labels = win[mine], labels[mine]
        indices = np.arange(win.shape[0])
        np.random.shuffle(indices)
        win, labels = win[indices], labels[indices].astype(np.float32)
        model.reset_metrics()
        bs = min(int(batch_size), self.engine.max_batch)
        if win.shape[0]:
            self.engine.evaluate_windows(win, labels, bs)   # one native call: the batches are walked inside the library
        return n, labels, model.evaluation_results()

    def get_data(self, mode: str, batch_size: int, features_length: int, truncation_strategy: str = "default",
                 augmentation_policy: dict = DEFAULT_POLICY):
        """Same contract as the reference (data.py:497-597): returns host arrays
        ``(x float32 [N,T,40], labels float64 [N], weights float64 [N])``."""
        self._need_engine()
        eng = self.engine
        if mode == "training":
            b = self.draw_training_batch(batch_size, features_length, truncation_strategy, augmentation_policy)
            chunks = []
            for s in range(0, batch_size, eng.max_batch):
                e = min(batch_size, s + eng.max_batch)
                eng.assemble(b["windows"][s:e], b["masks"][s:e], b["n_time"], b["n_freq"])
                chunks.append(eng.get_batch(e - s))
            x = np.concatenate(chunks) if len(chunks) > 1 else chunks[0]
            return x, b["labels"], b["weights"]
        if truncation_strategy == "none":
            raise NotImplementedError("variable-length ('none') evaluation batches are produced by the reference "
                                      "loader only for streaming TFLite tests, outside the MI355X path")
        win, labels, weights = self._eval_windows(mode, features_length, truncation_strategy)
        n = win.shape[0]
        indices = np.arange(n)
        np.random.shuffle(indices)  # data.py:593-595 (the guard there is always true)
        win = win[indices]
        if n == 0:
            return np.zeros((0,), np.float64), labels, weights  # np.array([]) in the reference
        chunks = []
        for s in range(0, n, eng.max_batch):
            e = min(n, s + eng.max_batch)
            eng.assemble(win[s:e], None, 0, 0)
            chunks.append(eng.get_batch(e - s))
        x = np.concatenate(chunks) if len(chunks) > 1 else chunks[0]
        return x, labels[indices], weights[indices]
