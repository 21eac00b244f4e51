"""Ragged spectrogram stores: the on-disk format either side of the hot path.

The reference opens every ``**/*_mmap/`` directory with the third-party
``mmap_ninja.ragged.RaggedMmap`` (reference ``microwakeword/data.py:25,171-211``) and only uses
``RaggedMmap(path)``, ``len()``, ``[i] -> ndarray [T_i, 40]`` (dtype uint16 or float32).

``mmap_ninja`` is not installed here and its layout is undocumented in the reference.  The
layout implemented below is the one recalled from mmap_ninja 0.7 (a ``data/`` numpy-memmap dir
holding every sample concatenated flat, plus ``starts/``, ``ends/``, ``shapes/``,
``flattened_shapes/`` numpy-memmap dirs; each numpy-memmap dir = ``data.ninja`` raw bytes +
``dtype.ninja`` + ``shape.ninja`` + ``order.ninja`` text files).  **It has not been verified
against a folder written by the real library** (no wheel, no network here).  Therefore, in this order:
if ``mmap_ninja`` is importable it is used; a ``flat_export.npz`` next to the store (written on a machine
that has the library by ``tools/export_ragged_to_flat.py``) is the verified interchange; only then the
recalled layout is tried, and it is accepted only if every file, dtype, order and offset invariant of that
layout holds — anything else raises ``ValueError`` instead of guessing.

The GPU path never touches these files per step: :func:`flatten_store` turns a store into the
three flat arrays (``data``, ``starts``, ``lens``) that are uploaded once into HBM.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Iterable, List, Sequence, Tuple

import numpy as np

FEATURE_BINS = 40


def _write_np_dir(d: Path, arr: np.ndarray) -> None:
    d.mkdir(parents=True, exist_ok=True)
    arr = np.ascontiguousarray(arr)
    arr.tofile(d / "data.ninja")
    (d / "dtype.ninja").write_text(str(arr.dtype))
    (d / "shape.ninja").write_text(",".join(str(s) for s in arr.shape))
    (d / "order.ninja").write_text("C")


def _open_np_dir(d: Path) -> np.ndarray:
    try:
        dtype = np.dtype((d / "dtype.ninja").read_text().strip())
        shape_txt = (d / "shape.ninja").read_text().strip().strip("()")
        shape = tuple(int(s) for s in shape_txt.replace(" ", "").split(",") if s)
    except (OSError, ValueError, TypeError) as e:
        raise ValueError("not a numpy-memmap directory: %s (%s)" % (d, e)) from None
    if int(np.prod(shape)) == 0:
        return np.zeros(shape, dtype)
    return np.memmap(d / "data.ninja", dtype=dtype, mode="r", shape=shape)


FLAT_EXPORT = "flat_export.npz"   # {data, starts, lens}: see tools/export_ragged_to_flat.py


def _check_invariants(path, data, starts, ends, shapes, flat_shapes):
    """Everything the layout implies; any violation means "this is not the format I think it is"."""
    def bad(msg):
        raise ValueError("unrecognised ragged store layout: %s: %s (install mmap_ninja, or export the store with "
                         "tools/export_ragged_to_flat.py where it is installed)" % (path, msg))
    if data.ndim != 1 or data.dtype not in (np.dtype(np.uint16), np.dtype(np.float32)):
        bad("data must be a flat uint16 / float32 array, got %s %s" % (data.dtype, data.shape))
    if starts.ndim != 1 or starts.shape != ends.shape or starts.dtype.kind not in "iu" or ends.dtype.kind not in "iu":
        bad("starts / ends must be integer vectors of one length")
    n = starts.shape[0]
    if n == 0:
        bad("empty store")
    st, en = starts.astype(np.int64), ends.astype(np.int64)
    if st[0] != 0 or np.any(en <= st) or np.any(st[1:] != en[:-1]) or en[-1] != data.shape[0]:
        bad("samples must tile the data array without gaps (starts[0] = 0, starts[i+1] = ends[i], ends[-1] = len(data))")
    if np.any((en - st) % FEATURE_BINS):
        bad("every sample must hold whole [T, %d] frames" % FEATURE_BINS)
    if shapes is not None:
        if shapes.dtype.kind not in "iu" or flat_shapes.dtype.kind not in "iu" or shapes.shape != (2 * n,) or flat_shapes.shape != (n,):
            bad("shapes / flattened_shapes must list one [T, %d] shape per sample" % FEATURE_BINS)
        sh = shapes.astype(np.int64).reshape(n, 2)
        if np.any(flat_shapes.astype(np.int64) != 2 * np.arange(n)) or np.any(sh[:, 1] != FEATURE_BINS) or np.any(sh[:, 0] * FEATURE_BINS != en - st):
            bad("shapes do not match the start / end offsets")


def write_ragged_store(path: str, samples: Iterable[np.ndarray]) -> None:
    """Writes ``samples`` (each ``[T_i, 40]``, one common dtype) as a ``*_mmap`` directory."""
    samples = [np.ascontiguousarray(s) for s in samples]
    if not samples:
        raise ValueError("empty store")
    dtype = samples[0].dtype
    if any(s.dtype != dtype for s in samples):
        raise ValueError("mixed dtypes in one store")
    out = Path(path)
    out.mkdir(parents=True, exist_ok=True)
    flat = np.concatenate([s.reshape(-1) for s in samples])
    sizes = np.array([s.size for s in samples], np.int64)
    ends = np.cumsum(sizes)
    starts = ends - sizes
    shapes = np.concatenate([np.array(s.shape, np.int64) for s in samples])
    nd = np.array([s.ndim for s in samples], np.int64)
    _write_np_dir(out / "data", flat)
    _write_np_dir(out / "starts", starts)
    _write_np_dir(out / "ends", ends)
    _write_np_dir(out / "shapes", shapes)
    _write_np_dir(out / "flattened_shapes", np.cumsum(nd) - nd)


class RaggedStoreReader:
    """Duck-type of ``RaggedMmap`` restricted to what the reference uses (``data.py:190-204``)."""

    def __init__(self, path: str):
        p = Path(str(path))
        self.path = str(p)
        self._real = None
        try:  # prefer the real library when it exists (verified layout by construction)
            from mmap_ninja.ragged import RaggedMmap  # type: ignore

            if RaggedMmap.__module__.startswith("mmap_ninja"):
                self._real = RaggedMmap(p)
        except Exception:
            self._real = None
        if self._real is None and (p / FLAT_EXPORT).is_file():
            # the verified interchange: written on a machine that has the real mmap_ninja by tools/export_ragged_to_flat.py
            z = np.load(p / FLAT_EXPORT)
            self.data = z["data"]
            self.starts = np.asarray(z["starts"], np.int64)
            self.ends = self.starts + np.asarray(z["lens"], np.int64) * FEATURE_BINS
            _check_invariants(self.path, self.data, self.starts, self.ends, None, None)
        elif self._real is None:
            # recalled (unverified) mmap_ninja layout: accepted only if EVERY invariant of that layout holds, so a folder
            # written by a different library version fails here, loudly, instead of yielding shifted spectrograms
            for sub in ("data", "starts", "ends", "shapes", "flattened_shapes"):
                for f in ("data.ninja", "dtype.ninja", "shape.ninja", "order.ninja"):
                    if not (p / sub / f).is_file():
                        raise ValueError("unrecognised ragged store layout: %s lacks %s/%s (install mmap_ninja, or export the store "
                                         "with tools/export_ragged_to_flat.py where it is installed)" % (p, sub, f))
                if (p / sub / "order.ninja").read_text().strip() != "C":
                    raise ValueError("unrecognised ragged store layout: %s/%s is not C-ordered" % (p, sub))
            self.data = _open_np_dir(p / "data")
            self.starts = np.asarray(_open_np_dir(p / "starts"))
            self.ends = np.asarray(_open_np_dir(p / "ends"))
            _check_invariants(self.path, self.data, self.starts, self.ends, np.asarray(_open_np_dir(p / "shapes")),
                              np.asarray(_open_np_dir(p / "flattened_shapes")))
            self.starts, self.ends = self.starts.astype(np.int64), self.ends.astype(np.int64)

    def __len__(self) -> int:
        return len(self._real) if self._real is not None else int(self.starts.shape[0])

    def __getitem__(self, i: int) -> np.ndarray:
        if self._real is not None:
            return self._real[i]
        s, e = int(self.starts[i]), int(self.ends[i])
        if (e - s) % FEATURE_BINS:
            raise ValueError("sample %d of %s is not [T,%d]" % (i, self.path, FEATURE_BINS))
        return self.data[s:e].reshape(-1, FEATURE_BINS)

    @property
    def dtype(self):
        return self[0].dtype


def flatten_store(store: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``store`` (anything list-like of ``[T_i,40]`` arrays) -> ``(flat, starts, lens)``.

    ``flat`` is 1-D (uint16 or float32, elements), ``starts[i]`` the element offset of sample i and
    ``lens[i]`` its frame count.  This is the HBM layout of a feature store.
    """
    if isinstance(store, RaggedStoreReader) and store._real is None:
        lens = (store.ends - store.starts) // FEATURE_BINS
        return np.asarray(store.data), store.starts.copy(), lens.astype(np.int32)
    arrs = [np.ascontiguousarray(store[i]) for i in range(len(store))]
    dt = arrs[0].dtype
    if dt not in (np.dtype(np.uint16), np.dtype(np.float32)):
        raise ValueError("feature stores must be uint16 or float32, got %s" % dt)
    lens = np.array([a.shape[0] for a in arrs], np.int32)
    sizes = lens.astype(np.int64) * FEATURE_BINS
    starts = np.cumsum(sizes) - sizes
    flat = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, dt)
    return flat, starts.astype(np.int64), lens


def find_store_dirs(features_dir: str, mode: str) -> List[str]:
    """Directory convention of the reference (``data.py:171-187``): every ``**/*_mmap/`` below
    ``<features_dir>/<mode>/``, in ``Path.glob`` order."""
    base = Path(os.path.abspath(os.path.join(features_dir, mode)))
    return [str(i) for i in base.glob("**/*_mmap/")]
