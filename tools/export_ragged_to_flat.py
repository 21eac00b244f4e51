#!/usr/bin/env python
"""Run where the real ``mmap_ninja`` is installed (the machine that generated the features):

    python tools/export_ragged_to_flat.py <features_dir> [<features_dir> ...]

For every ``**/*_mmap/`` directory below the given roots (the reference's convention, microwakeword/data.py:171-187)
this opens the store with ``mmap_ninja.ragged.RaggedMmap`` — the library itself, so the on-disk layout is whatever that
version writes — and drops a ``flat_export.npz`` into the directory: ``data`` (all samples concatenated flat, uint16 or
float32), ``starts`` (element offset of each sample) and ``lens`` (frames per sample).  ``microwakeword_amd.ragged``
prefers that file over its own (unverified) reading of the mmap_ninja layout, and the GPU loader uploads exactly these
three arrays.  Needs numpy and mmap_ninja only."""
import sys
from pathlib import Path

import numpy as np
from mmap_ninja.ragged import RaggedMmap

FEATURE_BINS = 40


def export(store_dir: Path) -> None:
    store = RaggedMmap(store_dir)
    n = len(store)
    if n == 0:
        raise SystemExit("%s is empty" % store_dir)
    first = np.asarray(store[0])
    if first.ndim != 2 or first.shape[1] != FEATURE_BINS or first.dtype not in (np.dtype(np.uint16), np.dtype(np.float32)):
        raise SystemExit("%s: samples must be [T, %d] uint16 / float32, got %s %s" % (store_dir, FEATURE_BINS, first.dtype, first.shape))
    lens = np.empty(n, np.int32)
    parts = []
    for i in range(n):
        a = np.ascontiguousarray(store[i])
        if a.ndim != 2 or a.shape[1] != FEATURE_BINS or a.dtype != first.dtype:
            raise SystemExit("%s: sample %d is %s %s" % (store_dir, i, a.dtype, a.shape))
        lens[i] = a.shape[0]
        parts.append(a.reshape(-1))
    sizes = lens.astype(np.int64) * FEATURE_BINS
    np.savez(store_dir / "flat_export.npz", data=np.concatenate(parts), starts=np.cumsum(sizes) - sizes, lens=lens)
    print("%s: %d samples, %d frames -> flat_export.npz" % (store_dir, n, int(lens.sum())))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    for root in sys.argv[1:]:
        dirs = sorted(Path(root).glob("**/*_mmap/"))
        if not dirs:
            raise SystemExit("no *_mmap directories below %s" % root)
        for d in dirs:
            export(d)
