"""TEST INFRASTRUCTURE ONLY — imports the *reference's own* ``microwakeword/data.py`` unchanged.

The reference data path is pure numpy (reference ``microwakeword/data.py:32-597``); it imports
third-party modules that are absent here (absl, mmap_ninja, microwakeword.audio.*).  This shim
registers stand-ins for those modules in ``sys.modules`` and then imports the reference file from
``/root/reference`` so it can be used to (a) generate the committed golden fixtures under
``tests/golden/`` and (b) validate ``oracle/data_oracle.py`` in this container.

It cannot travel: ``/root/reference`` does not exist on the GPU box, so nothing in ``-m gpu``
tests, ``smoke()`` or ``bench.py`` may import this module.  Nothing is ever written under
``/root/reference`` (bytecode writing is disabled before the import).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MWW_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "microwakeword", "data.py"))


class ListRaggedMmap:
    """Stand-in for ``mmap_ninja.ragged.RaggedMmap`` (reference ``data.py:25,190``): the
    reference only uses ``RaggedMmap(path)``, ``len()``, ``[i]`` -> ndarray.  ``path`` is looked
    up in a process-local registry of in-RAM ragged lists, or read from a ``.npz`` written by
    ``microwakeword_amd.ragged`` (keys ``data``, ``starts``, ``lens``)."""

    registry = {}

    def __init__(self, path):
        path = str(path).rstrip("/")
        if path in self.registry:
            self.items = self.registry[path]
        else:
            from microwakeword_amd.ragged import RaggedStoreReader  # format reader, numpy only

            self.items = RaggedStoreReader(path)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def load_reference_data_module():
    """Returns the reference ``microwakeword.data`` module object (imported from source)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if "mww_reference_data" in sys.modules:
        return sys.modules["mww_reference_data"]

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    logging = stub("absl.logging", info=lambda *a, **k: None, warning=lambda *a, **k: None)
    stub("absl", logging=logging)
    ragged = stub("mmap_ninja.ragged", RaggedMmap=ListRaggedMmap)
    stub("mmap_ninja", ragged=ragged)
    # the reference package itself is NOT imported (its __init__ is empty, but the audio
    # modules pull in TF); only the three names data.py needs are provided.
    pkg = stub("microwakeword")
    pkg.__path__ = []  # mark as package
    audio = stub("microwakeword.audio")
    audio.__path__ = []
    stub("microwakeword.audio.clips", Clips=type("Clips", (), {}))
    stub("microwakeword.audio.augmentation", Augmentation=type("Augmentation", (), {}))
    stub("microwakeword.audio.spectrograms", SpectrogramGeneration=type("SpectrogramGeneration", (), {}))

    spec = importlib.util.spec_from_file_location(
        "mww_reference_data", os.path.join(REFERENCE_ROOT, "microwakeword", "data.py")
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mww_reference_data"] = mod
    spec.loader.exec_module(mod)
    return mod
