"""Random (frames, batch, grid) sweep of the specialised MixedNet kernels (default and notebook topologies) against the oracle."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import engine_checks as ec   # noqa: E402
from microwakeword_amd import native   # noqa: E402

lib = native.NativeLib.get()
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for case in range(first, first + n):
    rng = np.random.default_rng(7000 + case)
    nb = rng.random() < 0.35
    flags = ec.NOTEBOOK if nb else ec.DEF
    T = int(rng.integers(110, 300)) if nb else int(rng.integers(52, 300))
    B = int(rng.integers(1, 48))
    grid = int(rng.choice([0, 1, 2, 3, 5, 8]))
    try:
        ec.check_train_steps(lib, B=B, T=T, steps=1, grid=grid, flags=flags)
    except ValueError as e:
        if "too short" in str(e):
            continue
        bad += 1
        print("FAIL case", case, nb, T, B, grid, "ValueError", str(e)[:200], flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL case", case, "notebook" if nb else "default", "T", T, "B", B, "grid", grid, type(e).__name__, str(e)[:300], flush=True)
print("done, failures:", bad)
