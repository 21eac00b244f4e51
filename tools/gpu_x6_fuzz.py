"""Random stride-1 flag sets out of the shape table with the x6 options drawn at random (conv1_x6 0 / 1, conv1_x6_fwd 0 / 1,
bwd_first_wide 0 / 1), random (frames, batch, grid): every case on the specialised block kernels against the float64 oracle
(check_train_steps).  usage: python tools/gpu_x6_fuzz.py <first case> <cases>   (MWW_HIP_LIB=tests/hipemu/libmww_emu.so: emulator)"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "tools")
import engine_checks as ec   # noqa: E402
from gpu_table_fuzz import random_table_flags   # noqa: E402
from microwakeword_amd import mixednet, native   # noqa: E402

if __name__ == "__main__":
    lib = native.NativeLib.get()
    first, n = int(sys.argv[1]), int(sys.argv[2])
    bad = ran = 0
    for case in range(first, first + n):
        flags, T, B, grid = random_table_flags(case)
        rng = np.random.default_rng(12000 + case)
        flags = dict(flags, stride=1, conv1_x6=int(rng.integers(0, 2)), conv1_x6_fwd=int(rng.integers(0, 2)), bwd_first_wide=int(rng.integers(0, 2)))
        if mixednet.kernel_family(flags, T, lib=lib)[0] != "block":
            continue
        try:
            ec.check_train_steps(lib, B=B, T=T, steps=1, grid=grid, flags=flags)
            ran += 1
        except ValueError as e:
            if "too short" in str(e):
                continue
            bad += 1
            print("FAIL case", case, flags, T, B, grid, "ValueError", str(e)[:200], flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("FAIL case", case, flags, "T", T, "B", B, "grid", grid, type(e).__name__, str(e)[:300], flush=True)
    print("x6 fuzz: %d cases run, failures: %d" % (ran, bad))
