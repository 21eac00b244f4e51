"""Inception topology fuzz with per-case reporting."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import engine_checks as ec   # noqa: E402
from microwakeword_amd import native   # noqa: E402

lib = native.NativeLib.get()
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for case in range(first, first + n):
    flags = ec.random_inception_flags(case)
    try:
        ec.check_inception_train_steps(lib, B=3, T=150, steps=1, grid=2, flags=flags)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL case", case, type(e).__name__, str(e)[:300])
        print("   ", flags, flush=True)
print("done, failures:", bad)
