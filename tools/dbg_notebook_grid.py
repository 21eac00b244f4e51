import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import engine_checks as ec
from microwakeword_amd import native
lib = native.NativeLib.get()
for B, grid in ((600, 512), (600, 256), (600, 0), (300, 512), (1024, 512), (600, 384)):
    try:
        w = ec.check_train_steps(lib, B=B, T=204, steps=1, grid=grid, flags=ec.NOTEBOOK)
        print("B", B, "grid", grid, "ok worst", w.get("grad"), flush=True)
    except AssertionError as e:
        print("B", B, "grid", grid, "FAIL", str(e)[:200], flush=True)
