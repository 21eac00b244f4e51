"""Ad-hoc larger fuzz sweep on the GPU (same checks as the -m gpu tests, more cases).
usage: python tools/gpu_big_fuzz.py [data cases [topology cases [seed offset [inception topology cases]]]]"""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import engine_checks as ec   # noqa: E402
from microwakeword_amd import native   # noqa: E402

lib = native.NativeLib.get()
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.time()
ec.check_data_fuzz(lib, cases=int(sys.argv[1]) if len(sys.argv) > 1 else 120, first=500 + off)
print("data fuzz ok", round(time.time() - t0, 1), flush=True)
ec.check_gather_fuzz(lib, cases=300, first=1000 + off)
print("gather fuzz ok", round(time.time() - t0, 1), flush=True)
ec.check_topology_fuzz(lib, cases=int(sys.argv[2]) if len(sys.argv) > 2 else 40, first=300 + off)
print("topology fuzz ok", round(time.time() - t0, 1), flush=True)
if len(sys.argv) > 4:
    ec.check_inception_topology_fuzz(lib, cases=int(sys.argv[4]), first=100 + off)
    print("inception topology fuzz ok", round(time.time() - t0, 1), flush=True)
