"""The Inception stem gathering from the feature stores against the materialised batch on random feature sets / policies /
window lengths (tests/engine_checks.py::check_inception_gathered_stem), more cases than the GPU suite runs."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import engine_checks as ec
from microwakeword_amd import native
lib = native.NativeLib()
t = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ec.check_inception_gathered_stem(lib, cases=n, first=100, B=37, lengths=(194, 176, 150, 200, 97, 201), graphs=(0, 1))
ec.check_inception_gathered_stem(lib, cases=8, first=200, B=300, lengths=(194, 176), grid=64)
print("stem gather fuzz: %d + 8 cases ok in %.1f s" % (n, time.time() - t))
