import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # never drop __pycache__ next to the read-only reference


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


EMU = os.path.join(ROOT, "tests", "hipemu", "libmww_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emulator_lib():
    """TEST INFRASTRUCTURE: compiles the unchanged product HIP sources as host C++ against tests/hipemu
    (fibers + emulated wave ops) and returns the path of the resulting library, or None without clang++."""
    import subprocess
    csrc = os.path.join(ROOT, "microwakeword_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("mww_lib.hip", "sampler.cpp")] + [os.path.join(ROOT, "tests", "hipemu", "hipemu.cpp")]
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc)]
    deps += [os.path.join(ROOT, "include", "mww.h"), os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h")]
    if not os.path.isfile(CLANG):
        return None
    if not os.path.isfile(EMU) or any(os.path.getmtime(d) > os.path.getmtime(EMU) for d in deps):
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", os.path.join(ROOT, "tests", "hipemu"),
               "-I", os.path.join(ROOT, "include"), "-Wno-unused-value", "-pthread"] + srcs + ["-o", EMU, "-ldl"]
        subprocess.run(cmd, check=True)
    return EMU


@pytest.fixture(scope="session")
def emu_lib():
    from microwakeword_amd import native
    path = build_emulator_lib()
    if path is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    return native.NativeLib(path)
