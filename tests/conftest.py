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
    from microwakeword_amd import build_native
    defines = os.environ.get("MWW_EMU_DEFINES", "").split()   # kernel-variant builds (-DMWW_...=..), into a file of their own
    if defines:
        return build_native.build_emulator(EMU.replace(".so", "_variant.so"), defines=defines)
    return build_native.build_emulator(EMU)


@pytest.fixture(scope="session")
def emu_lib():
    from microwakeword_amd import native
    path = build_emulator_lib()
    if path is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    return native.NativeLib(path)
