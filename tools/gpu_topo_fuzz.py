"""Topology fuzz with per-case reporting (which random MixedNet flag sets fail the parity check)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import engine_checks as ec   # noqa: E402
from microwakeword_amd import native   # noqa: E402

lib = native.NativeLib.get()
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for case in range(first, first + n):
    flags = ec.random_mixednet_flags(case)
    try:
        ec.check_graph_mixednet(lib, flags, B=3, T=70, steps=1, grid=2)
    except ValueError as e:
        if "too short" in str(e) or "at least 4 frames" in str(e):
            continue
        raise
    except AssertionError as e:
        bad += 1
        print("FAIL case", case, str(e)[:200])
        print("   ", {k: flags[k] for k in ("pointwise_filters", "repeat_in_block", "mixconv_kernel_sizes", "residual_connection",
                                             "first_conv_filters", "first_conv_kernel_size", "stride", "spatial_attention", "pooled", "max_pool")}, flush=True)
print("done, failures:", bad)
