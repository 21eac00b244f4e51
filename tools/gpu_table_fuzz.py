"""Random flag sets out of the round-5 shape table (csrc/block_launch.hip.h: widths 32 / 48 / 64 in any order, odd depthwise kernels
3..23 with MixConv groups, conv1 kernel 3 / 5, first depthwise 3 / 5 / 7, stride 1 / 2 / 3, 2..5 blocks) with random (frames, batch,
grid) sizes: every case must land on the specialised block kernels and match the float64 oracle (check_train_steps).
usage: python tools/gpu_table_fuzz.py <first case> <cases>      (MWW_HIP_LIB=tests/hipemu/libmww_emu.so runs it on the emulator)"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import engine_checks as ec   # noqa: E402
from microwakeword_amd import mixednet, native   # noqa: E402
from oracle import model_oracle as mo   # noqa: E402


def random_table_flags(seed):
    rng = np.random.default_rng(9000 + seed)
    nb = int(rng.integers(2, 6))
    pf = [int(rng.choice([32, 48, 64])) for _ in range(nb)]
    ks = []
    for i in range(nb):
        kmax = int(rng.choice([3, 5, 7])) if i == 0 else int(rng.choice(list(range(3, 24, 2))))
        n = int(rng.choice([1, 1, 2, 3]))
        smaller = [k for k in range(1, kmax, 2)]
        groups = sorted(int(v) for v in rng.choice(smaller, size=min(n - 1, len(smaller)), replace=False)) if n > 1 and smaller else []
        ks.append(groups + [kmax])
    flags = dict(mo.MIXEDNET_DEFAULTS, pointwise_filters=",".join(map(str, pf)), repeat_in_block=",".join(["1"] * nb),
                 residual_connection=",".join(["0"] * nb), mixconv_kernel_sizes=",".join(str(k) for k in ks),
                 first_conv_filters=32, first_conv_kernel_size=int(rng.choice([3, 5])), stride=int(rng.choice([1, 1, 2, 3])))
    need = flags["first_conv_kernel_size"] + flags["stride"] * (sum(k[-1] - 1 for k in ks) + 1)
    T = int(rng.integers(need + 2, need + 260))
    if flags["stride"] > 1 and rng.random() < 0.4:
        # tail mode on purpose (fwd_first_body.inc "tail rows"): a window of 64 + 1 .. 64 + K - 1 first-conv rows is ONE tile whose
        # rows behind the 64th ride along - the lengths the round-5 / round-6 tail k-step bugs needed
        ta = 64 + int(rng.integers(1, ks[0][-1]))
        t_tail = (ta - 1) * flags["stride"] + flags["first_conv_kernel_size"] + int(rng.integers(0, flags["stride"]))
        if t_tail >= need + 2:
            T = t_tail
    return flags, T, int(rng.integers(1, 40)), int(rng.choice([0, 1, 2, 3, 5, 8]))


if __name__ == "__main__":
    lib = native.NativeLib.get()
    first, n = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for case in range(first, first + n):
        flags, T, B, grid = random_table_flags(case)
        tag = "case %d: filters %s kernels %s conv1 %d stride %d T %d B %d grid %d" % (
            case, flags["pointwise_filters"], flags["mixconv_kernel_sizes"], flags["first_conv_kernel_size"], flags["stride"], T, B, grid)
        try:
            fam = mixednet.kernel_family(flags, T, lib=lib)
            assert fam[0] == "block", fam
            ec.check_train_steps(lib, B=B, T=T, steps=1, grid=grid, flags=flags)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("FAIL", tag, type(e).__name__, str(e)[:300], flush=True)
    print("table fuzz cases %d..%d done, failures: %d" % (first, first + n - 1, bad))
