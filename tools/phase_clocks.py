"""Profiling aid: per-phase shader-clock breakdown of the block kernels (thread 0 of every workgroup).
Runs a few eager train steps at the benchmark configuration with mww_set_option("ablate", 16)."""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microwakeword_amd import synthetic
from microwakeword_amd.data import FeatureHandler
from microwakeword_amd.model import Model

from microwakeword_amd import build_native, native
LIB = os.path.join(ROOT, "microwakeword_amd", "libmww_prof.so")   # bash tools/build_variant.sh prof -DMWW_PROFILE, built in the container
if not os.path.isfile(LIB):
    build_native.build_library(LIB, defines=["-DMWW_PROFILE"], slim=True)
B, T = int(os.environ.get("PC_B", "1024")), 194
model = Model(synthetic.DEFAULT_MIXEDNET_FLAGS, (T, 40), B, seed=42, max_batch=B, lib=native.NativeLib(LIB))
eng = model.engine
cfg, _ = synthetic.benchmark_config(1024, 1234)
random.seed(0); np.random.seed(0)
fh = FeatureHandler(cfg, engine=eng)
fh.use_private_rng()
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng.set_option("ablate", 16 | extra)
for opt in filter(None, os.environ.get("PC_OPTIONS", "").split(",")):   # e.g. PC_OPTIONS=pointwise_bf16=1 or storage_bf16=1,bwd_wide=0
    eng.set_option(opt.split("=")[0], int(opt.split("=")[1]))
print("== B = %d, options: %s" % (B, os.environ.get("PC_OPTIONS", "(fp32, wide backward)")))
for _ in range(3):
    fh.next_training_batch_on_device(B, T, "default", synthetic.SPEC_AUGMENT_POLICY)
    eng.train_step(B, 1e-3)
eng.synchronize()
FWD = ["commit+wait", "barrier1", "issue", "depthwise", "barrier2", "mfma", "stores", "barrier3"]
BWD = ["commit+wait", "barrier1", "issue+P1", "barrier2", "mfma", "barrier3", "P4", "barrier4"]
S = 12   # kClkSlots
for k in (2, 3, 4):
    for tag, names, grid in (("f", FWD, 1024), ("b", BWD, 512)):
        raw = eng.debug_read("clk%s%d" % (tag, k), 1, 2048 * S * 2)
        rec = raw.view(np.uint64).reshape(2048, S)[:min(grid, B)].astype(np.float64)
        clk, st = rec[:, :8], rec[:, 8:]
        tot = clk.sum(1)
        print("layer %d %s: total cycles/WG mean %.0f (min %.0f max %.0f)" % (k, "fwd" if tag == "f" else "bwd", tot.mean(), tot.min(), tot.max()))
        print("   " + "  ".join("%s=%.0f(%.0f%%)" % (n, v, 100 * v / tot.mean()) for n, v in zip(names, clk.mean(0))))
        if tag == "f":
            d = [(st[:, i + 1] - st[:, i]) for i in range(3)]
            print("   fwd prologue per WG (cycles, mean/max): entry->fold %.0f/%.0f  fold->weights %.0f/%.0f  weights->barrier %.0f/%.0f" %
                  (d[0].mean(), d[0].max(), d[1].mean(), d[1].max(), d[2].mean(), d[2].max()))
        else:
            print("   per WG: prologue %.0f  loop %.0f  epilogue %.0f cycles" % ((st[:, 1] - st[:, 0]).mean(), (st[:, 2] - st[:, 1]).mean(), (st[:, 3] - st[:, 2]).mean()))

# ---- who is slow?  loop time per workgroup grouped by XCD (blockIdx % 8) and by dispatch round-robin position
for k in (2, 4):
    raw = eng.debug_read("clkb%d" % k, 1, 2048 * S * 2)
    rec = raw.view(np.uint64).reshape(2048, S)[:min(512, B)].astype(np.float64)
    tot = rec[:, :8].sum(1)
    print("layer %d bwd loop cycles by XCD:" % k, [int(tot[x::8].mean()) for x in range(8)], " spread within XCD0: min %d max %d" % (tot[0::8].min(), tot[0::8].max()))
    half = len(tot) // 2
    print("   first half of the grid (first WG on each CU?) mean %d, second half mean %d" % (tot[:half].mean(), tot[half:].mean()))
    order = np.argsort(tot)
    print("   slowest 16 blockIdx:", order[-16:].tolist(), " fastest 16:", order[:16].tolist())
for k in (2, 3, 4):
    raw = eng.debug_read("clkf%d" % k, 1, 2048 * S * 2)
    rec = raw.view(np.uint64).reshape(2048, S)[:min(1024, B)].astype(np.float64)
    tot = rec[:, :8].sum(1)
    q = len(tot) // 4
    if q:
        print("layer %d fwd loop cycles by dispatch quartile (blockIdx >> 8):" % k, [int(tot[i * q:(i + 1) * q].mean()) for i in range(4)])
