"""Throughput of the PRODUCT train loop (microwakeword_amd.train.train on the package's own Model / FeatureHandler) on the
synthetic benchmark stores, next to bench.py's figure for the same step: what a user of the CLI gets.
usage: python tools/train_loop_throughput.py [steps] [batch] [mixednet|inception]"""
import random
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from microwakeword_amd import inception, mixednet, synthetic   # noqa: E402
from microwakeword_amd import train as tr   # noqa: E402
from microwakeword_amd.data import FeatureHandler   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = sys.argv[3] if len(sys.argv) > 3 else "mixednet"
T = 194
for verbose in (False, True):
    cfg, _ = synthetic.benchmark_config(4096, 1234, n_val=256, n_ambient=32)
    d = tempfile.mkdtemp(prefix="mww_loop_")
    cfg = dict(cfg, train_dir=d, summaries_dir=d + "/logs", batch_size=B, spectrogram_length=T, training_steps=[steps], learning_rates=[1e-3],
               time_mask_max_size=[5], time_mask_count=[2], freq_mask_max_size=[5], freq_mask_count=[2], positive_class_weight=[1.0],
               negative_class_weight=[1.0], eval_step_interval=steps, target_minimization=0.9, minimization_metric=None,
               maximization_metric="accuracy")
    random.seed(0)
    np.random.seed(0)
    if kind == "inception":
        model = inception.model(dict(synthetic.DEFAULT_INCEPTION_FLAGS), (T, 40), B, seed=42, max_batch=B)
    else:
        model = mixednet.model(synthetic.DEFAULT_MIXEDNET_FLAGS, (T, 40), B, seed=42, max_batch=B)
    fh = FeatureHandler(cfg, engine=model.engine)
    t0 = time.perf_counter()
    tr.train(model, cfg, fh, verbose=verbose)
    dt = time.perf_counter() - t0
    sys.stdout.write("\n")
    print("train.train (%s) verbose=%s: %d steps of batch %d in %.3f s = %.4f ms/step = %.2f M windows/s (one validation pass and the "
          "checkpoint writes at the end included)" % (kind, verbose, steps, B, dt, 1e3 * dt / steps, steps * B / dt / 1e6), flush=True)
    model.engine.close()
