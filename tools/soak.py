"""Soak / reproducibility run on the GPU: two engines, same seeds, N train steps each on device-assembled batches
(default options: fused input, statistics hand-over through fp64 atomics), periodic validation forwards in between.
Reports whether parameters, BN state and Adam moments are bit-identical between the runs and that the loss stays finite."""
import hashlib
import random
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microwakeword_amd import mixednet, synthetic   # noqa: E402
from microwakeword_amd.data import FeatureHandler   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B, T = 1024, 194


def run():
    random.seed(0)
    np.random.seed(0)
    model = mixednet.model(dict(synthetic.DEFAULT_MIXEDNET_FLAGS), (T, 40), B, seed=42, max_batch=B)
    eng = model.engine
    cfg, _ = synthetic.benchmark_config(4096, 1234, n_val=2048, n_ambient=64)
    fh = FeatureHandler(cfg, engine=eng)
    fh.use_private_rng()
    t0 = time.perf_counter()
    for k in range(N):
        fh.next_training_batch_on_device(B, T, "default", synthetic.SPEC_AUGMENT_POLICY)
        eng.train_step(B, 1e-3)
        if k % 500 == 499:
            fh.evaluate_on_device(model, "validation", T, "truncate_start", batch_size=1024)
    eng.synchronize()
    dt = time.perf_counter() - t0
    loss = eng.read_outputs(B)[2]
    m, v, step = eng.get_opt_state()
    h = hashlib.sha256()
    for a in (eng.get_params(), eng.get_bn_state(), m, v):
        h.update(np.ascontiguousarray(a).tobytes())
    eng.close()
    return h.hexdigest(), float(loss), dt, int(step)


a = run()
b = run()
print("run 1:", a)
print("run 2:", b)
print("bit-identical:", a[0] == b[0], " loss finite:", np.isfinite(a[1]) and np.isfinite(b[1]))
sys.exit(0 if (a[0] == b[0] and np.isfinite(a[1])) else 1)
