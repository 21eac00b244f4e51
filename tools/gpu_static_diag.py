"""Diagnostic: how far apart are Inception runs that differ only in the float32 summation order of their BN statistics?
Static-shape kernels (the stem sums per accumulator column) against the run-time-shape kernels, and - as the yardstick - the
run-time-shape kernels on two different grids (other partial sums, same kernels).  Prints the relative L2 distance of every
recorded array (probabilities / gradients per step, parameters, BN state) from the first run; the middle step runs a batch of
two windows (check_inception_static_shapes_are_schedule_only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import engine_checks as ec
from microwakeword_amd import native

lib = native.NativeLib()
B, steps = int(os.environ.get("DIAG_B", "67")), 3
for T in (100, 194):
    rng = np.random.default_rng(17)
    om = ec.perturbed_inception_oracle(T, ec.INC)
    x = (rng.integers(0, 667, size=(steps, B, T, 40)).astype(np.float32) * ec.SCALE).astype(np.float32)
    y = (rng.random((steps, B)) < 0.4).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
    outs = []
    runs = (("dynamic", 0, 0), ("static", 1, 0), ("dynamic, grid 300", 0, 300), ("static, grid 300", 1, 300))
    for name, static, grid in runs:
        lay, eng = ec.make_inception_engine(lib, T, B, om, ec.INC)
        eng.set_option("graph_static_shapes", static)
        if grid:
            eng.set_option("grid_graph", grid)
        got = []
        for k in range(steps):
            nb = B if k != 1 else min(B, 2)
            eng.set_batch(x[k][:nb]); eng.set_targets(y[k][:nb], w[:nb])
            eng.set_dropout_mask(np.ones((nb, ec.eng_dense_inputs(lay)), np.uint8))
            eng.train_step(nb, 1e-2)
            got.append(("probs%d" % k, eng.read_outputs(nb)[0].copy()))
            got.append(("grads%d" % k, eng.get_grads().copy()))
        got += [("params", eng.get_params().copy()), ("bn", eng.get_bn_state().copy())]
        outs.append(got); eng.close()
    for (name, _, _), other in zip(runs[1:], outs[1:]):
        print("T %d  %-20s vs dynamic: " % (T, name) + "  ".join(
            "%s %.1e" % (n, np.linalg.norm(b.astype(np.float64) - a) / max(np.linalg.norm(a.astype(np.float64)), 1e-30))
            for (n, a), (_, b) in zip(outs[0], other)))
    print("T %d  |grads| per step (dynamic): " % T + "  ".join("%.2e" % np.linalg.norm(a) for n, a in outs[0] if n.startswith("grads")))
