"""Per-tensor gradient error of one un-imposed train step of the Inception VARIANT topology against the float64 oracle, at several
batch sizes and engine options (debugging aid for tests/test_engine_gpu.py::test_gradients_without_imposed_masks_on_the_remaining_paths)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import engine_checks as ec  # noqa: E402
from microwakeword_amd import native  # noqa: E402


def run(lib, B, T, flags, options):
    rng = np.random.default_rng(11)
    x = ec.synth_x(rng, B, T)
    y = (rng.random(B) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
    om = ec.perturbed_inception_oracle(T, flags)
    lay, eng = ec.make_inception_engine(lib, T, B, om, flags)
    for k, v in options.items():
        eng.set_option(k, v)
    keep = (rng.random((B, lay.t_last * lay.c_last)) >= flags["dropout"]).astype(np.float32)
    eng.set_dropout_mask(keep)
    eng.set_batch(x)
    eng.set_targets(y, w)
    eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
    pr, z, loss = eng.read_outputs(B)
    lo, po, grads, _ = om.loss_and_grads(x, y, w, dropout_mask=keep)
    gref = lay.pack([grads[n].numpy().astype(np.float32) if kd == "param" else np.zeros(sh, np.float32) for n, sh, kd in lay.keras_vars])[0]
    g = eng.get_grads()
    scale = max(1e-6, float(np.abs(gref).max()))
    off, rows = 0, []
    for name, n in lay.segments():
        a, r = g[off:off + n], gref[off:off + n]
        den = max(np.linalg.norm(r), 1e-3 * scale * np.sqrt(n))
        rows.append((float(np.linalg.norm(a - r) / den), name, n, float(np.linalg.norm(r))))
        off += n
    eng.close()
    rows.sort(reverse=True)
    print("B=%d T=%d options=%s: loss err %.2e, prob err %.2e; worst tensors:" % (B, T, options, abs(loss - lo), float(np.abs(pr - po).max())))
    for l2, name, n, nr in rows[:6]:
        print("   %-28s n=%-6d |ref|=%.3e  rel L2 err %.3e" % (name, n, nr, l2))


if __name__ == "__main__":
    lib = native.NativeLib.get()
    for B in (32, 128, 512):
        run(lib, B, 194, ec.INC_VARIANT, {})
    run(lib, 512, 194, ec.INC_VARIANT, {"bn_inline": 0})
    run(lib, 512, 194, ec.INC_VARIANT, {"graph_static_shapes": 0})
    run(lib, 512, 194, ec.INC_VARIANT, {"grid_graph": 256})
    run(lib, 512, 120, ec.INC_VARIANT, {})
    run(lib, 512, 194, ec.INC, {})
