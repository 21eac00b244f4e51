import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import engine_checks as ec
from microwakeword_amd import native
lib = native.NativeLib.get()
def grads(B, T, flags, fuse, split, grid, inline=1):
    om = ec.perturbed_inception_oracle(T, flags)
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 667, size=(B, T, 40)).astype(np.float32) * ec.SCALE).astype(np.float32)
    y = (rng.random(B) < 0.4).astype(np.float32); w = np.ones(B, np.float32)
    lay, eng = ec.make_inception_engine(lib, T, B, om, flags, fuse)
    eng.set_option("graph_role_split", split)
    eng.set_option("bn_inline", inline)
    if grid: eng.set_option("grid_graph", grid)
    eng.set_batch(x); eng.set_targets(y, w)
    eng.set_dropout_mask(np.ones((B, ec.eng_dense_inputs(lay)), np.uint8))
    eng.train_step(B, 1e-2, flags=native.STEP_NO_APPLY)
    g = eng.get_grads().copy(); eng.close()
    return lay, g
for name, flags in (("variant", ec.INC_VARIANT), ("default", ec.INC)):
    for B, T in ((9, 150), (16, 150)):
        lay, ref = grads(B, T, flags, True, 0, 0)
        mx = np.abs(ref).max()
        for split, grid, inline in ((0, 4, 1), (0, 2, 1), (1, 0, 1), (0, 4, 0), (0, 3, 0)):
            _, g = grads(B, T, flags, True, split, grid, inline)
            d = np.abs(g - ref); off = 0; bad = []
            for nm, n in lay.segments():
                if d[off:off + n].max() > 1e-5 * mx: bad.append(nm)
                off += n
            print("%s B=%d split=%d grid=%d inline=%d: max rel diff %.2e  first bad tensors %s" % (name, B, split, grid, inline, d.max() / mx, bad[-3:]), flush=True)
