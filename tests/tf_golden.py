"""TEST INFRASTRUCTURE: consumers of tests/golden/tf_golden.npz, the golden vectors tools/make_tf_golden.py writes on a machine
that has TensorFlow and the reference (neither is available in the build container: the file may be absent, the tests then
skip).  The file is the only thing that can pin the MODEL half of the path to the reference (SURVEY §8c "parity unpinned"):
weights in `get_weights()` order, `model(x, training=False)`, the gradient / loss / metric list / updated weights of one
`train_on_batch` with the reference's [B,B] sample weights.  `synthesize` writes a file of the same schema from the oracle
itself, so that the consumers below stay exercised (that proves the plumbing, not parity)."""
import os

import numpy as np

from oracle import model_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_golden.npz")
MODES = ("per_sample", "keras_last_axis", "keras_first_axis")


def _val(s):
    for conv in (int, float):
        try:
            return conv(s)
        except ValueError:
            pass
    return s


def case_setup(z, case):
    """-> (kind, flags, T, weights list, x, y) of one case of the file."""
    args = [str(a) for a in z[case + "/flags"]]
    kind = "inception" if case.startswith("inception") else "mixednet"
    flags = dict(mo.INCEPTION_DEFAULTS if kind == "inception" else mo.MIXEDNET_DEFAULTS)
    for k, v in zip(args[0::2], args[1::2]):
        flags[k.lstrip("-")] = _val(v)
    n = len([k for k in z.files if k.startswith(case + "/w0/")])
    w0 = [z["%s/w0/%03d" % (case, i)] for i in range(n)]
    return kind, flags, int(z[case + "/frames"]), w0, z[case + "/x"], z[case + "/y"]


def combined(z, case, tag, mode):
    from microwakeword_amd.model import combine_weights
    cw = z["%s/%s/class_weights" % (case, tag)]
    return combine_weights(z["%s/%s/penalty" % (case, tag)], z[case + "/y"], float(cw[0]), float(cw[1]), mode)


def trainable_index(z, case, tag):
    """position in the full weight list of every trainable variable of the file, in the file's gradient order"""
    names = [str(n) for n in z[case + "/names"]]
    return [names.index(str(t)) for t in z["%s/%s/trainable_names" % (case, tag)]]


def check_oracle(z, case, fwd_tol=1e-3):
    """oracle/model_oracle.py against one case.  Returns the sample-weight reading (MODES) the reference's numbers follow."""
    kind, flags, T, w0, x, y = case_setup(z, case)
    om = mo.OracleModel(kind, flags, T, seed=0)
    assert [tuple(v.value.shape) for v in om.vars] == [tuple(w.shape) for w in w0], "Keras get_weights() order / shapes differ from the oracle's"
    om.set_weights(w0)
    assert np.abs(om.predict(x) - z[case + "/p_eval"]).max() <= fwd_tol
    # uniform weights: loss, probabilities, every gradient tensor, the Adam step and the BN moving statistics
    tr = [v.name for v in om.vars if v.trainable]
    idx = trainable_index(z, case, "uniform")
    assert len(idx) == len(tr)
    loss, p, grads, _ = om.loss_and_grads(x, y, combined(z, case, "uniform", "per_sample"))
    assert abs(loss - float(z[case + "/uniform/tape_loss"])) <= 1e-5 * max(1.0, abs(loss))
    assert np.abs(p - z[case + "/uniform/p_train"]).max() <= fwd_tol
    by_pos = {i: n for i, n in enumerate(v.name for v in om.vars)}
    for gi, pos in enumerate(idx):
        ref = z["%s/uniform/grad/%03d" % (case, gi)].astype(np.float64)
        got = grads[by_pos[pos]].numpy().reshape(ref.shape)
        assert np.linalg.norm(got - ref) <= 1e-3 * max(np.linalg.norm(ref), 1e-6), (case, by_pos[pos])
    om.train_step(x, y, combined(z, case, "uniform", "per_sample"), 1e-3)
    for i, v in enumerate(om.vars):
        ref = z["%s/uniform/w1/%03d" % (case, i)]
        tol = 0.05 * 1e-3 if v.trainable else 1e-5 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(v.value - ref).max() <= tol, (case, v.name)
    # non-uniform penalty x class weights: which reading of the [B,B] weight does Keras follow?
    matches = []
    for mode in MODES:
        om.set_weights(w0)
        loss, _, _, _ = om.loss_and_grads(x, y, combined(z, case, "weighted", mode))
        if abs(loss - float(z[case + "/weighted/tape_loss"])) <= 1e-4 * max(1.0, abs(loss)):
            matches.append(mode)
    assert matches, "none of the sample-weight readings reproduces the reference's weighted loss"
    return matches


def synthesize(path, cases=("mixednet_default",), mode="per_sample", batch=6):
    """A file of the schema tools/make_tf_golden.py writes, filled in by the oracle (plumbing test only)."""
    from microwakeword_amd.model import combine_weights
    flags_of = {"mixednet_default": ("mixednet", 194, ["--residual_connection", "0,0,0,0"]),
                "inception_default": ("inception", 194, ["--dropout", "0.0"])}
    blob = {"cases": np.array(list(cases)), "batch": np.int64(batch), "tensorflow_version": np.array("synthetic (oracle)")}
    rng = np.random.default_rng(5)
    for case in cases:
        kind, T, extra = flags_of[case]
        flags = dict(mo.INCEPTION_DEFAULTS if kind == "inception" else mo.MIXEDNET_DEFAULTS)
        for k, v in zip(extra[0::2], extra[1::2]):
            flags[k.lstrip("-")] = _val(v)
        om = mo.OracleModel(kind, flags, T, seed=3)
        w0 = om.get_weights()
        names = [v.name for v in om.vars]
        x = (rng.integers(0, 667, size=(batch, T, 40)).astype(np.float32) * np.float32(0.0390625)).astype(np.float32)
        y = (rng.random(batch) < 0.5).astype(np.float64)
        b = {"names": np.array(names), "x": x, "y": y, "frames": np.int64(T), "flags": np.array(extra), "p_eval": om.predict(x)}
        for i, w in enumerate(w0):
            b["w0/%03d" % i] = w
        for tag, pen, cw in (("uniform", np.ones(batch), (1.0, 1.0)), ("weighted", rng.choice([0.5, 1.0, 2.0], size=batch), (20.0, 1.0))):
            om.set_weights(w0)
            om.adam = None
            w = combine_weights(pen, y, cw[0], cw[1], mode if tag == "weighted" else "per_sample")
            loss, p, grads, _ = om.loss_and_grads(x, y, w)
            b[tag + "/penalty"], b[tag + "/class_weights"] = np.asarray(pen, np.float64), np.array(cw)
            b[tag + "/tape_loss"], b[tag + "/p_train"] = np.float64(loss), p
            tr = [v.name for v in om.vars if v.trainable]
            b[tag + "/trainable_names"] = np.array(tr)
            for i, n in enumerate(tr):
                b["%s/grad/%03d" % (tag, i)] = grads[n].numpy().astype(np.float32)
            om.train_step(x, y, w, 1e-3)
            for i, v in enumerate(om.vars):
                b["%s/w1/%03d" % (tag, i)] = v.value.copy()
        blob.update({"%s/%s" % (case, k): v for k, v in b.items()})
    np.savez_compressed(path, **blob)
    return path


def check_engine(lib, z, case, fwd_tol=1e-3):
    """The HIP engine (through the C ABI) against one case of the file: inference forward within the north-star tolerance,
    then the uniform-weight train step - loss, probabilities, flat gradient, Adam-updated parameters, BN moving statistics."""
    from microwakeword_amd import native
    from microwakeword_amd.layout import InceptionLayout, MixedNetLayout
    kind, flags, T, w0, x, y = case_setup(z, case)
    B = x.shape[0]
    lay = (InceptionLayout if kind == "inception" else MixedNetLayout)(flags, T)
    eng = native.Engine(lib=lib, **lay.engine_args(B))
    eng.set_grad_mask(lay.grad_mask())
    p0, s0 = lay.pack(w0)
    eng.set_params(p0)
    eng.set_bn_state(s0)
    eng.set_batch(x)
    eng.forward(B, training=False)
    assert np.abs(eng.read_outputs(B, want_loss=False)[0] - z[case + "/p_eval"]).max() <= fwd_tol
    w = combined(z, case, "uniform", "per_sample")
    eng.set_targets(y.astype(np.float32), w)
    eng.train_step(B, 1e-3)
    pr, _, loss = eng.read_outputs(B)
    ref_loss = float(z[case + "/uniform/tape_loss"])
    assert abs(loss - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    assert np.abs(pr - z[case + "/uniform/p_train"]).max() <= fwd_tol
    idx = trainable_index(z, case, "uniform")
    full = [np.zeros_like(a) for a in w0]
    for gi, pos in enumerate(idx):
        full[pos] = z["%s/uniform/grad/%03d" % (case, gi)].reshape(w0[pos].shape)
    gref = lay.pack(full)[0]
    g = eng.get_grads()
    assert np.linalg.norm(g - gref) <= 2e-3 * np.linalg.norm(gref)
    n = len(w0)
    p1, s1 = lay.pack([z["%s/uniform/w1/%03d" % (case, i)] for i in range(n)])
    well = np.abs(gref) > 1e-4 * np.abs(gref).max()
    assert np.abs(eng.get_params() - p1)[well].max() <= 0.05 * 1e-3
    assert np.abs(eng.get_bn_state() - s1).max() <= 1e-5 * max(1.0, float(np.abs(s1).max()))
    eng.close()
