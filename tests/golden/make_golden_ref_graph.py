"""Freezes what the REFERENCE'S OWN model builders compute into tests/golden/ref_graph_golden.npz (build container only).

``oracle/ref_model_shim.py`` executes ``/root/reference/microwakeword/mixednet.py`` / ``inception.py`` (and the layer files they
import) unchanged over stand-in Keras layer primitives in torch float64.  For every case below the fixture holds the inputs
(batch, labels, sample weights, dropout keep-mask, every variable's value in the order the reference creates them) and what
the reference's graph returns on them: inference and training-mode probabilities, the training-mode logits, the loss, the
gradient of the loss for every trainable variable and the BatchNorm moving statistics after the step.  The loss on top of the
logits is the oracle's statement of train.py:206,288-299 (``model_oracle.weighted_loss``) — the graph is the reference's, the loss
is not.  The fixture is data; the reference's sources do not travel.

    python tests/golden/make_golden_ref_graph.py            # needs /root/reference
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_model import CASES  # noqa: E402  (the same four topologies the frozen-oracle fixture uses)
from oracle import model_oracle as mo  # noqa: E402

FIXTURE = os.path.join(HERE, "ref_graph_golden.npz")
BATCH = {"mixednet_default": 4, "mixednet_notebook": 3, "mixednet_residual_heads": 3, "inception_default": 3}


def case_inputs(name):
    """-> (kind, flags, T, variable values in Keras order, x, y, w, keep-mask or None): seeded, reproducible without the reference."""
    kind, flags, T = CASES[name]
    B = BATCH[name]
    om = mo.OracleModel(kind, flags, T, seed=42)                       # only its initial values (numpy, glorot / BN defaults) are used
    rng = np.random.default_rng(1234 + len(name))
    values = []
    for v in om.vars:                                                  # BN / bias values that matter (defaults are 1 / 0)
        a = v.value
        if v.name.endswith(("bias", "beta", "moving_mean")):
            a = a + rng.normal(0, 0.1, a.shape).astype(np.float32)
        if v.name.endswith(("gamma", "moving_variance")):
            a = a + np.abs(rng.normal(0, 0.2, a.shape)).astype(np.float32)
        values.append(a.astype(np.float32))
    x = (rng.integers(0, 667, size=(B, T, 40)).astype(np.float32) * np.float32(0.0390625)).astype(np.float32)
    y = (rng.random(B) < 0.5).astype(np.float32)
    y[0], y[-1] = 1.0, 0.0
    w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
    keep = None
    if kind == "inception" and flags.get("dropout", 0) > 0:
        keep = (rng.random((B, values[-2].shape[0])) >= flags["dropout"]).astype(np.float32)
    return kind, flags, T, values, x, y, w, keep


def run_case(name):
    from oracle import ref_model_shim as rm
    kind, flags, T, values, x, y, w, keep = case_inputs(name)
    out = {"x": x, "y": y, "w": w}
    if keep is not None:
        out["keep"] = keep
    ev = rm.run_reference_model(kind, flags, x, values, training=False)
    out["p_eval"] = ev.probs.detach().reshape(-1).numpy()
    loss, p, grads, run = rm.reference_loss_and_grads(kind, flags, x, y, w, values, dropout_mask=keep, loss_fn=mo.weighted_loss)
    assert len(run.variables) == len(values)
    out["loss"], out["p_train"], out["z_train"] = np.float64(loss), p, run.logits.detach().reshape(-1).numpy()
    out["trainable"] = np.array([v.trainable for v in run.variables])
    out["created_as"] = np.array([v.name for v in run.variables])     # the stand-in's layer names, creation order (informational)
    for i, (v, g, val) in enumerate(zip(run.variables, grads, values)):
        out["value/%03d" % i] = val
        if v.trainable:
            out["grad/%03d" % i] = g.numpy().reshape(val.shape)
        elif v.updated is not None:
            out["moving/%03d" % i] = v.updated.numpy()
    return out


def build():
    blob = {}
    for name in CASES:
        for k, v in run_case(name).items():
            blob["%s/%s" % (name, k)] = v
    return blob


def load():
    return np.load(FIXTURE)


def main():
    blob = build()
    np.savez_compressed(FIXTURE, **blob)
    print("wrote", len(blob), "arrays,", os.path.getsize(FIXTURE), "bytes")


if __name__ == "__main__":
    main()
