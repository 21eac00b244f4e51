"""Freezes outputs of oracle/model_oracle.py on fixed seeds into tests/golden/model_oracle_golden.npz.

This does NOT pin the oracle to the reference (TensorFlow/Keras are not installable here, the reference ships no
fixtures: the model half stays "parity unpinned").  It guards the restatement against accidental
change between rounds: every number the HIP kernels are compared with comes from this file's producer.

    python tests/golden/make_golden_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

CASES = {
    "mixednet_default": ("mixednet", dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0"), 194),
    "mixednet_notebook": ("mixednet", dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0", first_conv_kernel_size=5, stride=3,
                                           pointwise_filters="64,64,64,64", mixconv_kernel_sizes="[5],[7,11],[9,15],[23]"), 204),
    "mixednet_residual_heads": ("mixednet", dict(mo.MIXEDNET_DEFAULTS, residual_connection="1,0,1,0", repeat_in_block="1,2,1,1",
                                                 spatial_attention=1, pooled=1, max_pool=1), 194),
    "inception_default": ("inception", dict(mo.INCEPTION_DEFAULTS), 194),
}


def run(kind, flags, T):
    om = mo.OracleModel(kind, flags, T, seed=42)
    rng = np.random.default_rng(7)
    B = 4
    x = (rng.integers(0, 667, size=(B, T, 40)).astype(np.float32) * np.float32(0.0390625)).astype(np.float32)
    y = np.array([1, 0, 0, 1], np.float32)
    w = np.array([1.0, 0.5, 2.0, 1.0], np.float32)
    keep = None
    if kind == "inception":
        n = (T - mo.inception_slices_dropped(flags)) * 16
        keep = (rng.random((B, n)) >= flags["dropout"]).astype(np.float32)
    out = {"p_eval": om.predict(x)}
    kw = {"dropout_mask": keep} if keep is not None else {}
    loss, p, grads, _ = om.loss_and_grads(x, y, w, **kw)
    out["loss"] = np.float64(loss)
    out["p_train"] = p
    out["grad_norms"] = np.array([float(g.norm()) for g in grads.values()])
    om.train_step(x, y, w, 1e-3, **kw)
    out["weights_after_sum"] = np.array([float(np.sum(v.value.astype(np.float64))) for v in om.vars])
    return out


def main():
    blob = {}
    for name, (kind, flags, T) in CASES.items():
        for k, v in run(kind, flags, T).items():
            blob["%s/%s" % (name, k)] = v
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_oracle_golden.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
