#!/usr/bin/env python
"""The hand-off to the reference's export path (train.py:448-462, model_train_eval.py:420-426, utils.convert_model_saved):
run on a machine that has the reference and TensorFlow/Keras,

    PYTHONPATH=<kahrendt/microWakeWord checkout> python tools/npz_to_keras_weights.py \
        --training_config trained_models/x/training_config.yaml --weights trained_models/x/best_weights.weights.h5 mixednet [model flags]

reads ``<weights>.npz`` (what ``microwakeword_amd.model.Model.save_weights`` wrote: every Keras variable in the order the
reference's builder creates them, keys ``%03d:<name>``), builds the reference's own Keras model for the same flags, maps creation
order to ``model.weights`` order (tools/keras_creation_order.py: they differ for residual blocks and Inception), checks the
shapes variable by variable, ``set_weights`` and writes the ``.weights.h5`` the reference's converter loads."""
import sys

import numpy as np

from keras_weights_to_npz import build

if __name__ == "__main__":
    flags, model, perm = build(sys.argv[1:])
    z = np.load(flags.weights + ".npz")
    created = [z[k] for k in sorted(z.files)]                      # creation order
    have = model.get_weights()
    if len(created) != len(have):
        raise SystemExit("the twin holds %d variables, the Keras model %d" % (len(created), len(have)))
    ws = [None] * len(have)
    for k, i in enumerate(perm):                                   # creation position k is model.weights[i]
        v = model.weights[i]
        if created[k].shape != have[i].shape:
            raise SystemExit("variable %d (%s): twin %s vs Keras %s" % (k, getattr(v, "path", v.name), created[k].shape, have[i].shape))
        ws[i] = created[k]
    model.set_weights(ws)
    model.save_weights(flags.weights)
    print("%d variables -> %s" % (len(ws), flags.weights))
