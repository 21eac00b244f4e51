#!/usr/bin/env python
"""Run on a machine that has the reference and TensorFlow/Keras (h5py comes with them):

    PYTHONPATH=<kahrendt/microWakeWord checkout> python tools/keras_weights_to_npz.py \
        --training_config trained_models/x/training_config.yaml --weights trained_models/x/best_weights.weights.h5 mixednet [model flags]

Builds the reference's own Keras model for the given flags (microwakeword/mixednet.py:278-386 or inception.py:232-340),
loads the ``.weights.h5`` checkpoint train.py writes (train.py:336-338,448-451) and stores its variables IN THE ORDER THE
REFERENCE'S BUILDER CREATES THEM - the order ``microwakeword_amd.model.Model.set_weights`` takes; ``model.get_weights()`` lists
layers by graph depth, which differs for residual blocks and Inception: tools/keras_creation_order.py - as the ``.npz`` twin
``<weights>.npz`` that ``Model.load_weights`` reads: keys ``%03d:<variable path>``.  The inverse is
``tools/npz_to_keras_weights.py``.  Nothing here imports microwakeword_amd."""
import argparse
import sys

import numpy as np

from keras_creation_order import creation_permutation, layer_creation_log


def build(argv):
    """The reference's own model for these flags, built the way its CLI builds it for evaluation
    (model_train_eval.py:400,420-426: load_config -> model_module.model(flags, shape=training_input_shape, batch_size=1))."""
    import microwakeword.inception as inception
    import microwakeword.mixednet as mixednet
    from microwakeword import model_train_eval
    ap = argparse.ArgumentParser()
    ap.add_argument("--training_config", required=True)
    ap.add_argument("--weights", required=True)
    sub = ap.add_subparsers(dest="model_name", required=True)
    mixednet.model_parameters(sub.add_parser("mixednet"))
    inception.model_parameters(sub.add_parser("inception"))
    flags = ap.parse_args(argv)
    module = {"mixednet": mixednet, "inception": inception}[flags.model_name]
    config = model_train_eval.load_config(flags, module)
    import tensorflow as tf
    with layer_creation_log(tf.keras.layers.Layer) as created:
        model = module.model(flags, shape=config["training_input_shape"], batch_size=1)
    return flags, model, creation_permutation(model.weights, created)


if __name__ == "__main__":
    flags, model, perm = build(sys.argv[1:])
    model.load_weights(flags.weights)
    have = model.get_weights()
    ws = [have[i] for i in perm]                                   # creation order
    names = [getattr(model.weights[i], "path", getattr(model.weights[i], "name", "v%d" % i)) for i in perm]
    if perm != sorted(perm):
        print("note: model.weights is not in creation order for this topology (%d of %d positions differ)" % (sum(a != b for a, b in enumerate(perm)), len(perm)))
    out = flags.weights + ".npz"
    np.savez(out, **{"%03d:%s" % (i, n): w for i, (n, w) in enumerate(zip(names, ws))})
    print("%d variables, %d values -> %s" % (len(ws), sum(w.size for w in ws), out))
