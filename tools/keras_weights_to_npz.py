#!/usr/bin/env python
"""Run on a machine that has the reference and TensorFlow/Keras (h5py comes with them):

    PYTHONPATH=<kahrendt/microWakeWord checkout> python tools/keras_weights_to_npz.py \
        --training_config trained_models/x/training_config.yaml --weights trained_models/x/best_weights.weights.h5 mixednet [model flags]

Builds the reference's own Keras model for the given flags (microwakeword/mixednet.py:278-386 or inception.py:232-340),
loads the ``.weights.h5`` checkpoint train.py writes (train.py:336-338,448-451) and stores ``model.get_weights()`` — in
Keras order, which is the order ``microwakeword_amd.model.Model.set_weights`` takes — as the ``.npz`` twin
``<weights>.npz`` that ``Model.load_weights`` reads: keys ``%03d:<variable path>``.  The inverse is
``tools/npz_to_keras_weights.py``.  Nothing here imports microwakeword_amd."""
import argparse
import sys

import numpy as np


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
    return flags, module.model(flags, shape=config["training_input_shape"], batch_size=1)


if __name__ == "__main__":
    flags, model = build(sys.argv[1:])
    model.load_weights(flags.weights)
    ws = model.get_weights()
    names = [getattr(v, "path", getattr(v, "name", "v%d" % i)) for i, v in enumerate(model.weights)]
    out = flags.weights + ".npz"
    np.savez(out, **{"%03d:%s" % (i, n): w for i, (n, w) in enumerate(zip(names, ws))})
    print("%d variables, %d values -> %s" % (len(ws), sum(w.size for w in ws), out))
