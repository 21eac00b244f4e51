#!/usr/bin/env python
"""Golden vectors of the MODEL half of the path from the reference itself.  Run on a machine that has TensorFlow (>= 2.16,
i.e. Keras 3 - setup.py:18 of the reference) and a checkout of kahrendt/microWakeWord:

    PYTHONPATH=<microWakeWord checkout> python tools/make_tf_golden.py [--out tests/golden/tf_golden.npz]

and commit the file it writes.  Nothing here imports microwakeword_amd or oracle/: the numbers are the reference's own.
tests/test_tf_golden.py (oracle, CPU) and tests/test_engine_gpu.py::test_tf_golden_vectors (HIP engine) consume the file
when it is present and skip otherwise.  TensorFlow is not installable in the build container, so this script has not been
executed there; it only uses public Keras API and the reference's own entry points:

  * the model is built by the reference: mixednet.model / inception.model (microwakeword/mixednet.py:278-386,
    inception.py:232-340) from its own argparse defaults (model_parameters) plus the listed overrides;
  * weights: seeded numpy values in `model.get_weights()` order (Keras layer-creation order), installed with
    `model.set_weights` - stored in the file, so the consumer needs no initialiser parity;
  * forward: `model(x, training=False)` (the "TF non-streaming forward" of BASELINE.json's north_star, tolerance 1e-3);
  * one train step exactly as train.py does it: compile with BinaryCrossentropy(from_logits=False), Adam(), the nine
    metrics (train.py:206-223), learning rate assigned (train.py:265), `train_on_batch(x, y[B,1], sample_weight=w[B] *
    vectorize(class_weights.get)(y[B,1]))` - the reference's [B,B] broadcast, train.py:288-293 - once with uniform and
    once with non-uniform penalty / class weights (what Keras makes of the [B,B] weight is the open question of SURVEY
    A.5), on a batch that contains saturated logits (the dense kernel is scaled up);
  * the gradient of the same step through `model.compute_loss` under a GradientTape (the call Keras' train_step makes),
    the returned metric list, and the weights after the step (Adam: lr, epsilon placement, bias correction; BN moving
    statistics: momentum, biased variance).
"""
import argparse
import sys

import numpy as np

CASES = {
    # name: (module, frames, extra command-line flags of the model parser)
    "mixednet_default": ("mixednet", 194, ["--residual_connection", "0,0,0,0"]),
    "mixednet_notebook": ("mixednet", 204, ["--residual_connection", "0,0,0,0", "--first_conv_kernel_size", "5", "--stride", "3",
                                            "--pointwise_filters", "64,64,64,64", "--mixconv_kernel_sizes", "[5], [7,11], [9,15], [23]"]),
    "inception_default": ("inception", 194, ["--dropout", "0.0"]),   # the Keras dropout generator cannot be reproduced outside TF
}
B = 16


def seeded_weights(shapes_names, rng):
    out = []
    for name, shape in shapes_names:
        n = name.lower()
        if "moving_variance" in n:
            w = rng.uniform(0.5, 1.5, size=shape)
        elif "moving_mean" in n or n.endswith("beta") or "bias" in n:
            w = rng.uniform(-0.1, 0.1, size=shape)
        elif n.endswith("gamma"):
            w = rng.uniform(0.8, 1.2, size=shape)
        else:
            fan = max(1, int(np.prod(shape[:-1])))
            w = rng.uniform(-1.0, 1.0, size=shape) * np.sqrt(3.0 / fan)
        out.append(np.asarray(w, np.float32))
    return out


def run_case(name, module_name, T, extra, tf):
    import microwakeword.inception as inception
    import microwakeword.mixednet as mixednet
    module = {"mixednet": mixednet, "inception": inception}[module_name]
    ap = argparse.ArgumentParser()
    module.model_parameters(ap)
    flags = ap.parse_args(extra)
    model = module.model(flags, shape=(T, 40), batch_size=B)
    names = [getattr(v, "path", getattr(v, "name", "v%d" % i)) for i, v in enumerate(model.weights)]
    shapes = [tuple(w.shape) for w in model.get_weights()]
    rng = np.random.default_rng(20260926)
    w0 = seeded_weights(list(zip(names, shapes)), rng)
    w0[-2] = w0[-2] * np.float32(6.0)      # dense kernel: some windows end up with |logit| > 17 (saturated sigmoid in float32)
    x = (rng.integers(0, 667, size=(B, T, 40)).astype(np.float32) * np.float32(0.0390625)).astype(np.float32)
    y = (rng.random(B) < 0.5).astype(np.float64)
    blob = {"names": np.array(names), "x": x, "y": y, "frames": np.int64(T), "flags": np.array(extra)}
    for i, w in enumerate(w0):
        blob["w0/%03d" % i] = w
    model.set_weights(w0)
    blob["p_eval"] = np.asarray(model(x, training=False)).reshape(-1)
    blob["logits_saturated"] = np.int64(np.sum((blob["p_eval"] <= 1e-7) | (blob["p_eval"] >= 1 - 1e-7)))

    cutoffs = np.linspace(0.0, 1.0, 101).tolist()
    for tag, penalty, cw in (("uniform", np.ones(B), {0: 1.0, 1: 1.0}),
                             ("weighted", rng.choice([0.5, 1.0, 2.0], size=B), {0: 20.0, 1: 1.0})):
        model.set_weights(w0)
        model.compile(optimizer=tf.keras.optimizers.Adam(), loss=tf.keras.losses.BinaryCrossentropy(from_logits=False),
                      metrics=[tf.keras.metrics.BinaryAccuracy(name="accuracy"), tf.keras.metrics.Recall(name="recall"),
                               tf.keras.metrics.Precision(name="precision"),
                               tf.keras.metrics.TruePositives(name="tp", thresholds=cutoffs),
                               tf.keras.metrics.FalsePositives(name="fp", thresholds=cutoffs),
                               tf.keras.metrics.TrueNegatives(name="tn", thresholds=cutoffs),
                               tf.keras.metrics.FalseNegatives(name="fn", thresholds=cutoffs),
                               tf.keras.metrics.AUC(name="auc"), tf.keras.metrics.BinaryCrossentropy(name="loss")])
        model.optimizer.learning_rate.assign(0.001)
        truth = y.reshape(-1, 1)
        combined = penalty * np.vectorize(cw.get)(truth)            # train.py:288-293: [B] * [B,1] -> [B,B]
        blob["%s/penalty" % tag] = np.asarray(penalty, np.float64)
        blob["%s/class_weights" % tag] = np.array([cw[0], cw[1]], np.float64)
        blob["%s/combined_shape" % tag] = np.array(combined.shape, np.int64)
        # gradient of the step's own loss call (keras.Model.compute_loss is what train_step evaluates)
        with tf.GradientTape() as tape:
            yp = model(x, training=True)
            loss = model.compute_loss(x=tf.constant(x), y=tf.constant(truth), y_pred=yp, sample_weight=tf.constant(combined))
        tv = model.trainable_variables
        grads = tape.gradient(loss, tv)
        blob["%s/tape_loss" % tag] = np.float64(loss.numpy())
        blob["%s/p_train" % tag] = np.asarray(yp).reshape(-1)
        blob["%s/trainable_names" % tag] = np.array([getattr(v, "path", v.name) for v in tv])
        for i, g in enumerate(grads):
            blob["%s/grad/%03d" % (tag, i)] = np.asarray(g, np.float32)
        model.set_weights(w0)   # (the training-mode forward above moved the BN moving statistics)
        model.reset_metrics()
        res = model.train_on_batch(x, truth, sample_weight=combined)
        flat = []
        for r in (res if isinstance(res, (list, tuple)) else [res]):
            flat.append(np.asarray(r, np.float64).reshape(-1))
        blob["%s/train_on_batch_result" % tag] = np.concatenate(flat)
        blob["%s/train_on_batch_lengths" % tag] = np.array([f.size for f in flat], np.int64)
        for i, w in enumerate(model.get_weights()):
            blob["%s/w1/%03d" % (tag, i)] = np.asarray(w, np.float32)
    return {"%s/%s" % (name, k): v for k, v in blob.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="tests/golden/tf_golden.npz")
    ap.add_argument("--cases", default=",".join(CASES))
    args = ap.parse_args()
    try:
        import tensorflow as tf
    except ImportError:
        sys.exit("TensorFlow is required (pip install 'tensorflow>=2.16'); run this where the reference itself runs")
    try:
        import keras
        kv = keras.__version__
    except ImportError:
        kv = "?"
    tf.keras.utils.set_random_seed(0)
    blob = {"tensorflow_version": np.array(tf.__version__), "keras_version": np.array(kv), "batch": np.int64(B),
            "cases": np.array(args.cases.split(","))}
    for name in args.cases.split(","):
        module_name, T, extra = CASES[name]
        blob.update(run_case(name, module_name, T, extra, tf))
        print("case", name, "done")
    np.savez_compressed(args.out, **blob)
    print("wrote %d arrays to %s (TensorFlow %s, Keras %s)" % (len(blob), args.out, tf.__version__, kv))


if __name__ == "__main__":
    main()
