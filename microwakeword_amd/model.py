"""The subset of ``tf.keras.Model`` that the reference train loop touches, backed by the MI355X
engine (C ABI in ``include/mww.h``).  Call sites in the reference:

  compile / make_train_function / train_function      microwakeword/train.py:223-227
  optimizer.learning_rate.assign(lr)                   microwakeword/train.py:265
  train_on_batch(x, y, sample_weight=) -> list         microwakeword/train.py:295-299 (indices 1,2,3,8,9 used)
  evaluate(x, y, batch_size=1024, return_dict=True)    microwakeword/train.py:50-58,89-96
  reset_metrics (swappable attribute)                  microwakeword/train.py:48,89,343
  save_weights / load_weights                          microwakeword/train.py:336-338,393-399,448-450,462;
                                                       microwakeword/model_train_eval.py:424-426
  summary(print_fn=)                                   microwakeword/utils.py:131-145
  get_weights / set_weights                            Keras order, SURVEY §A.4

Weights files: the reference writes Keras ``.weights.h5`` (needs h5py/Keras, absent here); this
class writes the same arrays in the same order as ``<path>`` + ``.npz`` twin (documented in
INTEGRATION.md) and reads either the twin or — if h5py is importable — a real ``.weights.h5``.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import numpy as np

from . import native
from .layout import FEATURE_BINS, MixedNetLayout


def glorot_uniform(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def initial_weights(layout: MixedNetLayout, seed: Optional[int] = None) -> List[np.ndarray]:
    """Keras default initialisers (glorot_uniform kernels with Keras' fan computation, zero biases,
    BN gamma=1 beta=0 mean=0 var=1) — SURVEY §A.1."""
    rng = np.random.default_rng(seed)
    out = []
    for name, shape, _ in layout.keras_vars:
        if name.endswith(".kernel"):
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                out.append(glorot_uniform(rng, shape, shape[2] * rf, shape[3] * rf))
            else:
                out.append(glorot_uniform(rng, shape, shape[0], shape[1]))
        elif name.endswith(("gamma", "moving_variance")):
            out.append(np.ones(shape, np.float32))
        else:
            out.append(np.zeros(shape, np.float32))
    return out


class _Assignable:
    def __init__(self, value):
        self.value = float(value)

    def assign(self, v):
        self.value = float(v)

    def numpy(self):
        return np.float32(self.value)


class _Optimizer:
    def __init__(self, lr=1e-3):
        self.learning_rate = _Assignable(lr)


class _Counts:
    """tp/fp/tn/fn results expose ``.numpy()`` in the reference (train.py:72,103-105)."""

    def __init__(self, a):
        self._a = np.asarray(a, np.float32)

    def numpy(self):
        return self._a


WEIGHT_BROADCAST_MODES = ("per_sample", "keras_last_axis", "keras_first_axis")
# How the reference's [B,B] sample weight (train.py:288-293: penalty[B] * class_weight(y)[B,1]) is reduced to per-sample
# weights - ONE default for both entry points (round 6): the package's own loop (train.train, config key
# ``sample_weight_broadcast``) and a [B,B] matrix handed to the drop-in ``train_on_batch`` by the reference's loop.  The default
# is the reference's ARITHMETIC as we read Keras 3 (keras/src/losses/loss.py: ``Loss.__call__`` -> ``reduce_weighted_values``:
# ``squeeze_or_expand_to_same_rank`` leaves a [B] loss vector and a [B,B] weight as they are, ``values * sample_weight``
# broadcasts the losses along the LAST axis, ``reduce_values`` divides the sum by ``prod(shape(values))`` = B*B):
# w_j = penalty_j * mean_i cw(y_i) - a user who switches between ``microwakeword_amd.train.train`` and the reference's loop
# gets the same loss curve.  ``per_sample`` (the evident intent, w_i = penalty_i * cw(y_i)) is the opt-in.  All readings are
# equal while class weights (or penalty weights) are uniform, which holds for every BASELINE configuration; which one
# TensorFlow really evaluates is settled by tests/test_tf_golden.py as soon as tests/golden/tf_golden.npz exists
# (tools/make_tf_golden.py) - that test fails until these constants name the reading the reference's numbers follow.
DEFAULT_WEIGHT_BROADCAST = "keras_last_axis"
MATRIX_WEIGHT_BROADCAST = DEFAULT_WEIGHT_BROADCAST


def readings_differ(penalty, labels, negative_class_weight, positive_class_weight):
    """Do the readings of the [B,B] weight matrix give different per-sample weights on THIS batch?  (Only if the class weights
    and the penalty weights both vary over it.)"""
    cw = np.where(np.asarray(labels).reshape(-1) > 0.5, positive_class_weight, negative_class_weight)
    pen = np.asarray(penalty).reshape(-1)
    return bool(cw.size and cw.min() != cw.max() and pen.min() != pen.max())


def combine_weights(penalty, labels, negative_class_weight, positive_class_weight, mode=DEFAULT_WEIGHT_BROADCAST):
    """The per-sample loss weights of a training batch.  train.py:288-293 multiplies penalty[B] by class_weight(y)[B,1],
    which broadcasts to a [B,B] matrix W[i,j] = penalty_j * cw(y_i) that Keras then reduces against the [B] per-sample
    losses.  ``per_sample`` is what the code evidently means: w_i = penalty_i * cw(y_i).  The two readings of
    what Keras 3 actually evaluates:
    ``keras_last_axis`` (the default of both entry points, DEFAULT_WEIGHT_BROADCAST) - the losses broadcast along the last axis of W and everything is divided by B*B:
    w_j = penalty_j * mean_i cw(y_i) (what keras/src/losses/loss.py reduce_weighted_values does as we read it: a [B] loss
    vector against a [B,B] weight is left as is by squeeze_or_expand_to_same_rank, multiplied with ordinary broadcasting
    and divided by the element count B*B - not verifiable here, TensorFlow is not installable); ``keras_first_axis`` -
    w_i = cw(y_i) * mean_j penalty_j.  per_sample equals keras_last_axis when the class weights are uniform over the batch
    and keras_first_axis when the penalty weights are; with both non-uniform (e.g. negative_class_weight 20 and mixed
    penalty_weight providers) the loss curves differ: set ``sample_weight_broadcast: per_sample`` in the training config to
    follow the reference's intent rather than its arithmetic."""
    penalty = np.asarray(penalty, np.float64).reshape(-1)
    cw = np.where(np.asarray(labels).reshape(-1) > 0.5, positive_class_weight, negative_class_weight).astype(np.float64)
    if mode == "per_sample":
        w = penalty * cw
    elif mode == "keras_last_axis":
        w = penalty * cw.mean()
    elif mode == "keras_first_axis":
        w = cw * penalty.mean()
    else:
        raise ValueError("sample_weight_broadcast must be one of %s" % (WEIGHT_BROADCAST_MODES,))
    return w.astype(np.float32)


class Model:
    def __init__(self, flags, shape, batch_size, device=0, stream=None, lib=None, seed=None, max_batch=None, layout=None,
                 name="mixednet"):
        if tuple(shape)[1] != FEATURE_BINS:
            raise ValueError("input shape must be (spectrogram_length, 40)")
        self.flags = flags
        self.name = name
        self.layout = layout if layout is not None else MixedNetLayout(flags, int(shape[0]))
        self.input_shape = (batch_size, int(shape[0]), FEATURE_BINS)
        self.batch_size = batch_size
        mb = int(max_batch or max(int(batch_size or 1), 1024))
        self.engine = native.Engine(lib=lib, device=device, stream=stream, **self.layout.engine_args(mb))
        self.engine.set_grad_mask(self.layout.grad_mask())
        self.set_weights(initial_weights(self.layout, seed))
        self.optimizer = _Optimizer()
        self.loss = None
        self.train_function = None
        self._compiled = False
        self.data_parallel = None   # parallel.DataParallel once join_data_parallel() was called
        self.sample_weight_broadcast = MATRIX_WEIGHT_BROADCAST   # how a [B,B] sample_weight matrix is reduced (combine_weights)

    # ---- Keras surface
    def compile(self, optimizer=None, loss=None, metrics=None):
        """The loss (BinaryCrossentropy, from_logits=False), optimizer (Adam defaults) and the nine
        metrics of train.py:206-221 are built into the engine; the arguments are accepted for
        signature compatibility and checked for the two things that would change the arithmetic."""
        if optimizer is not None and hasattr(optimizer, "learning_rate"):
            lr = optimizer.learning_rate
            self.optimizer.learning_rate.assign(float(lr.numpy()) if hasattr(lr, "numpy") else float(getattr(lr, "value", lr)))
        if getattr(loss, "from_logits", False):
            raise ValueError("the engine implements BinaryCrossentropy(from_logits=False) as the reference uses")
        self._compiled = True

    def make_train_function(self):
        self.train_function = self._train_function
        return self.train_function

    def _train_function(self, *a, **k):
        raise RuntimeError("train_function is internal to the engine; call train_on_batch")

    def count_params(self):
        return self.layout.keras_param_counts()[0]

    def get_weights(self):
        return self.layout.unpack(self.engine.get_params(), self.engine.get_bn_state())

    def set_weights(self, weights: Sequence[np.ndarray]):
        p, s = self.layout.pack(weights)
        self.engine.set_params(p)
        self.engine.set_bn_state(s)

    def reset_metrics(self):
        self.engine.metrics_reset()
        self._loss_sum, self._loss_n = 0.0, 0

    def join_data_parallel(self, group=None, sync_bn=False, grad_buckets=1, library_comm=True):
        """One process per GPU (``torch.distributed`` initialised by the caller): from here on ``train_on_batch`` /
        ``train_on_device_batch`` are data-parallel steps - local forward / backward on this rank's batch, one all-reduce
        of the flat gradient inside the engine's step, Adam on the average (parallel.DataParallel) - and
        ``evaluation_results`` reports the counters summed over the ranks (validation is sharded by
        ``FeatureHandler.evaluate_on_device``).  Collective: every rank calls it."""
        from .parallel import DataParallel
        self.data_parallel = DataParallel.for_engine(self.engine, group=group, sync_bn=sync_bn, grad_buckets=grad_buckets,
                                                     library_comm=library_comm)
        dp = self.data_parallel
        if dp.world > 1 and getattr(self.layout, "dropout", 0.0) > 0:
            # every engine starts from the same dropout seed and counter: without this the W ranks would apply the SAME
            # [B/W, features] mask at every step instead of B independent rows (Inception: dropout 0.2)
            self.engine.set_option("dropout_seed", 0x5EED * dp.world + dp.rank + 1)
        return dp

    def _metric_results(self, reduce=False):
        """``reduce``: the counters of every rank's context summed by ONE all-reduce of the raw state (collective; the
        device counters themselves stay rank-local, so accumulation across calls keeps working)."""
        raw = self.engine.metrics_raw()
        dp = self.data_parallel
        if reduce and dp is not None and dp.world > 1:
            raw = native.metrics_from_vector(dp.allreduce_host(native.metrics_to_vector(raw)))
        return native.metrics_from_raw(raw)

    def _per_sample_weights(self, sample_weight, n):
        if sample_weight is None:
            return np.ones(n, np.float32)
        sw = np.asarray(sample_weight, np.float64)
        if sw.ndim == 2 and sw.shape == (n, n) and n > 1:
            # train.py:291-293 broadcasts penalty[B] * class_weight(y)[B,1] to [B,B] with W[i,j] = penalty_j * cw(y_i).
            # Default (MATRIX_WEIGHT_BROADCAST): what Keras 3 evaluates for the matrix by our reading, the column means;
            # "per_sample" is the diagonal (the evident intent), "keras_first_axis" the row means.  All agree when either
            # factor is uniform.
            mode = self.sample_weight_broadcast
            if mode == "per_sample":
                sw = np.diagonal(sw)
            elif mode == "keras_last_axis":
                sw = sw.mean(axis=0)
            elif mode == "keras_first_axis":
                sw = sw.mean(axis=1)
            else:
                raise ValueError("sample_weight_broadcast must be one of %s" % (WEIGHT_BROADCAST_MODES,))
        sw = sw.reshape(-1)
        if sw.size != n:
            raise ValueError("sample_weight does not match the batch")
        return sw.astype(np.float32)

    def train_on_batch(self, x, y, sample_weight=None, return_dict=False):
        """Returns ``[loss, accuracy, recall, precision, tp[101], fp[101], tn[101], fn[101], auc, loss_metric]``
        (the compiled metric order of train.py:209-221; cumulative since the last reset_metrics)."""
        x = np.asarray(x, np.float32)
        n = x.shape[0]
        self.engine.set_batch(x)
        return self._step_on_device(n, y, sample_weight, return_dict)

    def train_on_device_batch(self, n, y=None, sample_weight=None, return_dict=False, want_results=True):
        """Same as train_on_batch for a batch that ``FeatureHandler.next_training_batch_on_device``
        already left in HBM together with its labels and weights (no host round trip of x).  Passing
        ``y`` replaces the targets that travelled with the batch.  ``want_results=False`` only enqueues the step and returns
        None: reading the loss and the metric counters back synchronises the host with the device, and a loop that does it
        every step runs at 0.50 ms per step where the step itself takes 0.31 (the counters keep accumulating on the device:
        the next call that asks gets the cumulative values; only the Keras loss tracker - entry 0 of the list, which
        train.py never reads - then covers the steps that were read)."""
        return self._step_on_device(n, y, sample_weight, return_dict, want_results)

    def _step_on_device(self, n, y, sample_weight, return_dict, want_results=True):
        if y is not None:
            y = np.asarray(y, np.float32).reshape(-1)
            self.engine.set_targets(y, self._per_sample_weights(sample_weight, n))
        self.engine.train_step(n, self.optimizer.learning_rate.value)
        if not want_results:
            return None
        _, _, loss = self.engine.read_outputs(n)
        m = self._metric_results()
        # Keras' first entry is the loss tracker: a running mean since the last reset_metrics()
        self._loss_sum = getattr(self, "_loss_sum", 0.0) + loss * n
        self._loss_n = getattr(self, "_loss_n", 0) + n
        self.last_batch_loss = loss
        run = self._loss_sum / self._loss_n
        if return_dict:
            return dict(m, loss=run)
        return [run, m["accuracy"], m["recall"], m["precision"], m["tp"], m["fp"], m["tn"], m["fn"], m["auc"], m["loss"]]

    def predict_on_batch(self, x):
        x = np.asarray(x, np.float32)
        out = []
        for s in range(0, x.shape[0], self.engine.max_batch):
            e = min(x.shape[0], s + self.engine.max_batch)
            self.engine.set_batch(x[s:e])
            self.engine.forward(e - s, training=False)
            out.append(self.engine.read_outputs(e - s, want_loss=False)[0])
        return np.concatenate(out).reshape(-1, 1)

    __call__ = predict_on_batch

    def evaluate(self, x, y, batch_size=1024, return_dict=True, verbose=0):
        """Forward-only pass in inference mode (BN moving statistics) that accumulates the compiled
        metrics; like Keras it resets them first (the reference swaps ``reset_metrics`` for a no-op
        to keep accumulating across two calls, train.py:88-96 — honoured because the attribute is
        looked up at call time)."""
        self.reset_metrics()
        x = np.asarray(x, np.float32)
        y = np.asarray(y, np.float32).reshape(-1)
        bs = min(int(batch_size), self.engine.max_batch)
        for s in range(0, x.shape[0], bs):
            e = min(x.shape[0], s + bs)
            self.engine.set_batch(x[s:e])
            self.engine.set_targets(y[s:e], np.ones(e - s, np.float32))
            self.engine.forward(e - s, training=False, update_metrics=True)
        res = self.evaluation_results(reduce=False)   # host arrays are not sharded: every rank scored all of x
        if return_dict:
            return res
        return [res["loss"], res["accuracy"], res["recall"], res["precision"], res["tp"], res["fp"], res["tn"], res["fn"],
                res["auc"], res["loss"]]

    def evaluation_results(self, reduce=True):
        """The ``return_dict=True`` result of ``evaluate`` from the counters accumulated so far (data-parallel: over all
        ranks - a collective call)."""
        m = self._metric_results(reduce=reduce)
        return dict(accuracy=m["accuracy"], recall=m["recall"], precision=m["precision"], auc=m["auc"], loss=m["loss"],
                    tp=_Counts(m["tp"]), fp=_Counts(m["fp"]), tn=_Counts(m["tn"]), fn=_Counts(m["fn"]))

    # ---- persistence
    def save_weights(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        ws = self.get_weights()
        np.savez(path + ".npz", **{"%03d:%s" % (i, n): w for i, ((n, _, _), w) in enumerate(zip(self.layout.keras_vars, ws))})

    def load_weights(self, path):
        twin = path + ".npz" if not path.endswith(".npz") else path
        if os.path.isfile(twin):
            z = np.load(twin)
            keys = sorted(z.files)
            self.set_weights([z[k] for k in keys])
            return
        if os.path.isfile(path):
            raise RuntimeError("%s is a Keras .weights.h5 checkpoint (train.py:336-338,448-451).  This package keeps the same variables in "
                               "the same order as an .npz twin; convert on the machine that has TensorFlow with "
                               "tools/keras_weights_to_npz.py (and back, for the reference's TFLite export, with tools/npz_to_keras_weights.py)" % path)
        raise FileNotFoundError(path)

    def save_optimizer_state(self, path):
        m, v, step = self.engine.get_opt_state()
        np.savez(path, m=m, v=v, step=np.int64(step), lr=np.float64(self.optimizer.learning_rate.value))

    def load_optimizer_state(self, path):
        z = np.load(path)
        self.engine.set_opt_state(z["m"], z["v"], int(z["step"]))

    def summary(self, print_fn=print):
        total, trainable = self.layout.keras_param_counts()
        print_fn("Model: %s on MI355X engine (%s)" % (self.name, self.engine.nl.version()))
        if getattr(self, "kernel_family", None):
            print_fn("Kernels: %s" % self.kernel_family)
        for line in self.layout.summary_lines():
            print_fn(line)
        print_fn("Total params: %d" % total)
        print_fn("Trainable params: %d" % trainable)
        print_fn("Non-trainable params: %d" % (total - trainable))
