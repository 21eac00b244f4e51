"""TEST INFRASTRUCTURE ONLY — imports the *reference's own* ``microwakeword/train.py`` unchanged, so that its ``train()`` /
``validate_nonstreaming()`` (reference ``microwakeword/train.py:41-163,166-462``) can drive this package's ``Model`` and
``FeatureHandler`` — the caller that "drops in for ``microwakeword.train``" names.

``train.py`` imports ``tensorflow`` for exactly these things (the arithmetic itself is behind ``model.*``):

  ``tf.keras.losses.BinaryCrossentropy(from_logits=False)``, ``tf.keras.optimizers.Adam()``, nine
  ``tf.keras.metrics.*(name=, thresholds=)`` constructors                       train.py:206-221  -> inert tokens handed to ``model.compile``
  ``tf_decorator.unwrap(model.train_function)``                                  train.py:226-227  -> ``(None, f)``
  ``tf.train.Checkpoint(optimizer=, model=)`` / ``.restore`` / ``.save`` /
  ``tf.train.latest_checkpoint``                                                 train.py:229-233,448-451,461  -> the package's own checkpoint twin
  ``tf.summary.create_file_writer`` / ``.as_default()`` / ``tf.summary.scalar`` / ``.flush()``   train.py:236-241,328-389  -> recorded in memory

TensorFlow is not installable in this image, so stand-ins for those names are registered in ``sys.modules`` —
exactly as ``oracle/ref_data_shim.py`` does for ``absl`` / ``mmap_ninja`` — and the reference file is executed from
``/root/reference`` as it lies there.  Nothing of Keras' arithmetic is emulated: loss, optimizer, metrics, evaluation are what
``microwakeword_amd.model.Model`` computes on its engine.  Every call the reference makes on the model and on the data
processor, and what came back, is recorded (``Trace``) so that a GPU test can replay the sequence where ``/root/reference`` does
not exist.

It cannot travel: nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may import this module.  Nothing is ever written
under ``/root/reference`` (bytecode writing is disabled before the import).
"""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MWW_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "microwakeword", "train.py"))


class _Token:
    """What a ``tf.keras`` constructor returns here: its name and arguments, nothing else."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return "<%s %s>" % (self.kind, {k: v for k, v in self.__dict__.items() if k not in ("kind", "thresholds")})


class Summaries:
    """``tf.summary`` stand-in: scalars are kept in ``records`` as (writer directory, name, step, value)."""

    records = []
    _current = [None]

    class Writer:
        def __init__(self, directory):
            self.directory = str(directory)

        @contextlib.contextmanager
        def as_default(self):
            prev = Summaries._current[0]
            Summaries._current[0] = self
            try:
                yield self
            finally:
                Summaries._current[0] = prev

        def flush(self):
            pass

    @staticmethod
    def create_file_writer(directory):
        return Summaries.Writer(directory)

    @staticmethod
    def scalar(name, value, step=None):
        w = Summaries._current[0]
        Summaries.records.append((w.directory if w else None, name, int(step), float(value)))


class Checkpoint:
    """``tf.train.Checkpoint(optimizer=, model=)`` backed by the package's checkpoint twin (``<prefix>.weights.npz`` +
    ``<prefix>.opt.npz``: weights, BN moving statistics, Adam slots and step — what the TF object would hold for this pair)."""

    def __init__(self, optimizer=None, model=None):
        self.model = model

    def restore(self, save_path):
        if save_path is None:        # tf.train.latest_checkpoint found nothing: TF's restore(None) is a no-op too
            return self
        self.model.load_weights(save_path + ".weights")
        self.model.load_optimizer_state(save_path + ".opt.npz")
        return self

    def save(self, file_prefix):
        os.makedirs(os.path.dirname(file_prefix), exist_ok=True)
        self.model.save_weights(file_prefix + ".weights")
        self.model.save_optimizer_state(file_prefix + ".opt.npz")
        return file_prefix


def latest_checkpoint(directory):
    prefix = os.path.join(directory, "ckpt")
    return prefix if os.path.isfile(prefix + ".weights.npz") and os.path.isfile(prefix + ".opt.npz") else None


def _tensorflow_stub():
    tf = types.ModuleType("tensorflow")
    keras = types.ModuleType("tensorflow.keras")
    losses = types.ModuleType("tensorflow.keras.losses")
    optimizers = types.ModuleType("tensorflow.keras.optimizers")
    metrics = types.ModuleType("tensorflow.keras.metrics")
    losses.BinaryCrossentropy = lambda from_logits=False, **kw: _Token("BinaryCrossentropy", from_logits=from_logits)
    optimizers.Adam = lambda learning_rate=0.001, **kw: _Token("Adam", learning_rate=learning_rate)
    for name in ("BinaryAccuracy", "Recall", "Precision", "TruePositives", "FalsePositives", "TrueNegatives", "FalseNegatives",
                 "AUC", "BinaryCrossentropy"):
        setattr(metrics, name, (lambda kind: (lambda name=None, thresholds=None, **kw: _Token(kind, name=name, thresholds=thresholds)))(name))
    keras.losses, keras.optimizers, keras.metrics = losses, optimizers, metrics
    train = types.ModuleType("tensorflow.train")
    train.Checkpoint, train.latest_checkpoint = Checkpoint, latest_checkpoint
    summary = types.ModuleType("tensorflow.summary")
    summary.create_file_writer, summary.scalar = Summaries.create_file_writer, Summaries.scalar
    tf.keras, tf.train, tf.summary = keras, train, summary
    python = types.ModuleType("tensorflow.python")
    util = types.ModuleType("tensorflow.python.util")
    tf_decorator = types.ModuleType("tensorflow.python.util.tf_decorator")
    tf_decorator.unwrap = lambda f: (None, f)      # train.py:227 keeps element 1: the undecorated train function
    util.tf_decorator = tf_decorator
    python.util = util
    tf.python = python
    return {"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.keras.losses": losses, "tensorflow.keras.optimizers": optimizers,
            "tensorflow.keras.metrics": metrics, "tensorflow.train": train, "tensorflow.summary": summary, "tensorflow.python": python,
            "tensorflow.python.util": util, "tensorflow.python.util.tf_decorator": tf_decorator}


class LogCapture:
    lines = []

    @staticmethod
    def info(fmt, *args):
        LogCapture.lines.append(fmt % args if args else fmt)

    warning = info


def load_reference_train_module():
    """Returns the reference ``microwakeword.train`` module object, executed from its source file with the stand-ins above.
    The stand-ins are registered only while the file is imported and removed afterwards (a later ``import tensorflow`` elsewhere
    in the process must keep failing loudly); the module keeps its own references."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if "mww_reference_train" in sys.modules:
        return sys.modules["mww_reference_train"]
    stubs = _tensorflow_stub()
    absl_logging = types.ModuleType("absl.logging")
    absl_logging.info, absl_logging.warning = LogCapture.info, LogCapture.warning
    absl = types.ModuleType("absl")
    absl.logging = absl_logging
    stubs.update({"absl": absl, "absl.logging": absl_logging})
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("mww_reference_train", os.path.join(REFERENCE_ROOT, "microwakeword", "train.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["mww_reference_train"] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def run_reference_cli(argv):
    """Executes the reference's ``microwakeword/model_train_eval.py`` AS ``__main__`` (its CLI: argparse, ``load_config``,
    ``input_data.FeatureHandler(config)``, ``model_module.model(flags, shape, batch_size)``, ``train_model`` ->
    ``train.train`` - reference ``model_train_eval.py:45-128,277-439``) with THIS package's modules bound to the names it imports:

      microwakeword.data / .mixednet / .inception   -> microwakeword_amd.data / .mixednet / .inception  (the drop-in surface)
      microwakeword.train                           -> the reference's own train.py (load_reference_train_module)
      microwakeword.layers.modes                    -> the reference's own layers/modes.py (pure Python)
      microwakeword.utils                           -> ``save_model_summary`` only (reference utils.py:131-145 restated: model.summary(print_fn=) into
                                                       <path>/model_summary.txt; the rest of utils.py is TFLite conversion)
      microwakeword.test                            -> empty (TFLite evaluation, not reached with the --test_* flags at 0)
      tensorflow / absl                             -> the stand-ins above (+ absl.logging.set_verbosity and its level names)

    Returns the module globals of the run."""
    import runpy
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    train_mod = load_reference_train_module()
    from microwakeword_amd import data as amd_data, inception as amd_inception, mixednet as amd_mixednet

    spec = importlib.util.spec_from_file_location("mww_reference_modes", os.path.join(REFERENCE_ROOT, "microwakeword", "layers", "modes.py"))
    modes = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(modes)

    utils = types.ModuleType("microwakeword.utils")

    def save_model_summary(model, path, file_name="model_summary.txt"):   # utils.py:131-145
        lines = []
        model.summary(print_fn=lambda x: lines.append(x))
        with open(os.path.join(path, file_name), "w") as fd:
            fd.write("\n".join(lines))

    utils.save_model_summary = save_model_summary
    stubs = _tensorflow_stub()
    absl_logging = types.ModuleType("absl.logging")
    absl_logging.info, absl_logging.warning = LogCapture.info, LogCapture.warning
    absl_logging.DEBUG, absl_logging.INFO, absl_logging.WARN, absl_logging.ERROR, absl_logging.FATAL = 1, 0, -1, -2, -3
    absl_logging.set_verbosity = lambda v: None
    absl = types.ModuleType("absl")
    absl.logging = absl_logging
    pkg = types.ModuleType("microwakeword")
    pkg.__path__ = []
    layers = types.ModuleType("microwakeword.layers")
    layers.__path__ = []
    layers.modes = modes
    pkg.data, pkg.train, pkg.test, pkg.utils, pkg.inception, pkg.mixednet, pkg.layers = (
        amd_data, train_mod, types.ModuleType("microwakeword.test"), utils, amd_inception, amd_mixednet, layers)
    stubs.update({"absl": absl, "absl.logging": absl_logging, "microwakeword": pkg, "microwakeword.data": amd_data,
                  "microwakeword.train": train_mod, "microwakeword.test": pkg.test, "microwakeword.utils": utils,
                  "microwakeword.inception": amd_inception, "microwakeword.mixednet": amd_mixednet, "microwakeword.layers": layers,
                  "microwakeword.layers.modes": modes})
    saved = {k: sys.modules.get(k) for k in stubs}
    saved_argv = sys.argv
    sys.modules.update(stubs)
    sys.argv = ["model_train_eval.py"] + list(argv)
    try:
        return runpy.run_path(os.path.join(REFERENCE_ROOT, "microwakeword", "model_train_eval.py"), run_name="__main__")
    finally:
        sys.argv = saved_argv
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ------------------------------------------------------------------------------------------------- call / return trace
class Trace:
    """Wraps a model and a data processor: every call the reference's loop makes on them is forwarded and recorded as
    ``(object, method, summary of the arguments, summary of the result)``.  ``steps`` keeps, per ``train_on_batch`` call, the
    learning rate in force and the five result entries the reference reads (indices 1, 2, 3, 8, 9: train.py:305-308,329-333)."""

    def __init__(self, model, data_processor, keep_batches=False):
        self.calls = []             # (object tag, method, positional arguments, keyword arguments, summary of the result)
        self.paths = []             # the path argument of every save_* / load_* call, in order
        self.steps = []
        self.batches = []           # (x, y, combined sample weights) of every train_on_batch when keep_batches
        self.evals = []             # return_dict results of model.evaluate, counters as arrays
        self.model = _Recorder(model, "model", self)
        self.data = _Recorder(data_processor, "data", self)
        self.keep_batches = keep_batches


def _enc(v):
    """simple values (and dicts of them) as they are - they go into the JSON fixture -, anything else as {"__placeholder__": summary}"""
    if isinstance(v, (bool, int, float, str, type(None))):
        return v
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, dict) and all(isinstance(k, str) and isinstance(_enc(x), (bool, int, float, str, type(None))) for k, x in v.items()):
        return {k: _enc(x) for k, x in v.items()}
    return {"__placeholder__": _brief(v)}


def _brief(v):
    if isinstance(v, np.ndarray):
        return "ndarray%s:%s" % (list(v.shape), v.dtype)
    if isinstance(v, (list, tuple)):
        return type(v).__name__ + "[%d]" % len(v)
    if isinstance(v, dict):
        return "dict{%s}" % ",".join(sorted(map(str, v)))
    if isinstance(v, (int, float, str, bool, type(None))):
        return repr(v)
    if callable(v):
        return "function:" + getattr(v, "__name__", "?")
    return type(v).__name__


class _Recorder:
    """Attribute access is forwarded; callables are wrapped.  ``reset_metrics`` stays swappable as an attribute
    (train.py:89 ``swap_attribute(model, "reset_metrics", lambda: None)``): a set attribute lands on the wrapped object, which is
    where ``Model.evaluate`` looks it up."""

    def __init__(self, obj, tag, trace):
        object.__setattr__(self, "_obj", obj)
        object.__setattr__(self, "_tag", tag)
        object.__setattr__(self, "_trace", trace)

    def __setattr__(self, name, value):
        self._trace.calls.append((self._tag, "setattr:" + name, [_enc(value)], {}, None))
        setattr(self._obj, name, value)

    def __getattr__(self, name):
        obj, tag, trace = self._obj, self._tag, self._trace
        val = getattr(obj, name)
        if name == "optimizer":
            return _OptimizerRecorder(val, trace)
        if not callable(val):
            return val

        def call(*a, **k):
            out = val(*a, **k)
            trace.calls.append((tag, name, [_enc(x) for x in a], {kk: _enc(vv) for kk, vv in sorted(k.items())}, _brief(out)))
            if name in ("save_weights", "load_weights", "save_optimizer_state", "load_optimizer_state"):
                trace.paths.append(str(a[0]))
            if tag == "model" and name in ("train_on_batch", "train_on_device_batch") and out is not None:
                trace.steps.append(dict(lr=float(obj.optimizer.learning_rate.value), result=[float(out[i]) for i in (1, 2, 3, 8, 9)]))
                if trace.keep_batches and name == "train_on_batch":
                    trace.batches.append((np.array(a[0], np.float32), np.array(a[1], np.float32), np.array(k.get("sample_weight"), np.float64)))
            if tag == "model" and name == "evaluate":
                trace.evals.append({kk: (np.array(vv.numpy()) if hasattr(vv, "numpy") else float(vv)) for kk, vv in out.items()})
            return out

        return call


class _OptimizerRecorder:
    def __init__(self, opt, trace):
        self._opt, self._trace = opt, trace

    @property
    def learning_rate(self):
        opt, trace = self._opt, self._trace

        class LR:
            def assign(self, v):
                trace.calls.append(("model", "optimizer.learning_rate.assign", [float(v)], {}, None))
                opt.learning_rate.assign(v)

            def numpy(self):
                return opt.learning_rate.numpy()

        return LR()
