"""TEST INFRASTRUCTURE ONLY — executes the *reference's own* model builders, unchanged and from where they lie
(``/root/reference/microwakeword/mixednet.py:278-386``, ``inception.py:233-338`` and the layer files they import:
``layers/stream.py``, ``strided_drop.py``, ``sub_spectral_normalization.py``, ``delay.py``, ``average_pooling2d.py``,
``modes.py``), so that the graph the oracle restates (``oracle/model_oracle.py``: MixConv split and right alignment, residual
placement, attention, pooling, Stream padding, StridedDrop, SubSpectralNormalization's reshape, Flatten order, the order in
which variables are created) is checked against the code that defines it instead of against a reading of it.

What is NOT the reference here: TensorFlow / Keras (``tensorflow>=2.16`` ⇒ Keras 3, reference ``setup.py:18``) is not
installable in this image, so a stand-in ``tensorflow`` module is registered in ``sys.modules`` (as ``ref_data_shim`` does for
``absl`` / ``mmap_ninja`` and ``ref_train_shim`` for the training loop) whose *layer primitives* restate the published Keras
semantics, eagerly, in torch float64:

  Conv2D / DepthwiseConv2D   NHWC cross-correlation, kernel [kh, kw, Cin, Cout] / [kh, kw, C, 1], ``valid`` or ``same`` (TensorFlow's
                             split: the odd cell goes right / bottom), strides, dilation, optional bias and activation
  BatchNormalization         axis -1, momentum 0.99, epsilon 1e-3; training: batch mean / biased variance, moving statistics
                             m <- 0.99 m + 0.01 batch; variables gamma, beta, moving_mean, moving_variance in that order
  Dense, Flatten (row-major over [T, W, C]), Activation, Dropout (inverted: keep-mask / (1 - rate), mask supplied by the caller),
  Average / MaxPooling2D (``valid``), Concatenate, Reshape, Identity; ``tf.split`` / ``tf.pad`` / ``tf.concat``,
  ``keras.ops.expand_dims`` / ``transpose``; ``Layer.__call__`` = build once with the input shape, then ``call``.

So the model half stays **unpinned against TensorFlow** for those primitives (a convolution is a convolution; BN's formula and
constants are the documented ones) — what this shim pins is everything the reference's files themselves decide.  Variables are
handed to the layers in creation order from a list the caller supplies (the oracle's ``Var`` list): a shape mismatch at any
position is an error, which checks the oracle's variable order and shapes too.

It cannot travel: ``tests/golden/make_golden_ref_graph.py`` freezes what it returns into ``tests/golden/ref_graph_golden.npz``
for the GPU box.  Nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may import this module; nothing is written under
``/root/reference`` (bytecode writing is disabled before the import).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("MWW_REFERENCE_ROOT", "/root/reference")
BN_MOMENTUM, BN_EPS = 0.99, 1e-3       # Keras BatchNormalization defaults (the reference passes none)


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "microwakeword", "mixednet.py"))


# ------------------------------------------------------------------------------------------------- tensors

class Shape(tuple):
    """What ``tensor.shape`` has to offer the reference: indexing, ``rank``, ``as_list()``."""
    @property
    def rank(self):
        return len(self)

    def as_list(self):
        return list(self)

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return Shape(r) if isinstance(i, slice) else r


class KT:
    """A Keras tensor that already has its value: a torch float64 tensor, NHWC."""
    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return Shape(self.t.shape)

    def __getitem__(self, idx):
        return KT(self.t[idx])

    def _v(self, o):
        return o.t if isinstance(o, KT) else o

    def __add__(self, o):
        return KT(self.t + self._v(o))

    __radd__ = __add__

    def __mul__(self, o):
        return KT(self.t * self._v(o))

    __rmul__ = __mul__


class Variable:
    def __init__(self, name, value, trainable):
        self.name, self.trainable = name, trainable
        self.value = value                      # torch float64; a leaf that requires grad when trainable
        self.updated = None                     # BN moving statistics after a training-mode call

    @property
    def shape(self):
        return Shape(self.value.shape)


class Run:
    """The state of one execution of a builder: the batch ``Input`` returns, the values the variables take (creation order), the
    mode, the dropout keep-mask; afterwards the variables created and the classifier's pre-activation."""
    current = None

    def __init__(self, x, values=None, training=True, dropout_mask=None):
        self.x = torch.as_tensor(np.asarray(x), dtype=torch.float64)
        self.values = None if values is None else [np.asarray(v) for v in values]
        self.training, self.dropout_mask = training, dropout_mask
        self.variables = []
        self.layers = []
        self.logits = None

    def new_variable(self, layer, name, shape, trainable, default):
        k = len(self.variables)
        shape = tuple(int(s) for s in shape)
        if self.values is None:
            val = torch.full(shape, float(default), dtype=torch.float64)
        else:
            if k >= len(self.values):
                raise ValueError("the reference creates more variables than supplied: #%d %s%s of %s" % (k, name, shape, layer.name))
            if tuple(self.values[k].shape) != shape:
                raise ValueError("variable #%d: the reference creates %s%s (%s), supplied %s" % (k, name, shape, layer.name, self.values[k].shape))
            val = torch.tensor(self.values[k], dtype=torch.float64)
        if trainable:
            val.requires_grad_(True)
        v = Variable("%s/%s" % (layer.name, name), val, trainable)
        (layer._trainable_variables if trainable else layer._non_trainable_variables).append(v)
        self.variables.append(v)
        return v


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(i) for i in v)


def _same_pads(n, k, s, d):
    """TensorFlow's SAME: output ceil(n / s); the odd padding cell goes to the end."""
    eff = (k - 1) * d + 1
    total = max((-(-n // s) - 1) * s + eff - n, 0)
    return total // 2, total - total // 2


def _activation(name, t):
    if name in (None, "linear"):
        return t
    if name == "relu":
        return torch.relu(t)
    if name == "sigmoid":
        return torch.sigmoid(t)
    raise ValueError("activation %r is not restated" % (name,))


# ------------------------------------------------------------------------------------------------- keras.layers

class Layer:
    _count = {}

    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        if kwargs:
            raise TypeError("unexpected layer arguments %s" % sorted(kwargs))
        base = type(self).__name__.lower()
        Layer._count[base] = Layer._count.get(base, 0) + 1
        self.name = name or "%s_%d" % (base, Layer._count[base])
        self.trainable = trainable
        self.built = False
        self._trainable_variables, self._non_trainable_variables = [], []      # what the layer owns itself (Keras 3's names)
        if Run.current is not None:
            Run.current.layers.append(self)

    def build(self, input_shape):
        self.built = True

    def add_weight(self, name=None, shape=None, initializer=None, trainable=True, dtype=None, **kw):
        return Run.current.new_variable(self, name, shape, trainable, 0.0)

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            self.build([i.shape for i in inputs] if isinstance(inputs, (list, tuple)) else inputs.shape)
            self.built = True
        return self.call(inputs, *args, **kwargs)

    def call(self, inputs):
        return inputs

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable}


class Wrapper(Layer):
    def __init__(self, layer, **kwargs):
        super().__init__(**kwargs)
        self.layer = layer


class Identity(Layer):
    pass


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation = activation

    def call(self, inputs):
        return KT(_activation(self.activation, inputs.t))


class _ConvBase(Layer):
    depthwise = False

    def __init__(self, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1), use_bias=True, activation=None, **kwargs):
        super().__init__(**kwargs)
        self.kernel_size, self.strides, self.dilation_rate = _pair(kernel_size), _pair(strides), _pair(dilation_rate)
        self.padding, self.use_bias, self.activation = padding, use_bias, activation
        if padding not in ("valid", "same"):
            raise ValueError("padding %r" % (padding,))

    def build(self, input_shape):
        cin = int(input_shape[-1])
        kh, kw = self.kernel_size
        if self.depthwise:
            self.kernel = Run.current.new_variable(self, "kernel", (kh, kw, cin, 1), True, 0.0)
            cout = cin
        else:
            self.kernel = Run.current.new_variable(self, "kernel", (kh, kw, cin, self.filters), True, 0.0)
            cout = self.filters
        self.bias = Run.current.new_variable(self, "bias", (cout,), True, 0.0) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        x = inputs.t.permute(0, 3, 1, 2)                                   # NHWC -> NCHW
        if self.padding == "same":
            ph = _same_pads(x.shape[2], self.kernel_size[0], self.strides[0], self.dilation_rate[0])
            pw = _same_pads(x.shape[3], self.kernel_size[1], self.strides[1], self.dilation_rate[1])
            x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
        k = self.kernel.value
        if self.depthwise:
            w, groups = k.permute(2, 3, 0, 1), x.shape[1]                   # [C, 1, kh, kw]
        else:
            w, groups = k.permute(3, 2, 0, 1), 1                            # [Cout, Cin, kh, kw]
        y = F.conv2d(x.contiguous(), w.contiguous(), None if self.bias is None else self.bias.value, stride=self.strides, dilation=self.dilation_rate, groups=groups)
        return KT(_activation(self.activation, y.permute(0, 2, 3, 1)))

    def get_config(self):
        c = super().get_config()
        c.update(kernel_size=self.kernel_size, strides=self.strides, padding=self.padding, dilation_rate=self.dilation_rate,
                 use_bias=self.use_bias, activation=self.activation)
        if not self.depthwise:
            c["filters"] = self.filters
        return c


class Conv2D(_ConvBase):
    def __init__(self, filters, kernel_size, **kwargs):
        self.filters = int(filters)
        super().__init__(kernel_size, **kwargs)


class DepthwiseConv2D(_ConvBase):
    depthwise = True

    def __init__(self, kernel_size, depth_multiplier=1, **kwargs):
        if depth_multiplier != 1:
            raise ValueError("depth_multiplier")
        super().__init__(kernel_size, **kwargs)


class BatchNormalization(Layer):
    def build(self, input_shape):
        n = int(input_shape[-1])
        new = Run.current.new_variable
        self.gamma, self.beta = new(self, "gamma", (n,), True, 1.0), new(self, "beta", (n,), True, 0.0)
        self.moving_mean, self.moving_variance = new(self, "moving_mean", (n,), False, 0.0), new(self, "moving_variance", (n,), False, 1.0)
        self.built = True

    def call(self, inputs):
        x = inputs.t
        dims = tuple(range(x.dim() - 1))
        if Run.current.training:
            mean = x.mean(dims)
            var = ((x - mean) ** 2).mean(dims)                              # biased
            self.moving_mean.updated = self.moving_mean.value * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            self.moving_variance.updated = self.moving_variance.value * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM)
        else:
            mean, var = self.moving_mean.value, self.moving_variance.value
        return KT((x - mean) * torch.rsqrt(var + BN_EPS) * self.gamma.value + self.beta.value)


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, **kwargs):
        super().__init__(**kwargs)
        self.units, self.activation, self.use_bias = int(units), activation, use_bias

    def build(self, input_shape):
        self.kernel = Run.current.new_variable(self, "kernel", (int(input_shape[-1]), self.units), True, 0.0)
        self.bias = Run.current.new_variable(self, "bias", (self.units,), True, 0.0) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        z = inputs.t @ self.kernel.value
        if self.bias is not None:
            z = z + self.bias.value
        Run.current.logits = z                                             # (Keras 3 keeps it too: ``_keras_logits`` on a sigmoid's output)
        return KT(_activation(self.activation, z))


class Flatten(Layer):
    def call(self, inputs):
        return KT(inputs.t.reshape(inputs.t.shape[0], -1))


class Reshape(Layer):
    def __init__(self, target_shape, **kwargs):
        super().__init__(**kwargs)
        self.target_shape = tuple(target_shape)

    def call(self, inputs):
        return KT(inputs.t.reshape((inputs.t.shape[0],) + self.target_shape))


class Dropout(Layer):
    def __init__(self, rate, **kwargs):
        super().__init__(**kwargs)
        self.rate = float(rate)

    def call(self, inputs):
        run = Run.current
        if not run.training or self.rate == 0.0:
            return inputs
        if run.dropout_mask is None:
            raise ValueError("a training-mode run through Dropout(%g) needs the keep-mask" % self.rate)
        mask = torch.as_tensor(np.asarray(run.dropout_mask), dtype=torch.float64).reshape(inputs.t.shape)
        return KT(inputs.t * mask / (1.0 - self.rate))


class Concatenate(Layer):
    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def call(self, inputs):
        return KT(torch.cat([i.t for i in inputs], dim=self.axis))


def concatenate(inputs, axis=-1, **kwargs):
    return Concatenate(axis=axis, **kwargs)(inputs)


class _Pool2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", **kwargs):
        super().__init__(**kwargs)
        self.pool_size = _pair(pool_size)
        self.strides = self.pool_size if strides is None else _pair(strides)
        self.padding = padding
        if padding != "valid":
            raise ValueError("pooling with padding %r is not restated" % (padding,))

    def call(self, inputs):
        x = inputs.t.permute(0, 3, 1, 2)
        y = self.pool(x, self.pool_size, self.strides)
        return KT(y.permute(0, 2, 3, 1))

    def get_config(self):
        c = super().get_config()
        c.update(pool_size=self.pool_size, strides=self.strides, padding=self.padding)
        return c


class AveragePooling2D(_Pool2D):
    pool = staticmethod(F.avg_pool2d)


class MaxPooling2D(_Pool2D):
    pool = staticmethod(F.max_pool2d)


def _never(name):
    """Classes the reference only names in ``isinstance`` checks on this path."""
    return type(name, (Layer,), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(NotImplementedError(name + " is not restated"))})


def Input(shape=None, batch_size=None, **kwargs):
    x = Run.current.x
    if tuple(x.shape[1:]) != tuple(shape) or (batch_size is not None and x.shape[0] != batch_size):
        raise ValueError("Input(shape=%s, batch_size=%s) against a batch of shape %s" % (shape, batch_size, tuple(x.shape)))
    return KT(x)


class Model:
    def __init__(self, inputs, outputs, **kwargs):
        self.inputs, self.outputs = inputs, outputs


def _tensorflow_stub():
    tf = types.ModuleType("tensorflow")
    keras = types.ModuleType("tensorflow.keras")
    layers = types.ModuleType("tensorflow.keras.layers")
    ops = types.ModuleType("tensorflow.keras.ops")
    for cls in (Layer, Wrapper, Identity, Activation, Conv2D, DepthwiseConv2D, BatchNormalization, Dense, Flatten, Reshape, Dropout,
                Concatenate, AveragePooling2D, MaxPooling2D):
        setattr(layers, cls.__name__, cls)
    for name in ("Conv1D", "DepthwiseConv1D", "SeparableConv1D", "SeparableConv2D", "Conv2DTranspose", "GlobalMaxPooling2D",
                 "GlobalAveragePooling2D"):
        setattr(layers, name, _never(name))
    layers.concatenate, layers.Input = concatenate, Input

    def deserialize(config):
        raise NotImplementedError("keras.layers.deserialize is not on the training path")

    layers.deserialize = deserialize
    ops.expand_dims = lambda x, axis: KT(x.t.unsqueeze(axis))
    ops.transpose = lambda x, axes: KT(x.t.permute(*axes))
    keras.layers, keras.ops, keras.Model = layers, ops, Model
    tf.keras = keras
    tf.float32 = "float32"
    tf.TensorShape = type("TensorShape", (Shape,), {"__new__": lambda cls, s=(): Shape.__new__(cls, tuple(s))})
    tf.split = lambda x, sizes, axis=0: [KT(p) for p in torch.split(x.t, list(sizes), dim=axis)]
    tf.concat = lambda xs, axis: KT(torch.cat([x.t for x in xs], dim=axis))
    tf.identity = lambda x: x
    tf.shape = lambda x: list(x.t.shape)
    tf.zeros = lambda shape, dtype=None: KT(torch.zeros(tuple(shape), dtype=torch.float64))
    tf.zeros_initializer = "zeros"

    def pad(x, paddings, mode="constant"):
        if str(mode).lower() != "constant":
            raise ValueError("tf.pad mode %r" % (mode,))
        flat = []
        for lo, hi in reversed([tuple(p) for p in paddings]):
            flat += [int(lo), int(hi)]
        return KT(F.pad(x.t, flat))

    tf.pad = pad
    return {"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.keras.layers": layers, "tensorflow.keras.ops": ops}


# ------------------------------------------------------------------------------------------------- loading the reference's files

_loaded = {}


def load_reference_model_modules():
    """-> {"mixednet": module, "inception": module}: the reference's files, imported once from REFERENCE_ROOT with the stand-ins
    above bound to ``tensorflow`` / ``absl``.  ``sys.modules`` is left as it was found."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    stubs = _tensorflow_stub()
    absl = types.ModuleType("absl")
    absl_logging = types.ModuleType("absl.logging")
    absl_logging.info = absl_logging.warning = absl_logging.error = lambda *a, **k: None
    absl.logging = absl_logging
    pkg = types.ModuleType("microwakeword")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "microwakeword")]
    stubs.update({"absl": absl, "absl.logging": absl_logging, "microwakeword": pkg})
    before = dict(sys.modules)
    for k in [k for k in sys.modules if k == "microwakeword" or k.startswith("microwakeword.")]:
        del sys.modules[k]
    sys.modules.update(stubs)
    try:
        _loaded["mixednet"] = importlib.import_module("microwakeword.mixednet")
        _loaded["inception"] = importlib.import_module("microwakeword.inception")
        for m in _loaded.values():
            assert os.path.realpath(m.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), m.__file__
    finally:
        for k in [k for k in sys.modules if k not in before]:
            del sys.modules[k]
        for k, v in before.items():
            sys.modules[k] = v
    return _loaded


class Flags(dict):
    """argparse-namespace view of a flag dict (the builders read ``flags.<name>``)."""
    __getattr__ = dict.__getitem__


def run_reference_model(kind, flags, x, values=None, training=True, dropout_mask=None):
    """Executes the reference's ``<kind>.model(flags, shape, batch_size)`` on the batch ``x`` [B, T, 40].
    -> Run (``.variables`` in creation order, ``.logits`` [B, 1], ``.probs`` [B, 1])."""
    mod = load_reference_model_modules()[kind]
    x = np.asarray(x)
    run = Run(x, values, training, dropout_mask)
    Run.current = run
    Layer._count = {}
    try:
        model = mod.model(Flags(flags), tuple(x.shape[1:]), x.shape[0])
    finally:
        Run.current = None
    run.probs = model.outputs.t
    return run


def reference_loss_and_grads(kind, flags, x, y, w, values, dropout_mask=None, loss_fn=None):
    """One training-mode forward of the reference's graph + the gradient of ``loss_fn(z, y, w)`` (the caller's statement of
    train.py:206,288-299) for every trainable variable.  -> (loss, probs [B], [grad per variable or None], Run)."""
    run = run_reference_model(kind, flags, x, values, True, dropout_mask)
    z = run.logits.reshape(-1)
    yt = torch.as_tensor(np.asarray(y, np.float64).reshape(-1))
    wt = torch.as_tensor(np.asarray(w, np.float64).reshape(-1))
    loss = loss_fn(z, yt, wt)
    loss = loss[0] if isinstance(loss, tuple) else loss
    tr = [v for v in run.variables if v.trainable]
    grads = torch.autograd.grad(loss, [v.value for v in tr], allow_unused=True)
    out, it = [], iter(grads)
    for v in run.variables:
        out.append(next(it) if v.trainable else None)
    return float(loss.detach()), run.probs.detach().reshape(-1).numpy(), out, run
