"""Flag parsing, shape derivation and the Keras-order <-> native-order weight mapping of MixedNet.

Mirrors the host-side arithmetic of the reference (no tensors involved):
  * ``parse``                         microwakeword/mixednet.py:25-40
  * ``spectrogram_slices_dropped``    microwakeword/mixednet.py:108-129
  * list-length check / ValueError    microwakeword/mixednet.py:298-305
  * ``_split_channels``               microwakeword/mixednet.py:132-136
  * layer creation order (= Keras ``get_weights()`` order)  microwakeword/mixednet.py:307-386, SURVEY §A.4

Native order (``include/mww.h``): trainable scalars in Keras order with every multi-kernel
MixConv fused into one ``[K_last, C]`` depthwise whose missing leading taps are structural zeros
(mixednet.py:218-230 + strided_drop.py:42 => right alignment), plus a 0/1 gradient mask; BN moving
statistics live in a separate state vector.
"""
from __future__ import annotations

import ast
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

FEATURE_BINS = 40


def parse(text):
    if not text:
        return []
    res = ast.literal_eval(text) if isinstance(text, str) else text
    if isinstance(res, tuple):
        return res
    if isinstance(res, list):
        return tuple(res)
    return [res]


def _flag(flags, name):
    return flags[name] if isinstance(flags, dict) else getattr(flags, name)


def split_channels(total_filters, num_groups):
    split = [total_filters // num_groups for _ in range(num_groups)]
    split[0] += total_filters - sum(split)
    return split


def spectrogram_slices_dropped(flags) -> int:
    dropped = 0
    if _flag(flags, "first_conv_filters") > 0:
        dropped += _flag(flags, "first_conv_kernel_size") - 1
    for repeat, ksize in zip(parse(_flag(flags, "repeat_in_block")), parse(_flag(flags, "mixconv_kernel_sizes"))):
        dropped += (repeat * (max(ksize) - 1)) * _flag(flags, "stride")
    return dropped


@dataclass
class BlockSpec:
    cin: int
    cout: int
    kernel_sizes: Tuple[int, ...]
    group_channels: Tuple[int, ...]
    tin: int
    tout: int

    @property
    def k(self):
        return self.kernel_sizes[-1]


class MixedNetLayout:
    def __init__(self, flags, frames: int):
        pf = list(parse(_flag(flags, "pointwise_filters")))
        rep = list(parse(_flag(flags, "repeat_in_block")))
        ksz = [tuple(k) if isinstance(k, (list, tuple)) else (k,) for k in parse(_flag(flags, "mixconv_kernel_sizes"))]
        res = list(parse(_flag(flags, "residual_connection")))
        for lst in (pf, rep, ksz, res):
            if len(pf) != len(lst):
                raise ValueError("all input lists have to be the same length")  # mixednet.py:298-305
        self.frames = int(frames)
        self.conv1_filters = int(_flag(flags, "first_conv_filters"))
        self.conv1_kernel = int(_flag(flags, "first_conv_kernel_size"))
        self.stride = int(_flag(flags, "stride"))
        unsupported = []
        if self.conv1_filters <= 0:
            unsupported.append("first_conv_filters == 0")
        if any(res):
            unsupported.append("residual_connection")
        if any(r != 1 for r in rep):
            unsupported.append("repeat_in_block != 1")
        if _flag(flags, "spatial_attention"):
            unsupported.append("spatial_attention")
        if _flag(flags, "pooled"):
            unsupported.append("pooled")
        if unsupported:
            raise NotImplementedError("MixedNet options outside the specialised block kernels (GraphMixedNetLayout covers them): "
                                      + ", ".join(unsupported))
        t = (self.frames - self.conv1_kernel) // self.stride + 1
        c = self.conv1_filters
        self.blocks: List[BlockSpec] = []
        for filters, ks in zip(pf, ksz):
            if max(ks) <= 1:
                raise NotImplementedError("blocks without a depthwise convolution (kernel size 1)")
            if any(k > ks[-1] for k in ks):
                raise ValueError("mixconv kernel sizes must be ascending: alignment uses the last one (mixednet.py:227)")
            groups = tuple(split_channels(c, len(ks))) if len(ks) > 1 else (c,)
            tout = t - (ks[-1] - 1)
            if tout <= 0:
                raise ValueError("spectrogram of %d frames is too short for this network" % frames)
            self.blocks.append(BlockSpec(c, int(filters), tuple(int(k) for k in ks), groups, t, tout))
            t, c = tout, int(filters)
        self.t_last, self.c_last = t, c
        # ---- Keras variable list (name, shape, kind) in get_weights() order
        self.keras_vars: List[Tuple[str, Tuple[int, ...], str]] = []
        kv = self.keras_vars
        kv.append(("conv1.kernel", (self.conv1_kernel, 1, FEATURE_BINS, self.conv1_filters), "param"))
        for bi, b in enumerate(self.blocks):
            for gi, (gc, k) in enumerate(zip(b.group_channels, b.kernel_sizes)):
                kv.append(("b%d.dw%d.kernel" % (bi, gi), (k, 1, gc, 1), "param"))
                kv.append(("b%d.dw%d.bias" % (bi, gi), (gc,), "param"))
            kv.append(("b%d.pw.kernel" % bi, (1, 1, b.cin, b.cout), "param"))
            kv.append(("b%d.bn.gamma" % bi, (b.cout,), "param"))
            kv.append(("b%d.bn.beta" % bi, (b.cout,), "param"))
            kv.append(("b%d.bn.moving_mean" % bi, (b.cout,), "state"))
            kv.append(("b%d.bn.moving_variance" % bi, (b.cout,), "state"))
        kv.append(("dense.kernel", (self.t_last * self.c_last, 1), "param"))
        kv.append(("dense.bias", (1,), "param"))
        self.n_params = (self.conv1_kernel * FEATURE_BINS * self.conv1_filters
                         + sum(b.k * b.cin + b.cin + b.cin * b.cout + 2 * b.cout for b in self.blocks)
                         + self.t_last * self.c_last + 1)
        self.n_state = sum(2 * b.cout for b in self.blocks)

    # ---- counts as Keras would report them
    def keras_param_counts(self):
        total = sum(int(np.prod(s)) for _, s, _ in self.keras_vars)
        trainable = sum(int(np.prod(s)) for _, s, kind in self.keras_vars if kind == "param")
        return total, trainable

    def engine_args(self, max_batch):
        return dict(frames=self.frames, conv1_filters=self.conv1_filters, conv1_kernel=self.conv1_kernel,
                    conv1_stride=self.stride, block_filters=[b.cout for b in self.blocks],
                    block_kernel=[b.k for b in self.blocks], max_batch=max_batch)

    # ---- Keras list -> native vectors
    def pack(self, weights: Sequence[np.ndarray]):
        if len(weights) != len(self.keras_vars):
            raise ValueError("expected %d weight arrays, got %d" % (len(self.keras_vars), len(weights)))
        params, state = [], []
        it = iter(zip(self.keras_vars, weights))

        def take(expect_suffix):
            (name, shape, _), w = next(it)
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != tuple(shape) or not name.endswith(expect_suffix):
                raise ValueError("weight %s: expected shape %s, got %s" % (name, shape, w.shape))
            return w

        params.append(take("conv1.kernel").reshape(-1))
        for b in self.blocks:
            dw = np.zeros((b.k, b.cin), np.float32)
            bias = np.zeros(b.cin, np.float32)
            c0 = 0
            for gc, k in zip(b.group_channels, b.kernel_sizes):
                kw = take(".kernel")[:, 0, :, 0]
                dw[b.k - k:, c0:c0 + gc] = kw
                bias[c0:c0 + gc] = take(".bias")
                c0 += gc
            params += [dw.reshape(-1), bias, take("pw.kernel").reshape(-1), take("gamma"), take("beta")]
            state += [take("moving_mean"), take("moving_variance")]
        params += [take("dense.kernel").reshape(-1), take("dense.bias")]
        return np.concatenate(params), np.concatenate(state)

    def unpack(self, params: np.ndarray, state: np.ndarray) -> List[np.ndarray]:
        params = np.asarray(params, np.float32).reshape(-1)
        state = np.asarray(state, np.float32).reshape(-1)
        if params.size != self.n_params or state.size != self.n_state:
            raise ValueError("vector sizes do not match this model")
        out: List[np.ndarray] = []
        po = so = 0

        def p(n):
            nonlocal po
            v = params[po:po + n]
            po += n
            return v

        def s(n):
            nonlocal so
            v = state[so:so + n]
            so += n
            return v

        out.append(p(self.conv1_kernel * FEATURE_BINS * self.conv1_filters).reshape(self.conv1_kernel, 1, FEATURE_BINS, self.conv1_filters).copy())
        for b in self.blocks:
            dw = p(b.k * b.cin).reshape(b.k, b.cin)
            bias = p(b.cin)
            c0 = 0
            for gc, k in zip(b.group_channels, b.kernel_sizes):
                out.append(dw[b.k - k:, c0:c0 + gc].reshape(k, 1, gc, 1).copy())
                out.append(bias[c0:c0 + gc].copy())
                c0 += gc
            out.append(p(b.cin * b.cout).reshape(1, 1, b.cin, b.cout).copy())
            out.append(p(b.cout).copy())
            out.append(p(b.cout).copy())
            out.append(s(b.cout).copy())
            out.append(s(b.cout).copy())
        out.append(p(self.t_last * self.c_last).reshape(-1, 1).copy())
        out.append(p(1).copy())
        return out

    def segments(self):
        """(name, size) of every contiguous piece of the native parameter vector, in order."""
        seg = [("conv1.kernel", self.conv1_kernel * FEATURE_BINS * self.conv1_filters)]
        for i, b in enumerate(self.blocks):
            seg += [("b%d.dw.kernel" % i, b.k * b.cin), ("b%d.dw.bias" % i, b.cin), ("b%d.pw.kernel" % i, b.cin * b.cout),
                    ("b%d.bn.gamma" % i, b.cout), ("b%d.bn.beta" % i, b.cout)]
        seg += [("dense.kernel", self.t_last * self.c_last), ("dense.bias", 1)]
        return seg

    def summary_lines(self):
        yield "input                      [B, %d, %d]" % (self.frames, FEATURE_BINS)
        yield "conv1 %dx1 /%d -> %d, relu    [B, %d, %d]" % (self.conv1_kernel, self.stride, self.conv1_filters, self.blocks[0].tin, self.conv1_filters)
        for i, b in enumerate(self.blocks):
            yield "block %d: mixconv %s + 1x1 %d->%d + BN + relu   [B, %d, %d]" % (i, list(b.kernel_sizes), b.cin, b.cout, b.tout, b.cout)
        yield "flatten + dense(1, sigmoid)  [B, 1]"

    def grad_mask(self) -> np.ndarray:
        parts = [np.ones(self.conv1_kernel * FEATURE_BINS * self.conv1_filters, np.float32)]
        for b in self.blocks:
            m = np.zeros((b.k, b.cin), np.float32)
            c0 = 0
            for gc, k in zip(b.group_channels, b.kernel_sizes):
                m[b.k - k:, c0:c0 + gc] = 1.0
                c0 += gc
            parts += [m.reshape(-1), np.ones(b.cin + b.cin * b.cout + 2 * b.cout, np.float32)]
        parts.append(np.ones(self.t_last * self.c_last + 1, np.float32))
        return np.concatenate(parts)


class InceptionLayout:
    """Shape derivation and Keras-order weight mapping of the reference's Inception model
    (microwakeword/inception.py:232-340) as a list of conv -> BN/SSN -> ReLU ops for
    ``mww_create_convnet`` (include/mww.h).

      * stem: Conv2D(k x 1, valid, no bias) + SubSpectralNormalization + ReLU   inception.py:261-279
      * block: branch1 1x1; branch2 1x1 -> kx1; branch3 1x1 -> kx1 -> kx1; StridedDrop of the leading
        frames of branch1/2 to branch3's length; concatenate; 1x1 reduce         inception.py:281-328
      * Flatten -> Dropout -> Dense(1, sigmoid)                                  inception.py:330-338
      * ``spectrogram_slices_dropped``                                           inception.py:212-230

    ``keras_vars`` is ``get_weights()`` order (layer-creation order).  The native order differs in one
    place: with plain BatchNormalization (sub-spectral groups == 1) the three 1x1 branch heads of a block
    read the same input, so they run as ONE convolution with the concatenated filters (``fuse_heads``);
    its kernel / gamma / beta / moving statistics are the three layers' arrays concatenated along the
    channel axis, and each consumer names its channel slice.  ``pack`` / ``unpack`` do that permutation.
    """

    def __init__(self, flags, frames: int, fuse_heads: bool = True):
        self.frames = int(frames)
        self.dropout = float(_flag(flags, "dropout"))
        stem = list(zip(parse(_flag(flags, "cnn1_filters")), parse(_flag(flags, "cnn1_kernel_sizes")),
                        parse(_flag(flags, "cnn1_subspectral_groups"))))
        blocks = list(zip(parse(_flag(flags, "cnn2_filters1")), parse(_flag(flags, "cnn2_filters2")),
                          parse(_flag(flags, "cnn2_kernel_sizes")), parse(_flag(flags, "cnn2_subspectral_groups")),
                          parse(_flag(flags, "cnn2_dilation"))))
        self.ops: List[dict] = []
        self.op_names: List[str] = []
        self.op_members: List[List[Tuple[str, int, int]]] = []   # per native op: (keras layer name, first channel, width)
        self.keras_vars: List[Tuple[str, Tuple[int, ...], str]] = []
        kidx = {}   # keras layer name -> index of its kernel in keras_vars (gamma +1, beta +2, mean +3, var +4)
        t, c = self.frames, FEATURE_BINS
        cur = (-1, 0, 0)   # (op index, first channel, width) of the running tensor; width 0 = whole

        def keras_layer(name, k, cin, filters, groups):
            if filters % groups:
                # sub_spectral_normalization.py:41-45
                raise ValueError("input_shape[3]: %d must be divisible by self.sub_groups %d " % (filters, groups))
            slots = groups if groups > 1 else filters
            kidx[name] = len(self.keras_vars)
            self.keras_vars.append((name + ".kernel", (int(k), 1, int(cin), int(filters)), "param"))
            self.keras_vars.append((name + ".bn.gamma", (slots,), "param"))
            self.keras_vars.append((name + ".bn.beta", (slots,), "param"))
            self.keras_vars.append((name + ".bn.moving_mean", (slots,), "state"))
            self.keras_vars.append((name + ".bn.moving_variance", (slots,), "state"))

        def add(names, srcs, cin, tin, k, dil, filters_each, groups):
            """one native op computing the Keras layers ``names`` (same input, same shape) side by side"""
            tout = tin - (k - 1) * dil
            if tout <= 0:
                raise ValueError("spectrogram of %d frames is too short for this network" % frames)
            filters = filters_each * len(names)
            self.ops.append(dict(src=[s[0] for s in srcs], drop=[s[3] for s in srcs], slice=[(s[1], s[2]) for s in srcs],
                                 kernel=int(k), dilation=int(dil), filters=int(filters), bn_groups=int(groups), cin=int(cin),
                                 tin=int(tin), tout=int(tout), slots=int(groups if groups > 1 else filters)))
            self.op_names.append("+".join(names))
            self.op_members.append([(n, i * filters_each, filters_each) for i, n in enumerate(names)])
            return len(self.ops) - 1, tout

        for i, (f, k, g) in enumerate(stem):
            keras_layer("stem%d" % i, k, c, f, g)
            oi, t = add(["stem%d" % i], [cur + (0,)], c, t, k, 1, f, g)
            cur, c = (oi, 0, 0), int(f)
        for i, (f1, f2, k, g, dil) in enumerate(blocks):
            n = "i%d." % i
            for nm, kk, ci in (("b1", 1, c), ("b2a", 1, c), ("b2b", k, f1), ("b3a", 1, c), ("b3b", k, f1), ("b3c", k, f1)):
                keras_layer(n + nm, kk, ci, f1, g)
            keras_layer(n + "red", 1, 3 * f1, f2, 1)
            if fuse_heads and g == 1:
                heads, _ = add([n + "b1", n + "b2a", n + "b3a"], [cur + (0,)], c, t, 1, 1, f1, 1)
                b1, b2a, b3a = (heads, 0, f1), (heads, f1, f1), (heads, 2 * f1, f1)
            else:
                b1 = (add([n + "b1"], [cur + (0,)], c, t, 1, 1, f1, g)[0], 0, 0)
                b2a = (add([n + "b2a"], [cur + (0,)], c, t, 1, 1, f1, g)[0], 0, 0)
                b3a = (add([n + "b3a"], [cur + (0,)], c, t, 1, 1, f1, g)[0], 0, 0)
            b2, t2 = add([n + "b2b"], [b2a + (0,)], f1, t, k, dil, f1, g)
            b3b, t3b = add([n + "b3b"], [b3a + (0,)], f1, t, k, dil, f1, g)
            b3, t3 = add([n + "b3c"], [(b3b, 0, 0, 0)], f1, t3b, k, dil, f1, g)
            red, t = add([n + "red"], [b1 + (t - t3,), (b2, 0, 0, t2 - t3), (b3, 0, 0, 0)], 3 * f1, t3, 1, 1, f2, 1)
            cur, c = (red, 0, 0), int(f2)
        if not self.ops:
            raise NotImplementedError("an Inception model without any convolution")
        self.t_last, self.c_last = t, c
        self._dense = len(self.keras_vars)
        self.keras_vars.append(("dense.kernel", (t * c, 1), "param"))
        self.keras_vars.append(("dense.bias", (1,), "param"))
        # native segments: (name, [keras var indices concatenated along the last axis], kind)
        self.native_segs: List[Tuple[str, List[int], str]] = []
        for name, members in zip(self.op_names, self.op_members):
            base = [kidx[m[0]] for m in members]
            for off, suffix, kind in ((0, ".kernel", "param"), (1, ".bn.gamma", "param"), (2, ".bn.beta", "param"),
                                      (3, ".bn.moving_mean", "state"), (4, ".bn.moving_variance", "state")):
                self.native_segs.append((name + suffix, [b + off for b in base], kind))
        self.native_segs.append(("dense.kernel", [self._dense], "param"))
        self.native_segs.append(("dense.bias", [self._dense + 1], "param"))
        self.n_params = sum(int(np.prod(s)) for _, s, kind in self.keras_vars if kind == "param")
        self.n_state = sum(int(np.prod(s)) for _, s, kind in self.keras_vars if kind == "state")

    def keras_param_counts(self):
        total = sum(int(np.prod(s)) for _, s, _ in self.keras_vars)
        return total, self.n_params

    def engine_args(self, max_batch):
        return dict(frames=self.frames, conv_ops=self.ops, dropout=self.dropout, max_batch=max_batch)

    def pack(self, weights: Sequence[np.ndarray]):
        if len(weights) != len(self.keras_vars):
            raise ValueError("expected %d weight arrays, got %d" % (len(self.keras_vars), len(weights)))
        ws = []
        for (name, shape, _), w in zip(self.keras_vars, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError("weight %s: expected shape %s, got %s" % (name, shape, w.shape))
            ws.append(w)
        params, state = [], []
        for _, idxs, kind in self.native_segs:
            v = ws[idxs[0]] if len(idxs) == 1 else np.concatenate([ws[i] for i in idxs], axis=-1)
            (params if kind == "param" else state).append(v.reshape(-1))
        return np.concatenate(params), np.concatenate(state)

    def unpack(self, params: np.ndarray, state: np.ndarray) -> List[np.ndarray]:
        params = np.asarray(params, np.float32).reshape(-1)
        state = np.asarray(state, np.float32).reshape(-1)
        if params.size != self.n_params or state.size != self.n_state:
            raise ValueError("vector sizes do not match this model")
        out: List[Optional[np.ndarray]] = [None] * len(self.keras_vars)
        po = so = 0
        for _, idxs, kind in self.native_segs:
            shapes = [self.keras_vars[i][1] for i in idxs]
            n = sum(int(np.prod(sh)) for sh in shapes)
            if kind == "param":
                flat, po = params[po:po + n], po + n
            else:
                flat, so = state[so:so + n], so + n
            merged = flat.reshape(shapes[0][:-1] + (sum(sh[-1] for sh in shapes),))
            c0 = 0
            for i, sh in zip(idxs, shapes):
                out[i] = merged[..., c0:c0 + sh[-1]].copy()
                c0 += sh[-1]
        return out

    def segments(self):
        return [(name, sum(int(np.prod(self.keras_vars[i][1])) for i in idxs)) for name, idxs, kind in self.native_segs if kind == "param"]

    def grad_mask(self) -> np.ndarray:
        return np.ones(self.n_params, np.float32)

    def summary_lines(self):
        yield "input                                   [B, %d, %d]" % (self.frames, FEATURE_BINS)
        for name, op in zip(self.op_names, self.ops):
            norm = "SSN(%d)" % op["bn_groups"] if op["bn_groups"] > 1 else "BN"
            yield "%-22s conv %dx1 d%d %d->%d + %s + relu   [B, %d, %d]" % (name, op["kernel"], op["dilation"], op["cin"], op["filters"],
                                                                           norm, op["tout"], op["filters"])
        yield "flatten + dropout(%g) + dense(1, sigmoid)   [B, 1]" % self.dropout


def inception_slices_dropped(flags) -> int:
    """inception.py:212-230."""
    dropped = 0
    for kernel_size in parse(_flag(flags, "cnn1_kernel_sizes")):
        dropped += kernel_size - 1
    for kernel_size, dilation in zip(parse(_flag(flags, "cnn2_kernel_sizes")), parse(_flag(flags, "cnn2_dilation"))):
        dropped += 2 * dilation * (kernel_size - 1)
    return dropped


class GraphMixedNetLayout:
    """Any MixedNet flag combination as a conv/BN graph for ``mww_create_convnet`` — the route taken when the
    specialised block kernels do not cover a shape (other filter counts / kernel sizes, ``repeat_in_block`` > 1,
    blocks without a depthwise convolution, ``first_conv_filters`` 0).  Per mixednet.py:307-360:

      first conv  : Conv2D(k1 x 1, stride, valid, no bias) -> ReLU                 -> op(conv, norm none, relu)
      every repeat: MixConv (depthwise + bias, groups fused to one [K_last, C] tap table with structural
                    zero taps, right-aligned like StridedDrop)                      -> op(depthwise, norm bias, linear)
                    Conv2D 1x1 (no bias) -> BatchNormalization -> ReLU              -> op(conv, norm bn, relu)

      residual    : Conv2D 1x1 (no bias) -> BatchNormalization of the block input (mixednet.py:340-345), added to
                    every repeat's BN output before its ReLU after StridedDrop of its leading frames (:354-358)
                                                                                    -> op(conv, norm bn, linear) + ``residual`` links

      heads       : ``spatial_attention`` (SpatialAttention(kernel_size=4), mixednet.py:234-275) and ``pooled`` /
                    ``max_pool`` (global average / max pooling, :372-381) when more than one frame remains -> engine head options
    ``keras_vars`` is ``get_weights()`` order; pack / unpack / grad_mask follow MixedNetLayout's conventions.
    """

    def __init__(self, flags, frames: int):
        pf = list(parse(_flag(flags, "pointwise_filters")))
        rep = list(parse(_flag(flags, "repeat_in_block")))
        ksz = [tuple(k) if isinstance(k, (list, tuple)) else (k,) for k in parse(_flag(flags, "mixconv_kernel_sizes"))]
        res = list(parse(_flag(flags, "residual_connection")))
        for lst in (pf, rep, ksz, res):
            if len(pf) != len(lst):
                raise ValueError("all input lists have to be the same length")  # mixednet.py:298-305
        self.frames = int(frames)
        self.dropout = 0.0
        f0, k0, stride = int(_flag(flags, "first_conv_filters")), int(_flag(flags, "first_conv_kernel_size")), int(_flag(flags, "stride"))
        self.ops: List[dict] = []
        self.op_names: List[str] = []
        self.keras_vars: List[Tuple[str, Tuple[int, ...], str]] = []
        self.items: List[dict] = []   # per op: how its native parameter block maps to Keras variables
        t, c, cur = self.frames, FEATURE_BINS, -1
        if f0 > 0:
            tout = (t - k0) // stride + 1
            if tout <= 0:
                raise ValueError("spectrogram of %d frames is too short for this network" % frames)
            self.ops.append(dict(src=[cur], drop=[0], kernel=k0, filters=f0, stride=stride, norm="none", act="relu", kind="conv",
                                 cin=c, tin=t, tout=tout))
            self.op_names.append("conv1")
            self.items.append(dict(kind="conv", kernel=len(self.keras_vars), shape=(k0, 1, c, f0)))
            self.keras_vars.append(("conv1.kernel", (k0, 1, c, f0), "param"))
            cur, t, c = 0, tout, f0
        for bi, (filters, repeat, ks, r) in enumerate(zip(pf, rep, ksz, res)):
            filters = int(filters)
            if any(k > ks[-1] for k in ks):
                raise ValueError("mixconv kernel sizes must be ascending: alignment uses the last one (mixednet.py:227)")
            res_op, res_t = None, 0
            if r:
                self.ops.append(dict(src=[cur], drop=[0], kernel=1, filters=filters, norm="bn", act="linear", kind="conv",
                                     cin=c, tin=t, tout=t))
                self.op_names.append("b%d.res" % bi)
                self.items.append(dict(kind="pw", kernel=len(self.keras_vars), shape=(1, 1, c, filters)))
                self.keras_vars.append(("b%d.res.kernel" % bi, (1, 1, c, filters), "param"))
                for suffix, kind in (("gamma", "param"), ("beta", "param"), ("moving_mean", "state"), ("moving_variance", "state")):
                    self.keras_vars.append(("b%d.res.bn.%s" % (bi, suffix), (filters,), kind))
                res_op, res_t = len(self.ops) - 1, t
            for ri in range(int(repeat)):
                p = "b%d.r%d" % (bi, ri)
                if max(ks) > 1:
                    K = int(ks[-1])
                    groups = tuple(split_channels(c, len(ks))) if len(ks) > 1 else (c,)
                    tout = t - (K - 1)
                    if tout <= 0:
                        raise ValueError("spectrogram of %d frames is too short for this network" % frames)
                    self.ops.append(dict(src=[cur], drop=[0], kernel=K, filters=c, norm="bias", act="linear", kind="depthwise",
                                         cin=c, tin=t, tout=tout))
                    self.op_names.append(p + ".dw")
                    it = dict(kind="depthwise", K=K, C=c, groups=[])
                    for gi, (gc, k) in enumerate(zip(groups, ks)):
                        it["groups"].append((len(self.keras_vars), int(gc), int(k)))
                        self.keras_vars.append(("%s.dw%d.kernel" % (p, gi), (int(k), 1, int(gc), 1), "param"))
                        self.keras_vars.append(("%s.dw%d.bias" % (p, gi), (int(gc),), "param"))
                    self.items.append(it)
                    cur, t = len(self.ops) - 1, tout
                self.ops.append(dict(src=[cur], drop=[0], kernel=1, filters=filters, norm="bn", act="relu", kind="conv",
                                     cin=c, tin=t, tout=t, residual=res_op, residual_drop=res_t - t))
                self.op_names.append(p + ".pw")
                self.items.append(dict(kind="pw", kernel=len(self.keras_vars), shape=(1, 1, c, filters)))
                self.keras_vars.append((p + ".pw.kernel", (1, 1, c, filters), "param"))
                for suffix, kind in (("gamma", "param"), ("beta", "param"), ("moving_mean", "state"), ("moving_variance", "state")):
                    self.keras_vars.append(("%s.bn.%s" % (p, suffix), (filters,), kind))
                cur, c = len(self.ops) - 1, filters
        if not self.ops or self.ops[-1]["norm"] != "bn":
            raise NotImplementedError("a MixedNet without any block")
        self.head_attention, self.head_pool, self._att = False, 0, None
        if t > 1:   # mixednet.py:362
            if _flag(flags, "spatial_attention"):
                if t < 4:
                    raise ValueError("spatial attention needs at least 4 frames after the last block")
                self.head_attention = True
                self._att = len(self.keras_vars)
                self.keras_vars.append(("attention.kernel", (4, 1, 2, 1), "param"))
                t -= 3
            if _flag(flags, "pooled"):
                self.head_pool = 2 if _flag(flags, "max_pool") else 1
                t = 1
        self.t_last, self.c_last = t, c
        self._dense = len(self.keras_vars)
        self.keras_vars.append(("dense.kernel", (t * c, 1), "param"))
        self.keras_vars.append(("dense.bias", (1,), "param"))
        self.n_params = sum(self._item_size(it) for it in self.items) + (8 if self.head_attention else 0) + t * c + 1
        self.n_state = sum(int(np.prod(s)) for _, s, kind in self.keras_vars if kind == "state")

    @staticmethod
    def _item_size(it):
        if it["kind"] == "depthwise":
            return it["K"] * it["C"] + it["C"]
        n = int(np.prod(it["shape"]))
        return n + (2 * it["shape"][3] if it["kind"] == "pw" else 0)

    def keras_param_counts(self):
        total = sum(int(np.prod(s)) for _, s, _ in self.keras_vars)
        return total, sum(int(np.prod(s)) for _, s, kind in self.keras_vars if kind == "param")

    def engine_args(self, max_batch):
        return dict(frames=self.frames, conv_ops=self.ops, dropout=0.0, max_batch=max_batch,
                    head_attention=self.head_attention, head_pool=self.head_pool)

    def pack(self, weights: Sequence[np.ndarray]):
        if len(weights) != len(self.keras_vars):
            raise ValueError("expected %d weight arrays, got %d" % (len(self.keras_vars), len(weights)))
        ws = []
        for (name, shape, _), w in zip(self.keras_vars, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError("weight %s: expected shape %s, got %s" % (name, shape, w.shape))
            ws.append(w)
        params, state = [], []
        for it in self.items:
            if it["kind"] == "depthwise":
                dw, bias, c0 = np.zeros((it["K"], it["C"]), np.float32), np.zeros(it["C"], np.float32), 0
                for vi, gc, k in it["groups"]:
                    dw[it["K"] - k:, c0:c0 + gc] = ws[vi][:, 0, :, 0]
                    bias[c0:c0 + gc] = ws[vi + 1]
                    c0 += gc
                params += [dw.reshape(-1), bias]
            else:
                params.append(ws[it["kernel"]].reshape(-1))
                if it["kind"] == "pw":
                    params += [ws[it["kernel"] + 1], ws[it["kernel"] + 2]]
                    state += [ws[it["kernel"] + 3], ws[it["kernel"] + 4]]
        if self._att is not None:
            params.append(ws[self._att].reshape(-1))
        params += [ws[self._dense].reshape(-1), ws[self._dense + 1]]
        return np.concatenate(params), (np.concatenate(state) if state else np.zeros(0, np.float32))

    def unpack(self, params: np.ndarray, state: np.ndarray) -> List[np.ndarray]:
        params = np.asarray(params, np.float32).reshape(-1)
        state = np.asarray(state, np.float32).reshape(-1)
        if params.size != self.n_params or state.size != self.n_state:
            raise ValueError("vector sizes do not match this model")
        out: List[Optional[np.ndarray]] = [None] * len(self.keras_vars)
        po = so = 0
        for it in self.items:
            if it["kind"] == "depthwise":
                K, C = it["K"], it["C"]
                dw, bias = params[po:po + K * C].reshape(K, C), params[po + K * C:po + K * C + C]
                po += K * C + C
                c0 = 0
                for vi, gc, k in it["groups"]:
                    out[vi] = dw[K - k:, c0:c0 + gc].reshape(k, 1, gc, 1).copy()
                    out[vi + 1] = bias[c0:c0 + gc].copy()
                    c0 += gc
            else:
                n = int(np.prod(it["shape"]))
                out[it["kernel"]] = params[po:po + n].reshape(it["shape"]).copy()
                po += n
                if it["kind"] == "pw":
                    f = it["shape"][3]
                    out[it["kernel"] + 1], out[it["kernel"] + 2] = params[po:po + f].copy(), params[po + f:po + 2 * f].copy()
                    out[it["kernel"] + 3], out[it["kernel"] + 4] = state[so:so + f].copy(), state[so + f:so + 2 * f].copy()
                    po, so = po + 2 * f, so + 2 * f
        if self._att is not None:
            out[self._att] = params[po:po + 8].reshape(4, 1, 2, 1).copy()
            po += 8
        n = self.t_last * self.c_last
        out[self._dense], out[self._dense + 1] = params[po:po + n].reshape(-1, 1).copy(), params[po + n:po + n + 1].copy()
        return out

    def segments(self):
        seg = []
        for name, it in zip(self.op_names, self.items):
            if it["kind"] == "depthwise":
                seg += [(name + ".kernel", it["K"] * it["C"]), (name + ".bias", it["C"])]
            else:
                seg.append((name + ".kernel", int(np.prod(it["shape"]))))
                if it["kind"] == "pw":
                    seg += [(name + ".bn.gamma", it["shape"][3]), (name + ".bn.beta", it["shape"][3])]
        if self._att is not None:
            seg.append(("attention.kernel", 8))
        return seg + [("dense.kernel", self.t_last * self.c_last), ("dense.bias", 1)]

    def grad_mask(self) -> np.ndarray:
        parts = []
        for it in self.items:
            if it["kind"] == "depthwise":
                m, c0 = np.zeros((it["K"], it["C"]), np.float32), 0
                for _, gc, k in it["groups"]:
                    m[it["K"] - k:, c0:c0 + gc] = 1.0
                    c0 += gc
                parts += [m.reshape(-1), np.ones(it["C"], np.float32)]
            else:
                parts.append(np.ones(self._item_size(it), np.float32))
        parts.append(np.ones((8 if self._att is not None else 0) + self.t_last * self.c_last + 1, np.float32))
        return np.concatenate(parts)

    def summary_lines(self):
        yield "input                                   [B, %d, %d]" % (self.frames, FEATURE_BINS)
        for name, op in zip(self.op_names, self.ops):
            yield "%-12s %-9s %dx1 %s %d->%d, %s, %s   [B, %d, %d]" % (name, op["kind"], op["kernel"], "/%d" % op.get("stride", 1), op["cin"],
                                                                       op["filters"], op["norm"], op["act"], op["tout"], op["filters"])
        head = ("spatial attention(4) + " if self.head_attention else "") + {0: "flatten", 1: "average pool", 2: "max pool"}[self.head_pool]
        yield "%s + dense(1, sigmoid)   [B, 1]" % head
