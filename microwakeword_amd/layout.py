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
from typing import List, Sequence, Tuple

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
            raise NotImplementedError("MixedNet options not implemented by the MI355X engine yet: " + ", ".join(unsupported))
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
