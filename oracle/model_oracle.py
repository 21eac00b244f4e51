"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch-CPU, fp64 or fp32) of the model half of the
reference train step.

Pinning.  **The graph is pinned to the reference's own code by execution** (round 6):
``oracle/ref_model_shim.py`` runs ``/root/reference/microwakeword/mixednet.py`` / ``inception.py`` and the layer files
they import, unmodified, and this module agrees with what they compute to 1e-11 (probabilities in both modes, loss,
every gradient, BN moving statistics, variable order and shapes) on the hand-picked and random topologies of
``tests/test_reference_graph.py``; the frozen outputs are ``tests/golden/ref_graph_golden.npz``
(``tests/golden/make_golden_ref_graph.py``).  **The layer primitives stay unpinned against TensorFlow**: the
reference's arithmetic lives in TensorFlow/Keras (``tensorflow>=2.16`` ⇒ Keras 3, reference ``setup.py:18``,
un-vendored, not installable here) and the reference has no tests/golden vectors for this path, so that run uses
stand-ins restating the published Keras semantics of Conv2D / DepthwiseConv2D / BatchNormalization / Dense / pooling
(the shim's header lists them), as this file does; loss, optimizer and metrics are pinned only by self-consistency tests
(``tests/test_model_oracle.py``: finite differences, BN train/eval consistency, layer shapes/param counts from SURVEY
§A.2/A.3) and by the independent numpy twin (``oracle/model_oracle_np.py``).  "Parity unpinned" therefore still
applies to: Keras' primitive arithmetic, BinaryCrossentropy / Adam / metric details, the ``[B,B]`` weight reduction.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path never does.

Restated (reference file:line):
  * graph of ``mixednet.model``              microwakeword/mixednet.py:278-386
  * ``MixConv`` split / right-alignment      microwakeword/mixednet.py:132-136,168-231
  * ``StridedDrop`` (drops LEADING frames)   microwakeword/layers/strided_drop.py:40-44
  * ``Stream`` in training = pass-through    microwakeword/layers/stream.py:654-695
  * graph of ``inception.model``             microwakeword/inception.py:46-143,233-338
  * ``SubSpectralNormalization``             microwakeword/layers/sub_spectral_normalization.py:38-62
  * loss / optimizer / metrics / weights     microwakeword/train.py:206-223,288-299
  * shape derivation                          microwakeword/model_train_eval.py:60-94,
                                              mixednet.py:108-129, inception.py:212-230
Keras-3 layer semantics (BatchNormalization momentum .99 / eps 1e-3 / biased moving variance,
Adam epsilon placement, BinaryCrossentropy clipping, confusion-matrix bucketing) follow the
Keras documentation/behaviour summarised in SURVEY Appendix A.
"""
from __future__ import annotations

import ast
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
KERAS_EPS = 1e-7
FEATURE_BINS = 40


# ----------------------------------------------------------------------------- flags / shapes

def parse(text):
    """mixednet.py:25-40 / inception.py:28-43."""
    if not text:
        return []
    res = ast.literal_eval(text) if isinstance(text, str) else text
    return list(res) if isinstance(res, (tuple, list)) else [res]


def _get(flags, name, default=None):
    if isinstance(flags, dict):
        return flags.get(name, default)
    return getattr(flags, name, default)


MIXEDNET_DEFAULTS = dict(pointwise_filters="48, 48, 48, 48", residual_connection="0,0,0,0,0",
                         repeat_in_block="1,1,1,1", mixconv_kernel_sizes="[5], [9], [13], [21]",
                         max_pool=0, first_conv_filters=32, first_conv_kernel_size=3,
                         spatial_attention=0, pooled=0, stride=1)
INCEPTION_DEFAULTS = dict(cnn1_filters="24", cnn1_kernel_sizes="5", cnn1_subspectral_groups="4",
                          cnn2_filters1="10,10,16", cnn2_filters2="10,10,16", cnn2_kernel_sizes="5,5,5",
                          cnn2_subspectral_groups="1,1,1", cnn2_dilation="1,1,1", dropout=0.2)


def mixednet_slices_dropped(flags) -> int:
    """mixednet.py:108-129."""
    dropped = 0
    if _get(flags, "first_conv_filters") > 0:
        dropped += _get(flags, "first_conv_kernel_size") - 1
    for repeat, ks in zip(parse(_get(flags, "repeat_in_block")), parse(_get(flags, "mixconv_kernel_sizes"))):
        dropped += (repeat * (max(ks) - 1)) * _get(flags, "stride")
    return dropped


def inception_slices_dropped(flags) -> int:
    """inception.py:212-230."""
    dropped = sum(k - 1 for k in parse(_get(flags, "cnn1_kernel_sizes")))
    for k, dil in zip(parse(_get(flags, "cnn2_kernel_sizes")), parse(_get(flags, "cnn2_dilation"))):
        dropped += 2 * dil * (k - 1)
    return dropped


def spectrogram_length(clip_duration_ms, window_step_ms, stride, dropped) -> Tuple[int, int]:
    """model_train_eval.py:60-88 -> (final_layer_length, spectrogram_length)."""
    desired = int(16000 * clip_duration_ms / 1000)
    window = int(16000 * 30 / 1000)
    step = int(stride * 16000 * window_step_ms / 1000)
    lmw = desired - window
    final = 0 if lmw < 0 else 1 + int(lmw / step)
    return final, final + dropped


def split_channels(total, groups):
    """mixednet.py:132-136."""
    split = [total // groups for _ in range(groups)]
    split[0] += total - sum(split)
    return split


# ----------------------------------------------------------------------------- weights container

@dataclass
class Var:
    name: str
    value: np.ndarray  # Keras layout
    trainable: bool = True


def glorot_uniform(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn_vars(prefix, n):
    return [Var(prefix + ".gamma", np.ones(n, np.float32)), Var(prefix + ".beta", np.zeros(n, np.float32)),
            Var(prefix + ".moving_mean", np.zeros(n, np.float32), False),
            Var(prefix + ".moving_variance", np.ones(n, np.float32), False)]


def mixednet_build(flags, T, seed=42) -> List[Var]:
    """Variables in Keras ``get_weights()`` order (layer-creation order), initialised as
    §8(d) prescribes (glorot-uniform kernels, zero biases, BN 1/0/0/1; ``default_rng(seed)``).
    Raises ValueError exactly where mixednet.py:298-305 does."""
    pf = parse(_get(flags, "pointwise_filters"))
    rep = parse(_get(flags, "repeat_in_block"))
    ksz = parse(_get(flags, "mixconv_kernel_sizes"))
    res = parse(_get(flags, "residual_connection"))
    for lst in (pf, rep, ksz, res):
        if len(pf) != len(lst):
            raise ValueError("all input lists have to be the same length")
    rng = np.random.default_rng(seed)
    vs: List[Var] = []
    c = FEATURE_BINS
    t = T
    f0, k0, stride = _get(flags, "first_conv_filters"), _get(flags, "first_conv_kernel_size"), _get(flags, "stride")
    if f0 > 0:
        vs.append(Var("conv1.kernel", glorot_uniform(rng, (k0, 1, c, f0), k0 * c, k0 * f0)))
        t = (t - k0) // stride + 1
        c = f0
    for bi, (filters, repeat, ks, r) in enumerate(zip(pf, rep, ksz, res)):
        ks = list(ks)
        if r:
            vs.append(Var("b%d.res.kernel" % bi, glorot_uniform(rng, (1, 1, c, filters), c, filters)))
            vs += _bn_vars("b%d.res.bn" % bi, filters)
        for ri in range(repeat):
            p = "b%d.r%d" % (bi, ri)
            if max(ks) > 1:
                groups = split_channels(c, len(ks)) if len(ks) > 1 else [c]
                for gi, (gc, k) in enumerate(zip(groups, ks)):
                    # Keras DepthwiseConv2D: kernel [k,1,C,1], glorot fans = (k*C, k*1)... Keras computes
                    # fans from the kernel shape (k,1,C,1): fan_in = k*1*C, fan_out = k*1*1
                    vs.append(Var("%s.dw%d.kernel" % (p, gi), glorot_uniform(rng, (k, 1, gc, 1), k * gc, k)))
                    vs.append(Var("%s.dw%d.bias" % (p, gi), np.zeros(gc, np.float32)))
                t = t - max(ks) + 1 if len(ks) == 1 else t - ks[-1] + 1
            vs.append(Var(p + ".pw.kernel", glorot_uniform(rng, (1, 1, c, filters), c, filters)))
            vs += _bn_vars(p + ".bn", filters)
            c = filters
    # head (mixednet.py:362-384): SpatialAttention(kernel_size=4) and/or global pooling, only if frames remain
    if t > 1:
        if _get(flags, "spatial_attention"):
            vs.append(Var("attention.kernel", glorot_uniform(rng, (4, 1, 2, 1), 4 * 2, 4 * 1)))
            t = t - 3
        if _get(flags, "pooled"):
            t = 1
    if t < 1:
        # (Keras fails inside a valid-padding convolution here; the harness wants one recognisable error)
        raise ValueError("spectrogram too short for this kernel stack: %d frames left before the dense layer" % t)
    vs.append(Var("dense.kernel", glorot_uniform(rng, (t * c, 1), t * c, 1)))
    vs.append(Var("dense.bias", np.zeros(1, np.float32)))
    return vs


def inception_build(flags, T, seed=42) -> List[Var]:
    rng = np.random.default_rng(seed)
    vs: List[Var] = []
    c, t = FEATURE_BINS, T
    for i, (f, k, g) in enumerate(zip(parse(_get(flags, "cnn1_filters")), parse(_get(flags, "cnn1_kernel_sizes")),
                                      parse(_get(flags, "cnn1_subspectral_groups")))):
        vs.append(Var("stem%d.kernel" % i, glorot_uniform(rng, (k, 1, c, f), k * c, k * f)))
        vs += _bn_vars("stem%d.bn" % i, g if g > 1 else f)
        t, c = t - k + 1, f
    for i, (f1, f2, k, g, dil) in enumerate(zip(*(parse(_get(flags, n)) for n in (
            "cnn2_filters1", "cnn2_filters2", "cnn2_kernel_sizes", "cnn2_subspectral_groups", "cnn2_dilation")))):
        def conv(name, kk, cin, cout):
            vs.append(Var("i%d.%s.kernel" % (i, name), glorot_uniform(rng, (kk, 1, cin, cout), kk * cin, kk * cout)))
            vs.extend(_bn_vars("i%d.%s.bn" % (i, name), g if g > 1 else cout))
        conv("b1", 1, c, f1)
        conv("b2a", 1, c, f1)
        conv("b2b", k, f1, f1)
        conv("b3a", 1, c, f1)
        conv("b3b", k, f1, f1)
        conv("b3c", k, f1, f1)
        vs.append(Var("i%d.red.kernel" % i, glorot_uniform(rng, (1, 1, 3 * f1, f2), 3 * f1, f2)))
        vs.extend(_bn_vars("i%d.red.bn" % i, f2))
        t, c = t - 2 * dil * (k - 1), f2
    vs.append(Var("dense.kernel", glorot_uniform(rng, (t * c, 1), t * c, 1)))
    vs.append(Var("dense.bias", np.zeros(1, np.float32)))
    return vs


# ----------------------------------------------------------------------------- forward graphs

def _round_bf16(t):
    """RNE to bfloat16 of the float32 value (what v_cvt_pk_bf16_f32 does to the engine's fp32 operands)."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _Bf16Pointwise(torch.autograd.Function):
    """1x1 convolution in the optional "bf16 with MFMA pointwise" mode (BASELINE configs[4]): every
    contraction — forward, input gradient, weight gradient — takes bf16-rounded operands and
    accumulates exactly.  x [B,Cin,T], w [Cout,Cin]."""

    @staticmethod
    def forward(ctx, x, w):
        xr, wr = _round_bf16(x), _round_bf16(w)
        ctx.save_for_backward(xr, wr)
        return torch.einsum("oc,bct->bot", wr, xr)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = _round_bf16(gy)
        return torch.einsum("oc,bot->bct", wr, gr), torch.einsum("bot,bct->oc", gr, xr)


class _StoredBatchNorm(torch.autograd.Function):
    """Training-mode BatchNorm of a block output under the engine's "storage_bf16" option (BASELINE configs[4]):
    the block output p lives in HBM as bf16, so every consumer sees round(p); the batch statistics are summed
    in-kernel from the unrounded fp32 values.  Backward as the engine evaluates it: the sums (sum g, sum g*xhat)
    come from the unrounded incoming gradient g while it is still in registers; the gradient that is stashed for
    the next launch is round(g) (``round_g``: every block but the last, whose g is formed from the dense kernel).
    x [B,C,T]; returns (y, mean, var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, round_g):
        mean = x.mean((0, 2))
        var = ((x - mean.reshape(1, -1, 1)) ** 2).mean((0, 2))
        rs = torch.rsqrt(var + BN_EPS)
        xhat = (_round_bf16(x) - mean.reshape(1, -1, 1)) * rs.reshape(1, -1, 1)
        ctx.save_for_backward(xhat, gamma, rs)
        ctx.round_g = round_g
        ctx.mark_non_differentiable(mean, var)
        return xhat * gamma.reshape(1, -1, 1) + beta.reshape(1, -1, 1), mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        xhat, gamma, rs = ctx.saved_tensors
        n = gy.shape[0] * gy.shape[2]
        s1, s2 = gy.sum((0, 2)), (gy * xhat).sum((0, 2))
        gs = _round_bf16(gy) if ctx.round_g else gy
        dx = (gamma * rs).reshape(1, -1, 1) * (gs - (s1 / n).reshape(1, -1, 1) - xhat * (s2 / n).reshape(1, -1, 1))
        return dx, s2, s1, None


class _Cursor:
    def __init__(self, tensors: Dict[str, torch.Tensor], training: bool):
        self.t = tensors
        self.training = training
        self.new_stats: Dict[str, torch.Tensor] = {}
        self.taps: Dict[str, torch.Tensor] = {}

    def conv(self, x, name, stride=1, dilation=1):
        """Keras Conv2D kernel [k,1,Cin,Cout], cross-correlation, valid, no bias; x is [B,C,T]."""
        w = self.t[name + ".kernel"][:, 0].permute(2, 1, 0)  # [Cout,Cin,k]
        return F.conv1d(x, w, stride=stride, dilation=dilation)

    def depthwise(self, x, name):
        w = self.t[name + ".kernel"][:, 0, :, 0].t().unsqueeze(1)  # [C,1,k]
        return F.conv1d(x, w, bias=self.t[name + ".bias"], groups=x.shape[1])

    def bn(self, x, name, groups=1):
        """Keras BatchNormalization(axis=-1); groups>1 = SubSpectralNormalization with W=1:
        channel c uses slot c % groups (sub_spectral_normalization.py:49-61)."""
        g, b = self.t[name + ".gamma"], self.t[name + ".beta"]
        B, C, T = x.shape
        if groups > 1:
            xs = x.reshape(B, C // groups, groups, T)
            dims = (0, 1, 3)
            shape = (1, 1, groups, 1)
        else:
            xs = x
            dims = (0, 2)
            shape = (1, C, 1)
        if self.training:
            mean = xs.mean(dims)
            var = ((xs - mean.reshape(shape)) ** 2).mean(dims)  # biased
            self.new_stats[name + ".moving_mean"] = self.t[name + ".moving_mean"] * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            self.new_stats[name + ".moving_variance"] = self.t[name + ".moving_variance"] * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM)
        else:
            mean, var = self.t[name + ".moving_mean"], self.t[name + ".moving_variance"]
        y = (xs - mean.reshape(shape)) * torch.rsqrt(var.reshape(shape) + BN_EPS) * g.reshape(shape) + b.reshape(shape)
        return y.reshape(B, C, T)

    def bn_stored(self, x, name, round_g):
        """BatchNorm over a block output that is stored as bf16 (see _StoredBatchNorm)."""
        g, b = self.t[name + ".gamma"], self.t[name + ".beta"]
        if self.training:
            y, mean, var = _StoredBatchNorm.apply(x, g, b, round_g)
            self.new_stats[name + ".moving_mean"] = self.t[name + ".moving_mean"] * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            self.new_stats[name + ".moving_variance"] = self.t[name + ".moving_variance"] * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM)
            return y
        mean, var = self.t[name + ".moving_mean"], self.t[name + ".moving_variance"]
        return (_round_bf16(x) - mean.reshape(1, -1, 1)) * torch.rsqrt(var.reshape(1, -1, 1) + BN_EPS) * g.reshape(1, -1, 1) + b.reshape(1, -1, 1)


def mixednet_logits(flags, tensors, x, training, taps=None, relu_masks=None):
    """x [B,T,40] -> logits [B].  ``taps`` (dict) receives named intermediates ([B,T,C] layout).
    ``relu_masks`` (test aid, see inception_logits): {"b<i>.r<j>": bool [B,C,T]} replaces the ReLU
    decisions of the named block outputs."""
    cur = _Cursor(tensors, training)
    net = x.transpose(1, 2)  # [B,40,T]
    f0, stride = _get(flags, "first_conv_filters"), _get(flags, "stride")
    if f0 > 0:
        net = cur.conv(net, "conv1", stride=stride)
        if taps is not None:
            taps["conv1.pre"] = net.transpose(1, 2)
        if relu_masks is not None and "conv1" in relu_masks:
            net = net * relu_masks["conv1"].to(net.dtype)   # (test aid, as for the block outputs below)
        else:
            net = torch.relu(net)
        if taps is not None:
            taps["conv1"] = net.transpose(1, 2)
    pf, rep = parse(_get(flags, "pointwise_filters")), parse(_get(flags, "repeat_in_block"))
    ksz, res = parse(_get(flags, "mixconv_kernel_sizes")), parse(_get(flags, "residual_connection"))
    for bi, (filters, repeat, ks, r) in enumerate(zip(pf, rep, ksz, res)):
        ks = list(ks)
        if r:
            residual = cur.bn(cur.conv(net, "b%d.res" % bi), "b%d.res.bn" % bi)
        for ri in range(repeat):
            p = "b%d.r%d" % (bi, ri)
            if max(ks) > 1:
                if len(ks) == 1:
                    net = cur.depthwise(net, p + ".dw0")
                else:
                    outs = []
                    parts = torch.split(net, split_channels(net.shape[1], len(ks)), dim=1)
                    for gi, part in enumerate(parts):
                        outs.append(cur.depthwise(part, "%s.dw%d" % (p, gi)))
                    last_t = outs[-1].shape[2]
                    # StridedDrop drops LEADING frames -> right alignment (strided_drop.py:42)
                    net = torch.cat([o[:, :, o.shape[2] - last_t:] for o in outs], dim=1)
                if taps is not None:
                    taps[p + ".dw"] = net.transpose(1, 2)
            stored = _get(flags, "st_bf16", False)
            if stored or _get(flags, "pw_bf16", False):
                net = _Bf16Pointwise.apply(net, tensors[p + ".pw.kernel"][0, 0].t())
            else:
                net = cur.conv(net, p + ".pw")
            if taps is not None:
                taps[p + ".pre_bn"] = net.transpose(1, 2)
            if stored:
                net = cur.bn_stored(net, p + ".bn", round_g=not (bi == len(pf) - 1 and ri == repeat - 1))
            else:
                net = cur.bn(net, p + ".bn")
            if r:
                residual = residual[:, :, residual.shape[2] - net.shape[2]:]
                net = net + residual
            if taps is not None:
                taps[p + ".bn_out"] = net.transpose(1, 2)
            if relu_masks is not None and p in relu_masks:
                net = net * relu_masks[p].to(net.dtype)
            else:
                net = torch.relu(net)
    if net.shape[2] > 1:
        if _get(flags, "spatial_attention"):
            # mixednet.py:244-275: per-frame mean and max over the channels -> Conv2D(1, (4,1), valid, no bias,
            # sigmoid) over time -> gate for the LAST T-3 frames of the input
            avg, mx = net.mean(dim=1, keepdim=True), net.max(dim=1, keepdim=True).values          # [B,1,T]
            wa = tensors["attention.kernel"][:, 0, :, 0].t().unsqueeze(0)                             # [1,2,4]
            att = torch.sigmoid(F.conv1d(torch.cat([avg, mx], dim=1), wa))                            # [B,1,T-3]
            net = net[:, :, net.shape[2] - att.shape[2]:] * att
            if taps is not None:
                taps["attention.out"] = net.transpose(1, 2)
        if _get(flags, "pooled"):
            # mixednet.py:372-381: pooling over the whole remaining time axis
            net = net.max(dim=2, keepdim=True).values if _get(flags, "max_pool") else net.mean(dim=2, keepdim=True)
    flat = net.transpose(1, 2).reshape(net.shape[0], -1)  # Keras Flatten of [B,T,1,C]: index t*C+c
    z = flat @ tensors["dense.kernel"][:, 0] + tensors["dense.bias"][0]
    return z, cur.new_stats


def inception_logits(flags, tensors, x, training, dropout_mask=None, taps=None, relu_masks=None):
    """``relu_masks`` (test aid): {op name: bool [B,C,T]} replaces the ReLU decisions ``v > 0`` of the
    named ops — used to compare gradients under identical decisions when an activation sits within
    float32 rounding of zero (the decision itself is checked separately)."""
    cur = _Cursor(tensors, training)

    def act(v, name):
        if taps is not None:
            taps[name + ".bn_out"] = v.transpose(1, 2)
        if relu_masks is not None and name in relu_masks:
            return v * relu_masks[name].to(v.dtype)
        return torch.relu(v)

    net = x.transpose(1, 2)
    def tap(name, t):
        if taps is not None:
            taps[name + ".pre_bn"] = t.transpose(1, 2)
        return t

    for i, g in enumerate(parse(_get(flags, "cnn1_subspectral_groups"))):
        net = act(cur.bn(tap("stem%d" % i, cur.conv(net, "stem%d" % i)), "stem%d.bn" % i, g), "stem%d" % i)
    for i, (g, dil) in enumerate(zip(parse(_get(flags, "cnn2_subspectral_groups")), parse(_get(flags, "cnn2_dilation")))):
        def cb(inp, name, d=1):
            return act(cur.bn(tap("i%d.%s" % (i, name), cur.conv(inp, "i%d.%s" % (i, name), dilation=d)), "i%d.%s.bn" % (i, name), g),
                       "i%d.%s" % (i, name))
        b1 = cb(net, "b1")
        b2 = cb(cb(net, "b2a"), "b2b", dil)
        b3 = cb(cb(cb(net, "b3a"), "b3b", dil), "b3c", dil)
        t3 = b3.shape[2]
        net = torch.cat([b1[:, :, b1.shape[2] - t3:], b2[:, :, b2.shape[2] - t3:], b3], dim=1)
        net = act(cur.bn(tap("i%d.red" % i, cur.conv(net, "i%d.red" % i)), "i%d.red.bn" % i), "i%d.red" % i)
    flat = net.transpose(1, 2).reshape(net.shape[0], -1)
    if training and _get(flags, "dropout", 0.0) > 0:
        keep = 1.0 - _get(flags, "dropout")
        if dropout_mask is None:
            raise ValueError("training-mode inception needs an explicit dropout keep-mask [B, T*C]")
        flat = flat * dropout_mask / keep  # inverted dropout
    z = flat @ tensors["dense.kernel"][:, 0] + tensors["dense.bias"][0]
    return z, cur.new_stats


# ----------------------------------------------------------------------------- loss / Adam / metrics

def keras_bce(p, y):
    """Probability form of Keras 3 ``binary_crossentropy(from_logits=False)``: clip to [eps, 1-eps]
    (train.py:206 WITHOUT the cached logits; ``BCE_FROM_LOGITS = False``)."""
    # the reference clips float32 probabilities with float32 bounds: 1 - 1e-7 is 0.99999988 there, not 0.9999999
    lo, hi = float(np.float32(KERAS_EPS)), float(np.float32(1.0) - np.float32(KERAS_EPS))
    pc = torch.clamp(p, lo, hi)
    return -(y * torch.log(pc) + (1.0 - y) * torch.log(1.0 - pc))


def keras_bce_logits(z, y):
    """What train.py:206 runs under Keras 3 + TensorFlow: ``activations.sigmoid`` caches its input on the
    output tensor (``_keras_logits``), the TF backend's ``binary_crossentropy`` finds it (``_get_logits``) and
    evaluates ``tf.nn.sigmoid_cross_entropy_with_logits``: max(z,0) - z*y + log1p(exp(-|z|)), no clipping,
    gradient sigmoid(z) - y everywhere."""
    return torch.clamp(z, min=0.0) - z * y + torch.log1p(torch.exp(-torch.abs(z)))


BCE_FROM_LOGITS = True   # the engine's default ("bce_from_logits" option); tests flip both together


def weighted_loss(z, y, w):
    """sum_over_batch_size reduction: sum(w_i * bce_i) / B (NOT / sum(w))."""
    p = torch.sigmoid(z)
    bce = keras_bce_logits(z, y) if BCE_FROM_LOGITS else keras_bce(p, y)
    return (bce * w).sum() / z.shape[0], p


class KerasAdam:
    """tf.keras.optimizers.Adam() defaults (train.py:207); epsilon OUTSIDE the bias correction
    (SURVEY §A.6)."""

    def __init__(self, shapes, dtype=torch.float64, beta1=0.9, beta2=0.999, eps=1e-7):
        self.m = [torch.zeros(s, dtype=dtype) for s in shapes]
        self.v = [torch.zeros(s, dtype=dtype) for s in shapes]
        self.t = 0
        self.b1, self.b2, self.eps = beta1, beta2, eps

    def apply(self, params: List[torch.Tensor], grads: List[torch.Tensor], lr: float):
        self.t += 1
        alpha = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = []
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m += (g - m) * (1.0 - self.b1)
            v += (g * g - v) * (1.0 - self.b2)
            out.append(p - alpha * m / (torch.sqrt(v) + self.eps))
        return out


class Metrics:
    """The nine compiled metrics of train.py:209-221 as cumulative state.  TP/FP/TN/FN @101 and
    AUC @200 use Keras's evenly-spaced-threshold bucketing (bucket = ceil(p*(n-1)) - 1 in fp32;
    bucket -1 dropped when the threshold list has no epsilon ends, clamped to 0 when it has)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.hist101 = np.zeros((2, 101), np.float64)
        self.hist200 = np.zeros((2, 200), np.float64)
        self.n = 0.0
        self.correct = 0.0
        self.tp5 = self.fp5 = self.fn5 = 0.0
        self.bce_sum = 0.0
        self.lab = np.zeros(2, np.float64)

    def update(self, p, y, z=None):
        """z: the logits behind p (loss metric in the logits form); None: probability form with the Keras clip."""
        p = np.asarray(p, np.float32).reshape(-1)
        y = np.asarray(y).reshape(-1) > 0.5
        pc = np.clip(p, np.float32(0.0), np.float32(1.0))
        for n, hist, eps_ends in ((101, self.hist101, False), (200, self.hist200, True)):
            b = (np.ceil(pc * np.float32(n - 1)) - np.float32(1.0)).astype(np.int32)
            if eps_ends:
                b = np.maximum(b, 0)
            for lab in (0, 1):
                sel = b[(y == bool(lab)) & (b >= 0)]
                np.add.at(hist[lab], sel, 1.0)
        pos = p > np.float32(0.5)
        self.n += p.size
        self.correct += float(np.sum(pos == y))
        self.tp5 += float(np.sum(pos & y))
        self.fp5 += float(np.sum(pos & ~y))
        self.fn5 += float(np.sum(~pos & y))
        self.lab += np.array([np.sum(~y), np.sum(y)], np.float64)
        if z is not None and BCE_FROM_LOGITS:
            zz = np.asarray(z, np.float64).reshape(-1)
            self.bce_sum += float(np.sum(np.maximum(zz, 0.0) - zz * y + np.log1p(np.exp(-np.abs(zz)))))
        else:
            pcl = np.clip(p.astype(np.float64), float(np.float32(KERAS_EPS)), float(np.float32(1.0) - np.float32(KERAS_EPS)))
            self.bce_sum += float(np.sum(-(y * np.log(pcl) + (~y) * np.log(1 - pcl))))

    @staticmethod
    def _div(a, b):
        return np.where(b != 0, a / np.where(b != 0, b, 1), 0.0)

    def result(self):
        tp = np.cumsum(self.hist101[1][::-1])[::-1]
        fp = np.cumsum(self.hist101[0][::-1])[::-1]
        fn, tn = self.lab[1] - tp, self.lab[0] - fp
        tp2 = np.cumsum(self.hist200[1][::-1])[::-1]
        fp2 = np.cumsum(self.hist200[0][::-1])[::-1]
        fn2, tn2 = self.lab[1] - tp2, self.lab[0] - fp2
        rec = self._div(tp2, tp2 + fn2)
        fpr = self._div(fp2, fp2 + tn2)
        auc = float(np.sum((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0))
        return dict(accuracy=float(self._div(self.correct, self.n)), recall=float(self._div(self.tp5, self.tp5 + self.fn5)),
                    precision=float(self._div(self.tp5, self.tp5 + self.fp5)), tp=tp, fp=fp, tn=tn, fn=fn, auc=auc,
                    loss=float(self._div(self.bce_sum, self.n)))


# ----------------------------------------------------------------------------- a whole model

class OracleModel:
    """Holds variables (Keras order), runs forward / train steps on CPU."""

    def __init__(self, kind: str, flags, T: int, seed=42, dtype=torch.float64):
        self.kind, self.flags, self.T, self.dtype = kind, flags, T, dtype
        self.vars = (mixednet_build if kind == "mixednet" else inception_build)(flags, T, seed)
        self.adam: Optional[KerasAdam] = None
        self.metrics = Metrics()

    # -- weights
    def get_weights(self):
        return [v.value.copy() for v in self.vars]

    def set_weights(self, ws):
        assert len(ws) == len(self.vars)
        for v, w in zip(self.vars, ws):
            assert v.value.shape == tuple(np.shape(w)), (v.name, v.value.shape, np.shape(w))
            v.value = np.array(w, np.float32)

    def n_params(self):
        return sum(v.value.size for v in self.vars), sum(v.value.size for v in self.vars if v.trainable)

    def _tensors(self, requires_grad):
        t = {}
        for v in self.vars:
            x = torch.tensor(v.value, dtype=self.dtype)
            if requires_grad and v.trainable:
                x.requires_grad_(True)
            t[v.name] = x
        return t

    def logits(self, x, training=False, tensors=None, dropout_mask=None, taps=None, relu_masks=None):
        tensors = tensors or self._tensors(False)
        x = torch.as_tensor(np.asarray(x), dtype=self.dtype)
        rm = None if relu_masks is None else {k: torch.as_tensor(np.asarray(v)) for k, v in relu_masks.items()}
        if self.kind == "mixednet":
            return mixednet_logits(self.flags, tensors, x, training, taps, rm)
        dm = None if dropout_mask is None else torch.as_tensor(np.asarray(dropout_mask), dtype=self.dtype)
        return inception_logits(self.flags, tensors, x, training, dm, taps, rm)

    def predict(self, x, training=False):
        return self.predict_with_logits(x, training)[0]

    def predict_with_logits(self, x, training=False):
        with torch.no_grad():
            z, _ = self.logits(x, training)
        return torch.sigmoid(z).numpy(), z.numpy()

    def loss_and_grads(self, x, y, w, dropout_mask=None, relu_masks=None):
        """-> (loss, probs, {name: grad}, new_moving_stats) for one batch, training mode."""
        t = self._tensors(True)
        z, new_stats = self.logits(x, True, t, dropout_mask, relu_masks=relu_masks)
        yt = torch.as_tensor(np.asarray(y, np.float64).reshape(-1), dtype=self.dtype)
        wt = torch.as_tensor(np.asarray(w, np.float64).reshape(-1), dtype=self.dtype)
        loss, p = weighted_loss(z, yt, wt)
        names = [v.name for v in self.vars if v.trainable]
        grads = torch.autograd.grad(loss, [t[n] for n in names])
        self.last_logits = z.detach().numpy()
        return float(loss.detach()), p.detach().numpy(), dict(zip(names, grads)), new_stats

    def train_step(self, x, y, w, lr, dropout_mask=None, relu_masks=None):
        """One ``train_on_batch`` (train.py:295-299): forward(training) -> weighted BCE -> grads ->
        Keras Adam -> BN moving stats -> metric update.  Returns (loss, probs)."""
        loss, p, grads, new_stats = self.loss_and_grads(x, y, w, dropout_mask, relu_masks)
        tr = [v for v in self.vars if v.trainable]
        if self.adam is None:
            self.adam = KerasAdam([v.value.shape for v in tr], self.dtype)
        params = [torch.tensor(v.value, dtype=self.dtype) for v in tr]
        new = self.adam.apply(params, [grads[v.name] for v in tr], lr)
        for v, n in zip(tr, new):
            v.value = n.numpy().astype(np.float32)
        for v in self.vars:
            if v.name in new_stats:
                v.value = new_stats[v.name].numpy().astype(np.float32)
        self.metrics.update(p, y, self.last_logits)
        return loss, p
