"""TEST INFRASTRUCTURE ONLY — a SECOND, independent CPU restatement of the model half of the hot path:
plain numpy float64, hand-derived backward passes, no autograd and no code shared with
``oracle/model_oracle.py`` (the torch restatement).  Purpose: the reference ships no tests, golden
vectors or checkpoints for the model (``find /root/reference -name '*.h5' -o -name '*.tflite' -o -name '*.npz'``
is empty) and TensorFlow cannot be installed here, so the torch oracle cannot be pinned against the
reference itself; two restatements written separately from the reference source that agree to 1e-9 on loss,
every gradient, the Adam step and the BN moving statistics are the strongest pin available
(``tests/test_model_oracle_np.py``).  Since round 6 the GRAPH of both is also checked against the reference's own
builders executed over stand-in Keras primitives (``oracle/ref_model_shim.py``, ``tests/test_reference_graph.py``, fixture
``tests/golden/ref_graph_golden.npz``); the primitives' arithmetic, loss, optimizer and metrics remain "unpinned against TF".

Written from (reference file:line):
  * microwakeword/mixednet.py:278-386  graph of ``model()``: expand_dims -> Stream(Conv2D(first_conv_filters,
    (k1,1), strides=(stride,1), valid, no bias)) -> ReLU -> per block [MixConv (:197-231: one DepthwiseConv2D((ks,1),
    valid) with bias, or ChannelSplit (:132-136 sizes, remainder to group 0) + StridedKeep(ks) (identity outside the
    streaming modes, strided_drop.py:79-83) + per-group DepthwiseConv2D + StridedDrop of the leading frames down to
    the LAST group's length + concat) -> Conv2D(filters, 1, no bias) -> BatchNormalization -> ReLU] ->
    Stream(Identity) -> Flatten -> Dense(1, sigmoid)
  * microwakeword/inception.py:46-141,232-340  stem Stream(Conv2D(f,(k,1),valid,no bias)) -> SubSpectralNormalization(g)
    -> ReLU; per block: three branches of conv2d_bn / conv2d_bn_delay (padding "None" => plain valid convolution, no
    Delay layer), StridedDrop of the leading frames of the shorter-receptive-field branches, channel concat,
    conv2d_bn(1x1); Stream(Flatten) -> Dropout -> Dense(1, sigmoid)
  * microwakeword/layers/sub_spectral_normalization.py:24-67  reshape [T,1,C] -> [T,C/g,g], BatchNormalization over the
    last axis (g slots; channel c uses slot c mod g), reshape back
  * microwakeword/layers/strided_drop.py:21-58  ``inputs[:, drop:, :, :]``
  * microwakeword/train.py:206-207,288-299  BinaryCrossentropy(from_logits=False) on the sigmoid output (Keras 3 + TF:
    evaluated from the cached logits), sample weights, Adam() defaults
Keras-3 layer semantics (not in /root/reference; from the Keras documentation): BatchNormalization momentum 0.99,
epsilon 1e-3, biased batch variance for both the normalisation and the moving average; loss reduction
``sum_over_batch_size``; Adam with epsilon added to the un-corrected sqrt(v).
"""
import math

import numpy as np

BN_EPS, BN_MOM = 1e-3, 0.99


def _ints(text):
    return [int(t) for t in str(text).replace("[", "").replace("]", "").split(",") if t.strip()]


def _groups(text):
    """'[5], [7,11]' -> [[5], [7, 11]]"""
    out, cur, depth, tok = [], [], 0, ""
    for ch in str(text):
        if ch == "[":
            depth, cur, tok = 1, [], ""
        elif ch == "]":
            if tok.strip():
                cur.append(int(tok))
            out.append(cur)
            depth, tok = 0, ""
        elif ch == "," and depth:
            if tok.strip():
                cur.append(int(tok))
            tok = ""
        elif depth:
            tok += ch
    return out


# ---------------------------------------------------------------------------------------------- graph nodes
class Node:
    """value = f(inputs); backward(dvalue) returns the gradient for each input and fills self.grads."""

    def __init__(self, *inputs):
        self.inputs = list(inputs)
        self.value = None
        self.dvalue = None
        self.grads = {}

    def params(self):
        return []


class Input(Node):
    def forward(self, training):
        pass

    def backward(self, dy):
        return []


class ConvTime(Node):
    """Conv2D with a (K,1) kernel on [B,T,1,Cin], 'valid' in time, no bias: y[b,t,:] = sum_k x[b,t*s+k*d,:] @ W[k]."""

    def __init__(self, x, name, stride=1, dilation=1):
        super().__init__(x)
        self.name, self.stride, self.dil = name, stride, dilation

    def params(self):
        return [self.name + ".kernel"]

    def forward(self, training):
        x = self.inputs[0].value
        W = self.w[self.name + ".kernel"].reshape(self.w[self.name + ".kernel"].shape[0], self.w[self.name + ".kernel"].shape[2], -1)
        K = W.shape[0]
        Tout = (x.shape[1] - (K - 1) * self.dil - 1) // self.stride + 1
        y = np.zeros((x.shape[0], Tout, W.shape[2]))
        for k in range(K):
            xs = x[:, k * self.dil: k * self.dil + (Tout - 1) * self.stride + 1: self.stride, :]
            y += xs @ W[k]
        self.value = y

    def backward(self, dy):
        x = self.inputs[0].value
        Wfull = self.w[self.name + ".kernel"]
        W = Wfull.reshape(Wfull.shape[0], Wfull.shape[2], -1)
        K, Tout = W.shape[0], dy.shape[1]
        dW = np.zeros_like(W)
        dx = np.zeros_like(x)
        for k in range(K):
            sl = slice(k * self.dil, k * self.dil + (Tout - 1) * self.stride + 1, self.stride)
            dW[k] = np.einsum("bti,bto->io", x[:, sl, :], dy)
            dx[:, sl, :] += dy @ W[k].T
        self.grads[self.name + ".kernel"] = dW.reshape(Wfull.shape)
        return [dx]


class DepthwiseTime(Node):
    """DepthwiseConv2D((K,1), valid) with bias: y[b,t,c] = sum_k x[b,t+k,c] w[k,c] + bias[c]."""

    def __init__(self, x, name):
        super().__init__(x)
        self.name = name

    def params(self):
        return [self.name + ".kernel", self.name + ".bias"]

    def forward(self, training):
        x = self.inputs[0].value
        w = self.w[self.name + ".kernel"][:, 0, :, 0]
        K = w.shape[0]
        Tout = x.shape[1] - K + 1
        y = np.zeros((x.shape[0], Tout, x.shape[2]))
        for k in range(K):
            y += x[:, k:k + Tout, :] * w[k]
        self.value = y + self.w[self.name + ".bias"]

    def backward(self, dy):
        x = self.inputs[0].value
        wfull = self.w[self.name + ".kernel"]
        w = wfull[:, 0, :, 0]
        K, Tout = w.shape[0], dy.shape[1]
        dw = np.zeros_like(w)
        dx = np.zeros_like(x)
        for k in range(K):
            dw[k] = np.sum(x[:, k:k + Tout, :] * dy, axis=(0, 1))
            dx[:, k:k + Tout, :] += dy * w[k]
        self.grads[self.name + ".kernel"] = dw.reshape(wfull.shape)
        self.grads[self.name + ".bias"] = dy.sum(axis=(0, 1))
        return [dx]


class BatchNormSlots(Node):
    """SubSpectralNormalization(g): g = 1 is BatchNormalization over the C channels; g > 1 reshapes [T,1,C] to
    [T,C/g,g] and normalises over that last axis, i.e. g parameter slots, channel c in slot c mod g."""

    def __init__(self, x, name, groups=1):
        super().__init__(x)
        self.name, self.g = name, groups
        self.new_stats = {}

    def params(self):
        return [self.name + ".gamma", self.name + ".beta"]

    def forward(self, training):
        x = self.inputs[0].value
        B, T, C = x.shape
        self.slots = C if self.g == 1 else self.g
        xr = x.reshape(B, T, C // self.slots, self.slots)
        gam, bet = self.w[self.name + ".gamma"], self.w[self.name + ".beta"]
        if training:
            mean = xr.mean(axis=(0, 1, 2))
            var = ((xr - mean) ** 2).mean(axis=(0, 1, 2))       # biased
            self.new_stats = {self.name + ".moving_mean": self.w[self.name + ".moving_mean"] * BN_MOM + mean * (1 - BN_MOM),
                              self.name + ".moving_variance": self.w[self.name + ".moving_variance"] * BN_MOM + var * (1 - BN_MOM)}
        else:
            mean, var = self.w[self.name + ".moving_mean"], self.w[self.name + ".moving_variance"]
        self.rstd = 1.0 / np.sqrt(var + BN_EPS)
        self.xhat = (xr - mean) * self.rstd
        self.training = training
        self.value = (self.xhat * gam + bet).reshape(B, T, C)

    def backward(self, dy):
        B, T, C = dy.shape
        dyr = dy.reshape(B, T, C // self.slots, self.slots)
        gam = self.w[self.name + ".gamma"]
        self.grads[self.name + ".gamma"] = np.sum(dyr * self.xhat, axis=(0, 1, 2))
        self.grads[self.name + ".beta"] = np.sum(dyr, axis=(0, 1, 2))
        dxh = dyr * gam
        if self.training:
            n = B * T * (C // self.slots)
            dx = self.rstd * (dxh - dxh.sum(axis=(0, 1, 2)) / n - self.xhat * np.sum(dxh * self.xhat, axis=(0, 1, 2)) / n)
        else:
            dx = dxh * self.rstd
        return [dx.reshape(B, T, C)]


class Relu(Node):
    def forward(self, training):
        self.value = np.maximum(self.inputs[0].value, 0.0)

    def backward(self, dy):
        return [dy * (self.inputs[0].value > 0)]


class DropFrames(Node):
    """StridedDrop / StridedKeep: x[:, n:, :]."""

    def __init__(self, x, n):
        super().__init__(x)
        self.n = n

    def forward(self, training):
        self.value = self.inputs[0].value[:, self.n:, :]

    def backward(self, dy):
        dx = np.zeros_like(self.inputs[0].value)
        dx[:, self.n:, :] = dy
        return [dx]


class Channels(Node):
    """ChannelSplit piece: x[:, :, c0:c1]."""

    def __init__(self, x, c0, c1):
        super().__init__(x)
        self.c0, self.c1 = c0, c1

    def forward(self, training):
        self.value = self.inputs[0].value[:, :, self.c0:self.c1]

    def backward(self, dy):
        dx = np.zeros_like(self.inputs[0].value)
        dx[:, :, self.c0:self.c1] = dy
        return [dx]


class Concat(Node):
    def forward(self, training):
        self.value = np.concatenate([i.value for i in self.inputs], axis=2)

    def backward(self, dy):
        out, o = [], 0
        for i in self.inputs:
            c = i.value.shape[2]
            out.append(dy[:, :, o:o + c])
            o += c
        return out


class FlattenDense(Node):
    """Flatten -> [Dropout with an explicit keep mask, inverted scaling] -> Dense(1): the logits."""

    def __init__(self, x, rate=0.0):
        super().__init__(x)
        self.rate = rate
        self.keep = None

    def params(self):
        return ["dense.kernel", "dense.bias"]

    def forward(self, training):
        x = self.inputs[0].value
        flat = x.reshape(x.shape[0], -1)
        if training and self.rate > 0:
            if self.keep is None:
                raise ValueError("training-mode dropout needs an explicit keep mask")
            flat = flat * self.keep / (1.0 - self.rate)
        self.flat = flat
        self.value = flat @ self.w["dense.kernel"][:, 0] + self.w["dense.bias"][0]

    def backward(self, dz):
        x = self.inputs[0].value
        self.grads["dense.kernel"] = (self.flat.T @ dz)[:, None]
        self.grads["dense.bias"] = np.array([dz.sum()])
        dflat = dz[:, None] * self.w["dense.kernel"][:, 0][None, :]
        if self.training and self.rate > 0:
            dflat = dflat * self.keep / (1.0 - self.rate)
        return [dflat.reshape(x.shape)]


# ---------------------------------------------------------------------------------------------- models
class NumpyModel:
    """kind 'mixednet' | 'inception'; weights by name (the labels of the torch oracle's variables, nothing else is shared)."""

    def __init__(self, kind, flags, frames):
        self.kind, self.flags, self.frames = kind, dict(flags), frames
        self.nodes = []
        self.inp = self._add(Input())
        self.head = self._build_mixednet() if kind == "mixednet" else self._build_inception()
        self.w = {}
        self.m, self.v, self.t = {}, {}, 0

    def _add(self, node):
        self.nodes.append(node)
        return node

    # mixednet.py:278-386
    def _build_mixednet(self):
        f = self.flags
        if any(_ints(f.get("residual_connection", "0"))) or f.get("spatial_attention") or f.get("pooled") or any(r != 1 for r in _ints(f["repeat_in_block"])):
            raise NotImplementedError("the independent oracle covers the sequential MixedNet topologies only")
        net = self.inp
        if int(f["first_conv_filters"]) > 0:
            net = self._add(ConvTime(net, "conv1", stride=int(f["stride"])))
            net = self._add(Relu(net))
        filters, kernels = _ints(f["pointwise_filters"]), _groups(f["mixconv_kernel_sizes"])
        cin = int(f["first_conv_filters"]) or 40
        for b, (fo, ks) in enumerate(zip(filters, kernels)):
            if max(ks) > 1:
                if len(ks) == 1:
                    net = self._add(DepthwiseTime(net, "b%d.r0.dw0" % b))
                else:
                    n = len(ks)
                    sizes = [cin // n] * n
                    sizes[0] += cin - sum(sizes)
                    outs, c0 = [], 0
                    for gi, (sz, k) in enumerate(zip(sizes, ks)):
                        piece = self._add(Channels(net, c0, c0 + sz))
                        c0 += sz
                        # StridedKeep(k) is a no-op when training (strided_drop.py: NON_STREAM keeps everything); the
                        # alignment happens after the convolution: drop the leading frames the longest kernel lacks
                        conv = self._add(DepthwiseTime(piece, "b%d.r0.dw%d" % (b, gi)))
                        outs.append((conv, k))
                    klast = ks[-1]
                    net = self._add(Concat(*[self._add(DropFrames(cv, klast - k)) if klast - k else cv for cv, k in outs]))
            net = self._add(ConvTime(net, "b%d.r0.pw" % b))
            net = self._add(BatchNormSlots(net, "b%d.r0.bn" % b))
            net = self._add(Relu(net))
            cin = fo
        return self._add(FlattenDense(net))

    # inception.py:232-340
    def _build_inception(self):
        f = self.flags
        net = self.inp
        for i, (fl, k, g) in enumerate(zip(_ints(f["cnn1_filters"]), _ints(f["cnn1_kernel_sizes"]), _ints(f["cnn1_subspectral_groups"]))):
            net = self._add(ConvTime(net, "stem%d" % i))
            net = self._add(BatchNormSlots(net, "stem%d.bn" % i, g))
            net = self._add(Relu(net))

        def cbr(x, name, g, dil=1):
            c = self._add(ConvTime(x, name, dilation=dil))
            return self._add(Relu(self._add(BatchNormSlots(c, name + ".bn", g))))

        for i, (f1, f2, k, g, d) in enumerate(zip(_ints(f["cnn2_filters1"]), _ints(f["cnn2_filters2"]), _ints(f["cnn2_kernel_sizes"]),
                                                   _ints(f["cnn2_subspectral_groups"]), _ints(f["cnn2_dilation"]))):
            span = d * (k - 1)
            b1 = cbr(net, "i%d.b1" % i, g)
            b2 = cbr(cbr(net, "i%d.b2a" % i, g), "i%d.b2b" % i, g, d)
            b3 = cbr(cbr(cbr(net, "i%d.b3a" % i, g), "i%d.b3b" % i, g, d), "i%d.b3c" % i, g, d)
            b1 = self._add(DropFrames(b1, 2 * span))
            b2 = self._add(DropFrames(b2, span))
            net = cbr(self._add(Concat(b1, b2, b3)), "i%d.red" % i, 1)
        return self._add(FlattenDense(net, float(f.get("dropout", 0.0))))

    # ---- weights
    def set_weights(self, by_name):
        self.w = {k: np.asarray(v, np.float64).copy() for k, v in by_name.items()}
        for n in self.nodes:
            n.w = self.w

    def trainable_names(self):
        return [p for n in self.nodes for p in n.params()]

    # ---- passes
    def logits(self, x, training, keep=None):
        self.inp.value = np.asarray(x, np.float64)
        self.head.keep = None if keep is None else np.asarray(keep, np.float64)
        for n in self.nodes:
            n.training = training
            n.forward(training)
        return self.head.value

    def loss_and_grads(self, x, y, w, keep=None):
        """Weighted BCE in the logits form (train.py:206 under Keras 3 + TF), reduction sum_over_batch_size;
        returns (loss, probs, {name: grad}, {name: new moving statistic})."""
        z = self.logits(x, True, keep)
        y = np.asarray(y, np.float64).reshape(-1)
        w = np.asarray(w, np.float64).reshape(-1)
        B = z.shape[0]
        bce = np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z)))
        p = 1.0 / (1.0 + np.exp(-z))
        loss = float(np.sum(bce * w) / B)
        # reverse sweep with gradient accumulation for values that have several consumers
        for n in self.nodes:
            n.dvalue = None
        self.head.dvalue = w * (p - y) / B
        for n in reversed(self.nodes):
            if n.dvalue is None:
                continue
            for src, g in zip(n.inputs, n.backward(n.dvalue)):
                src.dvalue = g if src.dvalue is None else src.dvalue + g
        grads, stats = {}, {}
        for n in self.nodes:
            grads.update(n.grads)
            if isinstance(n, BatchNormSlots):
                stats.update(n.new_stats)
        return loss, p, grads, stats

    def train_step(self, x, y, w, lr, keep=None, beta1=0.9, beta2=0.999, eps=1e-7):
        """One Keras ``train_on_batch``: gradients, Adam (epsilon outside the bias correction), moving statistics."""
        loss, p, grads, stats = self.loss_and_grads(x, y, w, keep)
        self.t += 1
        alpha = lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)
        for name, g in grads.items():
            m = self.m.get(name, np.zeros_like(g))
            v = self.v.get(name, np.zeros_like(g))
            m = m + (g - m) * (1.0 - beta1)
            v = v + (g * g - v) * (1.0 - beta2)
            self.m[name], self.v[name] = m, v
            self.w[name] = self.w[name] - alpha * m / (np.sqrt(v) + eps)
        for name, s in stats.items():
            self.w[name] = s
        for n in self.nodes:
            n.w = self.w
        return loss, p
