"""Self-consistency tests that pin oracle/model_oracle.py (the reference ships no tests for the
model path — parity unpinned, see the oracle header): shapes / parameter counts from SURVEY
§A.2-A.3, finite differences, BN train/eval consistency, MixConv right alignment, Keras-Adam
and metric semantics."""
import math

import numpy as np
import pytest
import torch

from oracle import model_oracle as mo

DEF = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0")


def test_default_flags_do_not_construct():
    # mixednet.py:52-57 vs :298-305 — five residual entries against four blocks
    with pytest.raises(ValueError):
        mo.mixednet_build(mo.MIXEDNET_DEFAULTS, 194)


def test_shapes_and_param_counts():
    assert mo.mixednet_slices_dropped(DEF) == 46
    assert mo.spectrogram_length(1500, 10, 1, 46) == (148, 194)
    m = mo.OracleModel("mixednet", DEF, 194)
    assert m.n_params() == (22561, 22177)
    taps = {}
    x = np.random.default_rng(0).random((2, 194, 40), np.float32)
    m.logits(x, True, taps=taps)
    assert [taps["b%d.r0.pre_bn" % i].shape[1:] for i in range(4)] == [(188, 48), (180, 48), (168, 48), (148, 48)]
    assert taps["conv1"].shape[1:] == (192, 32)
    assert mo.inception_slices_dropped(mo.INCEPTION_DEFAULTS) == 28
    assert mo.OracleModel("inception", mo.INCEPTION_DEFAULTS, 176).n_params()[0] == 17901
    assert mo.OracleModel("inception", mo.INCEPTION_DEFAULTS, 194).n_params()[0] == 18189
    nb = dict(DEF, first_conv_kernel_size=5, stride=3, first_conv_filters=32, pointwise_filters="64,64,64,64",
              mixconv_kernel_sizes="[5],[7,11],[9,15],[23]")
    assert mo.mixednet_slices_dropped(nb) == 4 + 3 * (4 + 10 + 14 + 22)
    assert mo.spectrogram_length(1500, 10, 3, mo.mixednet_slices_dropped(nb))[1] == 204
    mnb = mo.OracleModel("mixednet", nb, 204)
    taps = {}
    mnb.logits(np.zeros((1, 204, 40), np.float32), False, taps=taps)
    assert [taps["b%d.r0.pre_bn" % i].shape[1] for i in range(4)] == [63, 53, 39, 17]
    assert mnb.n_params()[0] == 26049  # notebook cell 10 configuration


def small_model(kind="mixednet", seed=1):
    if kind == "mixednet":
        flags = dict(DEF, pointwise_filters="6,5", repeat_in_block="1,1", mixconv_kernel_sizes="[3],[3,5]",
                     residual_connection="0,1", first_conv_filters=4)
        T = 20
    else:
        flags = dict(mo.INCEPTION_DEFAULTS, cnn1_filters="8", cnn2_filters1="4", cnn2_filters2="4", cnn2_kernel_sizes="3",
                     cnn2_subspectral_groups="1", cnn2_dilation="1", dropout=0.0)
        T = 16
    m = mo.OracleModel(kind, flags, T, seed)
    rng = np.random.default_rng(seed)
    ws = [w + rng.normal(0, 0.05, w.shape).astype(np.float32) for w in m.get_weights()]
    for v, w in zip(m.vars, ws):
        if v.name.endswith("moving_variance"):
            w[:] = np.abs(w) + 0.5
    m.set_weights(ws)
    return m, T


@pytest.mark.parametrize("kind", ["mixednet", "inception"])
def test_finite_difference_gradients(kind):
    m, T = small_model(kind)
    rng = np.random.default_rng(5)
    x = rng.random((3, T, 40)) * 3
    y = np.array([1.0, 0.0, 1.0])
    w = np.array([1.0, 2.0, 0.5])
    loss, _, grads, _ = m.loss_and_grads(x, y, w)
    checked = 0
    for v in m.vars:
        if not v.trainable:
            continue
        g = grads[v.name].numpy()
        idx = tuple(rng.integers(0, s) for s in v.value.shape)
        base = v.value.copy()
        h = 1e-3
        vals = []
        for sgn in (+1, -1):
            v.value = base.copy().astype(np.float64)
            v.value[idx] += sgn * h
            t = {u.name: torch.tensor(u.value, dtype=torch.float64) for u in m.vars}
            z, _ = (mo.mixednet_logits if kind == "mixednet" else mo.inception_logits)(m.flags, t, torch.tensor(x), True)
            vals.append(float(mo.weighted_loss(z, torch.tensor(y), torch.tensor(w))[0]))
        v.value = base
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[idx]) <= 1e-5 + 1e-4 * abs(fd), (v.name, fd, g[idx])
        checked += 1
    assert checked >= 8


def test_bn_train_eval_consistency():
    m, T = small_model()
    x = np.random.default_rng(2).random((4, T, 40)) * 2
    z_train, stats = m.logits(x, True)
    # moving <- batch stats exactly (momentum 0): new = old*0.99 + batch*0.01  =>  batch = (new - .99 old)/.01
    for v in m.vars:
        if v.name in stats:
            v.value = ((stats[v.name].numpy() - 0.99 * v.value.astype(np.float64)) / 0.01)
    m.vars = [mo.Var(v.name, np.asarray(v.value, np.float64), v.trainable) for v in m.vars]
    t = {v.name: torch.tensor(v.value, dtype=torch.float64) for v in m.vars}
    z_eval, _ = mo.mixednet_logits(m.flags, t, torch.tensor(x), False)
    np.testing.assert_allclose(z_eval.numpy(), z_train.detach().numpy(), rtol=1e-7, atol=1e-8)


def test_mixconv_right_alignment():
    """Group g (kernel ks_g) output frame j = sum_i w_g[i] * x[j + (ks_last - ks_g) + i]  (SURVEY §A.2)."""
    m, T = small_model()
    taps = {}
    x = np.random.default_rng(3).random((1, T, 40))
    m.logits(x, False, taps=taps)
    w = {v.name: v.value.astype(np.float64) for v in m.vars}
    # input of block 1 = relu(bn(pw(dw(conv1)))) of block 0; recompute from taps
    pre = taps["b0.r0.pre_bn"].numpy()[0]
    a = (pre - w["b0.r0.bn.moving_mean"]) / np.sqrt(w["b0.r0.bn.moving_variance"] + 1e-3) * w["b0.r0.bn.gamma"] + w["b0.r0.bn.beta"]
    a = np.maximum(a, 0)  # [T0, 6]
    got = taps["b1.r0.dw"].numpy()[0]
    splits = mo.split_channels(6, 2)
    assert splits == [3, 3]
    c0 = 0
    for gi, (gc, k) in enumerate(zip(splits, (3, 5))):
        kw, kb = w["b1.r0.dw%d.kernel" % gi][:, 0, :, 0], w["b1.r0.dw%d.bias" % gi]
        for j in range(got.shape[0]):
            exp = sum(kw[i] * a[j + (5 - k) + i, c0:c0 + gc] for i in range(k)) + kb
            np.testing.assert_allclose(got[j, c0:c0 + gc], exp, rtol=1e-9, atol=1e-12)
        c0 += gc
    assert got.shape[0] == a.shape[0] - 4


def test_keras_adam_first_steps():
    ad = mo.KerasAdam([(1,)])
    p = [torch.tensor([1.0], dtype=torch.float64)]
    g = [torch.tensor([0.5], dtype=torch.float64)]
    p1 = ad.apply(p, g, 1e-3)
    # t=1: m=.05 v=.00025*... alpha = lr*sqrt(1-b2)/(1-b1); update = alpha*m/(sqrt(v)+eps)
    m, v = 0.05, 0.25 * 0.001
    alpha = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(float(p1[0]) - (1.0 - alpha * m / (math.sqrt(v) + 1e-7))) < 1e-15
    # epsilon placement differs from torch.optim.Adam: tiny gradients expose it
    ad2 = mo.KerasAdam([(1,)])
    q = ad2.apply(p, [torch.tensor([1e-9], dtype=torch.float64)], 1e-3)
    torch_style = 1.0 - 1e-3 * (1e-10 / 0.1) / (math.sqrt(1e-21 / 0.001) + 1e-7)
    assert abs(float(q[0]) - torch_style) > 1e-6


def test_loss_logits_form_is_sum_over_batch_size_and_unclipped():
    """Keras 3 + TF: the cached logits of the sigmoid output are used, so saturated wrong samples cost |z| (not
    -log(1e-7)) and keep their gradient."""
    z = torch.tensor([0.3, -1.2, 40.0, -40.0], dtype=torch.float64, requires_grad=True)
    y = torch.tensor([1.0, 0.0, 0.0, 1.0], dtype=torch.float64)
    w = torch.tensor([1.0, 3.0, 1.0, 1.0], dtype=torch.float64)
    assert mo.BCE_FROM_LOGITS
    loss, p = mo.weighted_loss(z, y, w)
    b = [math.log(1 + math.exp(-0.3)), math.log(1 + math.exp(-1.2)), 40.0, 40.0]
    assert abs(float(loss.detach()) - (b[0] + 3 * b[1] + b[2] + b[3]) / 4) < 1e-9
    (gz,) = torch.autograd.grad(loss, z)
    np.testing.assert_allclose(gz.numpy(), (w * (torch.sigmoid(z.detach()) - y) / 4).numpy(), atol=1e-15)
    assert abs(float(gz[2]) - 0.25) < 1e-12 and abs(float(gz[3]) + 0.25) < 1e-12   # no dead zone


def test_loss_probability_form_is_clipped(monkeypatch):
    monkeypatch.setattr(mo, "BCE_FROM_LOGITS", False)
    z = torch.tensor([0.3, -1.2, 40.0, -40.0], dtype=torch.float64)
    y = torch.tensor([1.0, 0.0, 0.0, 1.0], dtype=torch.float64)
    w = torch.tensor([1.0, 3.0, 1.0, 1.0], dtype=torch.float64)
    loss, p = mo.weighted_loss(z, y, w)
    # float32 bounds as in the reference's float32 graph: the upper clip is 1 - 1.19e-7, so a saturated wrong
    # positive costs -log(1.19e-7), a saturated wrong negative -log(1e-7)
    hi = float(np.float32(1.0) - np.float32(1e-7))
    b = [math.log(1 + math.exp(-0.3)), math.log(1 + math.exp(-1.2)), -math.log(1 - hi), -math.log(float(np.float32(1e-7)))]
    assert abs(float(loss) - (b[0] + 3 * b[1] + b[2] + b[3]) / 4) < 1e-6


def test_metrics_match_direct_threshold_compare():
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.random(500).astype(np.float32), np.float32([0.0, 1.0, 0.5, 0.25, 0.75, 0.01])])
    y = (rng.random(p.size) < 0.4).astype(np.float64)
    mt = mo.Metrics()
    mt.update(p[:200], y[:200])
    mt.update(p[200:], y[200:])
    r = mt.result()
    th = np.linspace(0.0, 1.0, 101)
    pos = y > 0.5
    # strict '>' against thresholds; Keras buckets in fp32, which can move a value sitting within one
    # ulp of a threshold — none of the random values here do
    tp = np.array([np.sum((p > np.float32(t)) & pos) for t in th])
    fp = np.array([np.sum((p > np.float32(t)) & ~pos) for t in th])
    np.testing.assert_array_equal(r["tp"], tp)
    np.testing.assert_array_equal(r["fp"], fp)
    np.testing.assert_array_equal(r["fn"], pos.sum() - tp)
    np.testing.assert_array_equal(r["tn"], (~pos).sum() - fp)
    assert abs(r["accuracy"] - np.mean((p > 0.5) == pos)) < 1e-12
    assert abs(r["recall"] - tp[50] / pos.sum()) < 1e-12
    assert 0.0 <= r["auc"] <= 1.0
    # perfect separation -> AUC 1
    m2 = mo.Metrics()
    m2.update(np.float32([0.9, 0.8, 0.1, 0.2]), [1, 1, 0, 0])
    assert abs(m2.result()["auc"] - 1.0) < 1e-9


def test_train_step_reduces_loss():
    m = mo.OracleModel("mixednet", DEF, 194, dtype=torch.float32)
    rng = np.random.default_rng(0)
    x = rng.random((8, 194, 40), np.float32) * 26
    y = np.array([1, 0, 1, 0, 1, 0, 1, 0], np.float64)
    x[y > 0.5, 50:60, :] += 10.0
    losses = [m.train_step(x, y, np.ones(8), 1e-3)[0] for _ in range(6)]
    assert losses[-1] < losses[0]


def test_mixednet_heads_restatement_against_a_plain_numpy_forward():
    """SpatialAttention / pooled heads (mixednet.py:234-275,362-381): the torch graph of the oracle against an
    independent loop-level numpy forward on a tiny case; parameter bookkeeping of the extra attention kernel."""
    flags = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0", pointwise_filters="8,8", repeat_in_block="1,1",
                 mixconv_kernel_sizes="[3],[3]", first_conv_filters=8, spatial_attention=1, pooled=1, max_pool=1)
    T = 20
    om = mo.OracleModel("mixednet", flags, T, seed=5)
    names = [v.name for v in om.vars]
    assert names[-3:] == ["attention.kernel", "dense.kernel", "dense.bias"]
    assert om.vars[-3].value.shape == (4, 1, 2, 1) and om.vars[-2].value.shape == (8, 1)   # pooled: one frame left
    rng = np.random.default_rng(0)
    x = rng.random((2, T, 40)) * 5
    taps = {}
    z, _ = om.logits(x, False, taps=taps)
    # numpy: last block output a [B,T',C] -> attention -> max pool -> dense
    t = {v.name: v.value.astype(np.float64) for v in om.vars}
    pre = taps["b1.r0.pre_bn"].detach().numpy()
    a = np.maximum((pre - t["b1.r0.bn.moving_mean"]) / np.sqrt(t["b1.r0.bn.moving_variance"] + 1e-3) * t["b1.r0.bn.gamma"]
                   + t["b1.r0.bn.beta"], 0.0)
    B, Ta, C = a.shape
    wa = t["attention.kernel"][:, 0, :, 0]                       # [4][2]
    out = np.zeros((B, Ta - 3, C))
    for b in range(B):
        avg, mx = a[b].mean(axis=1), a[b].max(axis=1)
        for tt in range(Ta - 3):
            pre_s = sum(wa[j, 0] * avg[tt + j] + wa[j, 1] * mx[tt + j] for j in range(4))
            out[b, tt] = a[b, tt + 3] * (1.0 / (1.0 + np.exp(-pre_s)))
    zz = out.max(axis=1) @ t["dense.kernel"][:, 0] + t["dense.bias"][0]
    np.testing.assert_allclose(z.detach().numpy(), zz, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(taps["attention.out"].detach().numpy(), out, rtol=1e-10, atol=1e-12)


def test_bf16_pointwise_function_gradients_use_rounded_operands():
    x = torch.tensor(np.random.default_rng(1).normal(size=(2, 5, 7)), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(np.random.default_rng(2).normal(size=(3, 5)), dtype=torch.float64, requires_grad=True)
    y = mo._Bf16Pointwise.apply(x, w)
    xr, wr = mo._round_bf16(x.detach()), mo._round_bf16(w.detach())
    np.testing.assert_allclose(y.detach().numpy(), torch.einsum("oc,bct->bot", wr, xr).numpy(), rtol=0, atol=0)
    gy = torch.tensor(np.random.default_rng(3).normal(size=y.shape), dtype=torch.float64)
    gx, gw = torch.autograd.grad(y, [x, w], gy)
    gr = mo._round_bf16(gy)
    np.testing.assert_allclose(gx.numpy(), torch.einsum("oc,bot->bct", wr, gr).numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(gw.numpy(), torch.einsum("bot,bct->oc", gr, xr).numpy(), rtol=0, atol=0)
    assert float((xr - x.detach()).abs().max()) > 0          # the rounding is real


def test_oracle_outputs_frozen():
    """Regression guard: the restatement still produces the numbers frozen in tests/golden/model_oracle_golden.npz
    (made by tests/golden/make_golden_model.py from this same oracle — not a reference pin, see its docstring)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_model", os.path.join(here, "golden", "make_golden_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(os.path.join(here, "golden", "model_oracle_golden.npz"))
    for name, (kind, flags, T) in mod.CASES.items():
        got = mod.run(kind, flags, T)
        for k, v in got.items():
            np.testing.assert_allclose(v, gold["%s/%s" % (name, k)], rtol=1e-9, atol=1e-12, err_msg="%s/%s" % (name, k))
