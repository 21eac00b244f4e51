"""Parity checks shared by the GPU tests (libmww_hip.so on an MI355X, ``-m gpu``) and the CPU
debug run of the same kernel sources under tests/hipemu.  Every check compares the native engine
(through the C ABI) with oracle/ on identical seeded inputs."""
import os
import random
import sys

import numpy as np
import torch

from microwakeword_amd import native
from microwakeword_amd.data import FeatureHandler
from microwakeword_amd.layout import MixedNetLayout
from oracle import data_oracle as do
from oracle import model_oracle as mo

DEF = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0")
POLICY = dict(freq_mix_prob=0.0, time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)
SCALE = np.float32(0.0390625)
FWD_TOL = 1e-3  # north_star: forward parity within 1e-3 on fp32 outputs


def perturbed_oracle(T, seed=42, flags=DEF, dtype=torch.float64):
    om = mo.OracleModel("mixednet", flags, T, seed=seed, dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    ws = []
    for v, w in zip(om.vars, om.get_weights()):
        if v.name.endswith(("bias", "beta", "moving_mean")):
            w = w + rng.normal(0, 0.1, w.shape).astype(np.float32)
        if v.name.endswith(("gamma", "moving_variance")):
            w = w + np.abs(rng.normal(0, 0.2, w.shape)).astype(np.float32)
        ws.append(w)
    om.set_weights(ws)
    return om


def make_engine(lib, T, max_batch, om=None, flags=DEF):
    lay = MixedNetLayout(flags, T)
    eng = native.Engine(lib=lib, **lay.engine_args(max_batch))
    eng.set_grad_mask(lay.grad_mask())
    if flags.get("pw_bf16"):
        eng.set_option("pointwise_bf16", 1)
    if flags.get("st_bf16"):
        eng.set_option("storage_bf16", 1)
    if flags.get("bwd_wide") is not None:   # threads per workgroup of the block backward kernels (kernels_bwdw.hip.h)
        eng.set_option("bwd_wide", flags["bwd_wide"])
    if flags.get("conv1_x6") is not None:   # conv1 weight gradient (backward kernel) as bf16 slice products (common.hip.h) or exact-fp32 MFMA
        eng.set_option("conv1_x6", flags["conv1_x6"])
    if flags.get("bwd_first_wide") is not None:   # stride-1 first block with conv1_x6: the 512-thread form of its backward kernel
        eng.set_option("bwd_first_wide", flags["bwd_first_wide"])
    if flags.get("conv1_x6_fwd") is not None:   # ... the first convolution of the forward kernel (off by default: measured slower)
        eng.set_option("conv1_x6_fwd", flags["conv1_x6_fwd"])
    if om is not None:
        p, s = lay.pack(om.get_weights())
        eng.set_params(p)
        eng.set_bn_state(s)
    return lay, eng


def check_conv1_x6_against_the_f32_form(lib, B=6, T=194, flags=DEF, raw_u16_range=False, fwd_too=True):
    """Options "conv1_x6" (the conv1 weight gradient; on by default) and "conv1_x6_fwd" (the first convolution; off by default,
    ``fwd_too`` switches both): six bf16 slice products per fp32 product on v_mfma_f32_16x16x32_bf16 instead of the exact-fp32 MFMA.  Both forms on the same batch: probabilities within 2e-6 (the split
    drops terms below 2^-24 of each product), each form's gradients against the float64 oracle with its own ReLU decisions
    imposed (check_train_steps: L2 <= 1e-4 per tensor), and the two forms' gradients against each other: within 5e-6 of each
    tensor unless a ReLU decision differs between them - last-bit differences of a pre-activation within rounding of zero flip
    relu'(0+-), which moves the gradient by that ONE element's share, ~1 / sqrt(elements) = 1e-3 relative (seen at B = 48,
    T = 194: every tensor upstream of the flipped element off by 2e-3, the dense kernel - whose gradient multiplies the
    activation, ~0 there - by 1e-6) - then within 2e-2.  And NOT bit-identical everywhere (then the option is not wired)."""
    rng = np.random.default_rng(11)
    if raw_u16_range:   # every uint16 value the micro-frontend store could hold, not just 0..666
        x = (rng.integers(0, 65536, size=(B, T, 40)).astype(np.float32) * SCALE).astype(np.float32)
    else:
        x = synth_x(rng, B, T)
    y = (rng.random(B) < 0.5).astype(np.float32)
    outs = {}
    for form in (0, 1):
        om = perturbed_oracle(T, flags=flags) if flags is not DEF else perturbed_oracle(T)
        lay, eng = make_engine(lib, T, B, om, flags=dict(flags, conv1_x6=form, conv1_x6_fwd=form if fwd_too else 0))
        eng.set_batch(x)
        eng.set_targets(y, np.ones(B, np.float32))
        eng.train_step(B, 1e-3)
        outs[form] = (eng.read_outputs(B)[0].copy(), eng.get_grads().copy(), lay)
        eng.close()
    p0, g0, lay = outs[0]
    p1, g1, _ = outs[1]
    assert np.abs(p0 - p1).max() <= 2e-6, np.abs(p0 - p1).max()
    worst, detail = 0.0, []
    off = 0
    floor = 2e-2 * float(np.linalg.norm(g0))   # tensors whose gradient is analytically zero (depthwise biases in front of a BN) hold the rounding noise of long cancelling sums
    for name, n in lay.segments():
        a, b = g0[off:off + n], g1[off:off + n]
        off += n
        e = float(np.linalg.norm(a - b) / max(np.linalg.norm(a), floor))
        worst = max(worst, e)
        detail.append("%s %.2e" % (name, e))
    assert worst <= 2e-2, (worst, detail)
    assert not np.array_equal(g0, g1), "conv1_x6 0 and 1 gave bit-identical gradients: the option is not wired"
    return worst


def synth_x(rng, B, T):
    return (rng.integers(0, 667, size=(B, T, 40)).astype(np.float32) * SCALE).astype(np.float32)


def oracle_grads_native_order(lay, om, grads):
    arrs = []
    for name, shape, kind in lay.keras_vars:
        on = name
        for b in range(len(lay.blocks)):
            on = on.replace("b%d." % b, "b%d.r0." % b)
        arrs.append(grads[on].numpy().astype(np.float32) if kind == "param" else np.zeros(shape, np.float32))
    return lay.pack(arrs)[0]


# ------------------------------------------------------------------------------------------ data
def golden_stores(gold, tag):
    def grab(prov, mode):
        out, i = [], 0
        while "%s/in/%s/%s/%d" % (tag, prov, mode, i) in gold:
            out.append(gold["%s/in/%s/%s/%d" % (tag, prov, mode, i)])
            i += 1
        return [out] if out else []

    return {p: {m: grab(p, m) for m in do.MODES} for p in ("pos", "neg", "cut")}


def golden_config(gold, tag):
    st = golden_stores(gold, tag)
    return {"stride": 1, "window_step_ms": 10, "features": [
        dict(type="mmap", stores=st["pos"], truth=True, sampling_weight=2.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
        dict(type="mmap", stores=st["neg"], truth=False, sampling_weight=10.0, penalty_weight=1.5, truncation_strategy="random"),
        dict(type="mmap", stores=st["cut"], truth=False, sampling_weight=3.0, penalty_weight=0.5, truncation_strategy="fixed_right_cutoff", fixed_right_cutoffs=[0, 5, 11]),
    ]}


def check_get_data_against_reference_golden(lib, gold, tag):
    """FeatureHandler (native sampler + HIP assemble) reproduces the reference's own outputs bit for bit."""
    T = 194
    _, eng = make_engine(lib, T, 16)
    random.seed(3)
    np.random.seed(3)
    fh = FeatureHandler(golden_config(gold, tag), engine=eng)
    for call in range(2):
        x, y, w = fh.get_data("training", 16, T, "default", POLICY)
        assert x.dtype == np.float32 and x.shape == (16, T, 40)
        np.testing.assert_array_equal(x, gold["%s/train%d/xc" % (tag, call)].astype(np.float32) * SCALE)
        np.testing.assert_array_equal(y, gold["%s/train%d/y" % (tag, call)])
        np.testing.assert_array_equal(w, gold["%s/train%d/w" % (tag, call)])
    x, y, w = fh.get_data("validation", 16, T, "truncate_start")
    np.testing.assert_array_equal(x, gold[tag + "/val/xc"].astype(np.float32) * SCALE)
    np.testing.assert_array_equal(y, gold[tag + "/val/y"])
    np.testing.assert_array_equal(w, gold[tag + "/val/w"])
    x, y, w = fh.get_data("validation_ambient", 16, T, "split")
    np.testing.assert_array_equal(x, gold[tag + "/amb/xc"].astype(np.float32) * SCALE)
    np.testing.assert_array_equal(y, gold[tag + "/amb/y"])
    assert [fh.get_mode_size(m) for m in ("training", "validation", "validation_ambient")] == list(gold[tag + "/sizes"])
    np.testing.assert_array_equal([fh.get_mode_duration(m) for m in ("training", "validation", "validation_ambient")], gold[tag + "/durations"])
    eng.close()


def check_sampler_matches_oracle_descriptors(lib, B=64, n_samples=48, seed=0):
    """Mask indices / windows bit-exact against the oracle's descriptor draw, and the RNG streams
    are left exactly where the reference would leave them."""
    T = 194
    pos, neg = do.synthetic_stores(n_samples, 1234)
    _, eng = make_engine(lib, T, B)
    cfg = {"stride": 1, "window_step_ms": 10, "features": [
        dict(type="mmap", stores={"training": [pos]}, truth=True, sampling_weight=2.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
        dict(type="mmap", stores={"training": [neg]}, truth=False, sampling_weight=10.0, penalty_weight=1.0, truncation_strategy="random")]}
    random.seed(seed)
    np.random.seed(seed)
    fh = FeatureHandler(cfg, engine=eng)
    b1 = fh.draw_training_batch(B, T, "default", POLICY)
    b2 = fh.draw_training_batch(B, T, "default", POLICY)
    tail_native = (random.random(), np.random.random())
    random.seed(seed)
    np.random.seed(seed)
    provs = [do.index_provider({"training": [pos]}, True, 2.0, 1.0, "truncate_start", 1, 0.01),
             do.index_provider({"training": [neg]}, False, 10.0, 1.0, "random", 1, 0.01)]
    for b in (b1, b2):
        x, y, w, descs, order = do.get_data(provs, "training", B, T, "default", POLICY)
        np.testing.assert_array_equal(b["order"], order)
        for j, d in enumerate(descs):
            assert b["provider"][j] == d.provider
            dw = b["draw_windows"][j]
            assert (dw["pad_rows"], dw["copy_rows"]) == (d.pad_rows, d.copy_rows)
            exp_masks = [list(m) for m in d.time_masks + d.freq_masks]
            assert b["draw_masks"][j].tolist() == exp_masks
        np.testing.assert_array_equal(b["labels"], y)
        np.testing.assert_array_equal(b["weights"], w)
        eng.assemble(b["windows"], b["masks"], b["n_time"], b["n_freq"])
        np.testing.assert_array_equal(eng.get_batch(B), x)
    assert tail_native == (random.random(), np.random.random())
    eng.close()


# ------------------------------------------------------------------------------------------ model
# the topology of the reference's training notebook (cell 10): first conv 5x1 stride 3, 64 pointwise
# filters, multi-kernel MixConv groups; spectrogram_length 204
# BASELINE configs[4]: 1x1 contractions with bf16 operands (the oracle rounds the same operands)
BF16 = dict(DEF, pw_bf16=True)
# ... and with the block outputs p_k / stashed gradients g_k held in HBM as bf16 on top of it ("storage_bf16")
BF16_STORED = dict(DEF, st_bf16=True)


def _lowp(flags):
    return bool(flags.get("pw_bf16") or flags.get("st_bf16"))
NOTEBOOK = dict(DEF, first_conv_kernel_size=5, stride=3, first_conv_filters=32, pointwise_filters="64,64,64,64",
                mixconv_kernel_sizes="[5],[7,11],[9,15],[23]")
# crosses of the two documented topologies that the specialised block kernels also instantiate (either width with either
# kernel set, either first conv)
CROSSED = (dict(DEF, mixconv_kernel_sizes="[5],[7,11],[9,15],[23]"),
           dict(NOTEBOOK, mixconv_kernel_sizes="[5],[9],[13],[21]"),
           dict(DEF, pointwise_filters="64,64,64,64"),
           dict(NOTEBOOK, pointwise_filters="48,48,48,48"),
           dict(NOTEBOOK, stride=1))


def check_forward_parity(lib, B=5, T=194, training=False, grid=None, flags=DEF, lowp_tap_tol=5e-3):
    om = perturbed_oracle(T, flags=flags)
    lay, eng = make_engine(lib, T, max(B, 2), om, flags=flags)
    if grid:
        for k in ("grid_fwd", "grid_head"):
            eng.set_option(k, grid)
    rng = np.random.default_rng(7)
    x = synth_x(rng, B, T)
    eng.set_batch(x)
    eng.forward(B, training=training)
    pr, z, _ = eng.read_outputs(B, want_loss=False)
    taps = {}
    zo, _ = om.logits(x, training, taps=taps)
    po = torch.sigmoid(zo).numpy()
    # bf16 mode vs the operand-rounding oracle: bounded by the rounding-boundary noise described below
    fwd_tol = 5e-3 if _lowp(flags) else FWD_TOL
    assert np.abs(pr - po).max() <= fwd_tol, (pr, po)
    assert np.abs(z - zo.detach().numpy()).max() <= (2e-2 if _lowp(flags) else 1e-3) * max(1.0, np.abs(zo.detach().numpy()).max())
    for k, b in enumerate(lay.blocks):
        got = eng.debug_read("p%d" % (k + 1), B, B * b.tout * b.cout).reshape(B, b.tout, b.cout)
        ref = taps["b%d.r0.pre_bn" % k].detach().numpy()
        # bf16 mode: an engine fp32 operand and its oracle fp64 twin ~1e-6 apart round to different bf16
        # values with probability ~3e-4, each a 0.4 % operand error
        # (the bound is on the MAXIMUM over B * T_k * C elements of a heavy-tailed flip noise: a batch four times the size
        # needs `lowp_tap_tol` raised - the 99.99th percentile keeps the small-batch bound)
        tap_tol = lowp_tap_tol if _lowp(flags) else 2e-5
        err = np.abs(got - ref)
        assert err.max() <= tap_tol * max(1.0, np.abs(ref).max()), (k, err.max())
        if _lowp(flags):
            assert np.quantile(err, 0.9999) <= 5e-3 * max(1.0, np.abs(ref).max()), (k, np.quantile(err, 0.9999))
    eng.close()
    return float(np.abs(pr - po).max())


def check_gradients_unimposed(lib, B=1024, T=194, bound=1e-2, seed=11, flags=None, kind="mixednet", noise_factor=None):
    """One train step against the float64 oracle WITHOUT reading the engine's ReLU decisions back: the comparison the
    mask-imposing checks cannot give (a wrong mask would be copied into the oracle there).  A float32-vs-float64 flip
    of a near-zero unit may move a tensor's gradient by ~1/sqrt(units), hence the loose per-tensor L2 bound; a mask
    bug moves it by O(1).  `kind` "inception" runs the conv/BN graph engine (dropout mask injected), "graph_mixednet" a
    MixedNet flag set on the generic graph kernels, `flags` any topology.  `noise_factor` (the bf16 modes; topologies whose
    sub-spectral BN slots sum many channels into gradients that nearly cancel): the bound of a tensor is that factor times the
    distance between the float32 and the float64 ORACLE on the same batch - the rounding / decision-flip noise of the
    arithmetic itself, measured here - but never below `bound`.  (fp32 modes keep the tight loss / probability bounds.)"""
    rng = np.random.default_rng(seed)
    x = synth_x(rng, B, T)
    y = (rng.random(B) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
    kw = {}
    if kind == "inception":
        flags = flags or INC
        om = perturbed_inception_oracle(T, flags)
        lay, eng = make_inception_engine(lib, T, B, om, flags)
        keep = (rng.random((B, lay.t_last * lay.c_last)) >= flags["dropout"]).astype(np.float32)
        eng.set_dropout_mask(keep)
        kw = {"dropout_mask": keep}
    elif kind == "graph_mixednet":
        from microwakeword_amd.layout import GraphMixedNetLayout
        flags = flags or GRAPH_MIXEDNET
        om = perturbed_oracle(T, flags=flags)
        lay = GraphMixedNetLayout(flags, T)
        eng = native.Engine(lib=lib, **lay.engine_args(B))
        eng.set_grad_mask(lay.grad_mask())
        p0, s0 = lay.pack(om.get_weights())
        eng.set_params(p0)
        eng.set_bn_state(s0)
    else:
        om = perturbed_oracle(T, flags=flags or DEF)
        lay, eng = make_engine(lib, T, B, om, flags=flags or DEF)
    eng.set_batch(x)
    eng.set_targets(y, w)
    eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
    pr, z, loss = eng.read_outputs(B)
    lo, po, grads, _ = om.loss_and_grads(x, y, w, **kw)
    lowp = noise_factor is not None
    lowp_fwd = lowp and _lowp(flags or {})   # only the bf16 modes loosen the forward bounds

    def flat(grads):
        if kind in ("inception", "graph_mixednet"):
            return lay.pack([grads[n].numpy().astype(np.float32) if kd == "param" else np.zeros(sh, np.float32)
                             for n, sh, kd in lay.keras_vars])[0]
        return oracle_grads_native_order(lay, om, grads)

    gref = flat(grads)
    gnoise = None
    if lowp:
        om32 = (perturbed_inception_oracle(T, flags, dtype=torch.float32) if kind == "inception"
                else perturbed_oracle(T, flags=flags or DEF, dtype=torch.float32))
        lo32, po32, grads32, _ = om32.loss_and_grads(x, y, w, **kw)
        gnoise = flat(grads32)
    if lowp_fwd:
        assert abs(loss - lo) <= max(1e-3, noise_factor * abs(lo32 - lo)) * max(1.0, abs(lo)), (loss, lo, lo32)
        assert np.abs(pr - po).max() <= max(5e-3, noise_factor * float(np.abs(po32 - po).max()))
    else:
        assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo)), (loss, lo)
        assert np.abs(pr - po).max() <= FWD_TOL
    g = eng.get_grads()
    scale = max(1e-6, float(np.abs(gref).max()))
    off, worst = 0, 0.0
    for name, n in lay.segments():
        a, r = g[off:off + n], gref[off:off + n]
        if name.endswith("dw.bias"):     # true gradient exactly zero (cancelled by the BatchNorm): noise, bounded absolutely
            tol = 2e-3 * scale if not lowp else max(2e-3 * scale, noise_factor * float(np.abs(gnoise[off:off + n] - r).max()))
            assert np.abs(a - r).max() <= tol, (name, np.abs(a - r).max())
            off += n
            continue
        den = max(np.linalg.norm(r), 1e-3 * scale * np.sqrt(n))
        l2 = float(np.linalg.norm(a - r) / den)
        lim = bound if not lowp else max(bound, noise_factor * float(np.linalg.norm(gnoise[off:off + n] - r) / den))
        off += n
        assert l2 <= lim, (name, l2, lim)
        worst = max(worst, l2)
    eng.close()
    return worst


def check_train_steps(lib, B=6, T=194, steps=2, grid=2, lr=1e-3, graphs=False, flags=DEF):
    """loss, probabilities, flat gradient, Adam-updated weights, BN moving statistics and the
    metric counters after `steps` train_on_batch calls."""
    om = perturbed_oracle(T, flags=flags)
    lay, eng = make_engine(lib, T, B, om, flags=flags)
    if grid:
        for k in ("grid_fwd", "grid_bwd", "grid_head"):
            eng.set_option(k, grid)
    if graphs:
        eng.set_option("graphs", 1)
    rng = np.random.default_rng(11)
    worst = {}
    engine_outputs = []   # (probabilities, logits, labels) of every step as the engine produced them
    oracle_probs = []     # the float64 oracle's probabilities of every step
    # bf16-operand mode: the oracle rounds the same operands, but an fp32 value (engine) and its fp64 twin
    # (oracle) within 1e-7 of a bf16 rounding boundary round apart (a few per 1e5 operands, each a 0.4 %
    # operand error), so the bounds are those of that noise instead of fp32 rounding
    # In that mode a ReLU network's gradient is also far more exposed to decision flips (an operand error
    # eps flips ~0.4*eps of the units, moving the gradient by ~sqrt of that), so the engine's own ReLU
    # decisions are read back, checked to differ from the oracle's only at near-zero values, and imposed
    # on the oracle (as in the Inception check): what is compared are identical graphs.
    lowp = _lowp(flags)
    loss_tol, l2_tol, el_tol, med_tol = (1e-3, 2e-2, 6e-2, 6e-3) if lowp else (1e-5, 1e-4, 1e-3, 2e-5)
    if flags.get("st_bf16"):
        # two more rounded tensors per block; the float32 and the float64 oracle differ from each other by as much
        # (median 3e-3, worst 1e-2 per tensor on the default topology: the noise is that of the mode, not of the engine)
        l2_tol, med_tol = 3e-2, 1.2e-2
    for s in range(steps):
        x = synth_x(rng, B, T)
        y = (rng.random(B) < 0.5).astype(np.float32)
        w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.train_step(B, lr)
        pr, z, loss = eng.read_outputs(B)
        engine_outputs.append((pr.copy(), z.copy(), y.copy()))
        # The engine's own ReLU decisions at the BN outputs are read back, checked to differ from the float64 oracle's only
        # where the oracle's value is within rounding of zero, and imposed on the oracle: the gradients compared below are
        # those of identical graphs.  (One flipped unit moves every upstream gradient by ~1/sqrt(units): a random sweep over
        # (frames, batch) sizes hit such a unit in 13 % of the cases, each time with |value| < 2e-7.)
        taps, masks, flips = {}, {}, 0
        om.logits(x, True, taps=taps)
        if "conv1.pre" in taps and not flags.get("st_bf16"):
            # ... and the first convolution's own ReLU (no BN in front of it): the B = 600 notebook batch of the GPU suite holds
            # one pre-activation of 2e-7 that float32 and float64 put on different sides of zero (3e-4 of conv1's gradient)
            ref0 = taps["conv1.pre"].detach().numpy()
            a0 = eng.debug_read("a0", B, ref0.size).reshape(ref0.shape)
            m0 = a0 > 0
            d0 = m0 != (ref0 > 0)
            flips += int(d0.sum())
            assert np.abs(ref0[d0]).max(initial=0.0) <= (2e-2 if lowp else 2e-5) * max(1.0, np.abs(ref0).max()), np.abs(ref0[d0]).max()
            masks["conv1"] = np.ascontiguousarray(m0.transpose(0, 2, 1))
        for k, b in enumerate(lay.blocks):
            pk = eng.debug_read("p%d" % (k + 1), B, B * b.tout * b.cout).reshape(B, b.tout, b.cout).astype(np.float64)
            bn = eng.debug_read("bn%d" % (k + 1), B, 9 * b.cout).reshape(9, b.cout).astype(np.float64)
            m = (pk * bn[0] + bn[1]) > 0
            ref = taps["b%d.r0.bn_out" % k].detach().numpy()
            diff = m != (ref > 0)
            flips += int(diff.sum())
            assert np.abs(ref[diff]).max(initial=0.0) <= (2e-2 if lowp else 2e-5) * max(1.0, np.abs(ref).max()), (k, np.abs(ref[diff]).max())
            masks["b%d.r0" % k] = np.ascontiguousarray(m.transpose(0, 2, 1))
        n_units = sum(B * b.tout * b.cout for b in lay.blocks)
        assert flips <= (2e-3 * n_units if lowp else max(8, 2e-5 * n_units)), flips
        lo, po, grads, _ = om.loss_and_grads(x, y, w, relu_masks=masks)
        oracle_probs.append(np.asarray(po, np.float64).copy())
        g = eng.get_grads()
        gref = oracle_grads_native_order(lay, om, grads)
        scale = max(1e-6, float(np.abs(gref).max()))
        worst["grad"] = max(worst.get("grad", 0), float(np.abs(g - gref).max() / scale))
        assert abs(loss - lo) <= loss_tol * max(1.0, abs(lo)), (loss, lo)
        assert np.abs(pr - po).max() <= (5e-3 if lowp else FWD_TOL)
        # per parameter tensor: error relative to that tensor's own gradient scale.  The depthwise
        # biases are followed by a BatchNorm, so their true gradient is exactly zero: what any fp32
        # implementation returns there is cancellation noise (sum of O(B*T) terms), bounded in
        # absolute terms instead.
        off = 0
        for name, n in lay.segments():
            a, r = g[off:off + n], gref[off:off + n]
            off += n
            if name.endswith("dw.bias"):
                assert np.abs(a - r).max() <= (2e-2 if lowp else 2e-3) * scale, (s, name, np.abs(a - r).max(), scale)
            else:
                # fp32 (engine) vs fp64 (oracle): an activation within ~1e-7 of zero can take the other
                # side of the ReLU in one of them (expected ~once per 1e6 activations); such a flip moves a
                # handful of weight gradients by that element's contribution.  Hence a tight bound on the
                # relative L2 error of the tensor and a looser one on any single element.
                seg_scale = max(float(np.abs(r).max()), 1e-3 * scale)
                l2 = float(np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-3 * scale * np.sqrt(n)))
                # measured: 1e-6 typical.  The BN gamma / beta and dense-bias gradients are short signed sums that may nearly
                # cancel: float32 summation noise reaches a few 1e-4 of their norm in ~1 % of random cases
                loose = name == "dense.bias" or name.endswith((".bn.gamma", ".bn.beta"))
                assert l2 <= (max(l2_tol, 1e-3) if loose else l2_tol), (s, name, l2)
                assert np.abs(a - r).max() <= el_tol * seg_scale, (s, name, np.abs(a - r).max(), seg_scale)
                worst["grad"] = max(worst.get("grad", 0), l2)
                worst.setdefault("l2s", []).append(l2)
        om.train_step(x, y, w, lr, relu_masks=masks)
        p_ref, s_ref = lay.pack(om.get_weights())
        p_got, s_got = eng.get_params(), eng.get_bn_state()
        # Adam normalises every step to ~lr, so compare in units of lr.  Parameters whose true gradient
        # is (mathematically) zero — the depthwise biases, cancelled by the BatchNorm that follows —
        # carry only fp32 rounding noise that Adam amplifies to O(lr) in ANY fp32 implementation:
        # they are excluded here and bounded by 2*lr instead.
        well = np.abs(gref) > (1e-2 if lowp else 1e-4) * scale
        assert np.abs(p_got - p_ref)[well].max() <= (0.2 if lowp else 0.05) * lr, (s, np.abs(p_got - p_ref)[well].max())
        assert np.abs(p_got - p_ref).max() <= 2.0 * lr
        assert np.abs(s_got - s_ref).max() <= (1e-4 if lowp else 1e-5) * max(1.0, np.abs(s_ref).max())
        worst["param"] = max(worst.get("param", 0), float(np.abs(p_got - p_ref)[well].max()))
        # keep both sides on identical weights so that errors do not compound between steps
        eng.set_params(p_ref)
        eng.set_bn_state(s_ref)
    m = native.metrics_from_raw(eng.metrics_raw())
    r = om.metrics.result()
    if lowp:
        # a probability 1e-4 away from one of the 101 cutoffs may be bucketed on the other side; at thousands of windows
        # (probabilities up to 5e-3 apart against a bucket width of 1e-2) every cutoff sees a few crossings in both directions:
        # the counts are then bounded per cutoff
        n_win = B * steps
        slack, nz = (1, 2) if n_win <= 64 else (1 + n_win // 512, 101)
        for k in ("tp", "fp", "tn", "fn"):
            assert np.abs(m[k] - r[k]).max() <= slack and np.count_nonzero(m[k] != r[k]) <= nz, (k, np.abs(m[k] - r[k]).max())
    else:
        # (1) the metric kernel, exactly: the reference's metric definitions (oracle Metrics) applied to the ENGINE's own
        # probabilities / logits give the engine's counters bit for bit
        exact = mo.Metrics()
        for pe, ze, ye in engine_outputs:
            exact.update(pe, ye, ze)
        e = exact.result()
        for k in ("accuracy", "recall", "precision", "auc"):
            assert abs(m[k] - e[k]) < 1e-9, (k, m[k], e[k])
        for k in ("tp", "fp", "tn", "fn"):
            np.testing.assert_array_equal(m[k], e[k])
        # (2) ... and against the float64 oracle's own probabilities the counters differ only by windows whose probability
        # sits within float32 rounding of one of the cutoffs (a 600-window batch meets such a window in ~20 % of the cases:
        # 200 cutoffs x 2e-6): at most a handful, each moving a counter by one
        n_win = B * steps
        slack = 0 if n_win <= 64 else 1 + n_win // 256
        # (a small batch is held to equality unless the oracle's own probability of a window lies within float32 rounding of
        # one of the 101 cutoffs k / 100 - tools/gpu_table_fuzz.py case 1433, one window in ~3000 random cases)
        pall = np.concatenate(oracle_probs) * 100.0
        slack += int((np.abs(pall - np.round(pall)) < 2e-4).sum())
        for k in ("accuracy", "recall", "precision"):
            assert abs(m[k] - r[k]) <= (1e-6 if slack == 0 else (slack + 1.0) / n_win), (k, m[k], r[k])
        # (the AUC has its own 200 cutoffs k / 199: a window whose oracle probability lies within float32 rounding of one of them
        # moves one point of the ROC curve - tools/gpu_table_fuzz.py case 3183 on the emulator, 3e-7 away)
        near_auc = int((np.abs(pall / 100.0 * 199.0 - np.round(pall / 100.0 * 199.0)) < 4e-4).sum())
        labels = np.concatenate([ye for _, _, ye in engine_outputs])
        fewer = max(1.0, min(float((labels > 0.5).sum()), float((labels <= 0.5).sum())))   # one window moves TPR or FPR at one cutoff by 1 / its class size
        assert abs(m["auc"] - r["auc"]) <= (1e-6 if slack == 0 else 1e-3) + near_auc / fewer, (m["auc"], r["auc"])
        for k in ("tp", "fp", "tn", "fn"):
            assert np.abs(m[k] - r[k]).max() <= slack and np.count_nonzero(m[k] != r[k]) <= 2 * slack, (k, np.abs(m[k] - r[k]).max())
    assert abs(m["loss"] - r["loss"]) < (loss_tol if lowp else 1e-5)
    worst["l2_max"], worst["l2_median"] = float(np.max(worst["l2s"])), float(np.median(worst["l2s"]))
    assert np.median(worst.pop("l2s")) <= med_tol, (worst["l2_median"], worst["l2_max"])   # the typical tensor agrees to fp32 rounding (bf16 mode: to its boundary noise)
    mm, vv, step = eng.get_opt_state()
    assert step == steps
    eng.close()
    return worst


def check_saturated_logits_loss(lib, B=8, T=194):
    """Saturated logits (|z| ~ 40): the default loss is the logits form Keras 3 + TensorFlow evaluates for
    train.py:206 (no clip: the loss grows like |z| and dL/dz = w (p - y) / B stays alive); option
    "bce_from_logits" 0 switches the engine to the clipped probability form (loss capped near 16, zero gradient).
    Both are compared with the oracle's restatement of the same form."""
    for logits_form in (True, False):
        om = perturbed_oracle(T)
        ws = om.get_weights()
        names = [v.name for v in om.vars]
        ws[names.index("dense.bias")] = np.float32([40.0])
        om.set_weights(ws)
        lay, eng = make_engine(lib, T, B, om)
        if not logits_form:
            eng.set_option("bce_from_logits", 0)
        rng = np.random.default_rng(11)
        x = synth_x(rng, B, T)
        y = (np.arange(B) % 2).astype(np.float32)
        w = (1.0 + 0.5 * rng.random(B)).astype(np.float32)
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.metrics_reset()
        eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
        pr, z, loss = eng.read_outputs(B)
        assert z.min() > 20.0                      # every sample saturated towards 1
        old = mo.BCE_FROM_LOGITS
        mo.BCE_FROM_LOGITS = logits_form
        try:
            lo, po, grads, _ = om.loss_and_grads(x, y, w)
            met = mo.Metrics()
            met.update(po, y, om.last_logits)
        finally:
            mo.BCE_FROM_LOGITS = old
        assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo)), (logits_form, loss, lo)
        if logits_form:
            assert lo > 0.4 * 30.0                # the wrong (y = 0) half pays ~|z| each, weights >= 1
        else:
            assert lo < 0.5 * 1.5 * 16.2           # ... or at most -log(1.19e-7) = 15.9 each
        g = eng.get_grads()
        gref = oracle_grads_native_order(lay, om, grads)
        scale = float(np.abs(gref).max())
        if logits_form:
            assert scale > 1e-3                    # the gradient is alive
            assert np.abs(g - gref).max() <= 2e-3 * scale, np.abs(g - gref).max() / scale
        else:
            assert scale < 1e-6 and np.abs(g).max() < 1e-6   # dead zone of the clip
        m = native.metrics_from_raw(eng.metrics_raw())
        assert abs(m["loss"] - met.result()["loss"]) <= 1e-5 * max(1.0, met.result()["loss"]), (logits_form, m["loss"], met.result()["loss"])
        eng.close()


def check_variable_batch_sizes(lib, T=194, sizes=(16, 4, 4, 1, 16), graphs=False):
    """train_on_batch with a different batch size per call (fewer windows than statistics accumulator rows after a
    larger batch): every step has to match a fresh engine that starts from the same weights.  Regression test for
    stale rows in the BN statistics hand-over (common.hip.h publish_stat)."""
    om = perturbed_oracle(T)
    rng = np.random.default_rng(3)
    lay, eng = make_engine(lib, T, max(sizes), om)
    if graphs:
        eng.set_option("graphs", 1)
    for B in sizes:
        x = synth_x(rng, B, T)
        y = (rng.random(B) < 0.5).astype(np.float32)
        w = np.ones(B, np.float32)
        p0, s0 = eng.get_params(), eng.get_bn_state()
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.train_step(B, 1e-3)
        loss = eng.read_outputs(B)[2]
        g = eng.get_grads()
        lay2, fresh = make_engine(lib, T, max(sizes), om)
        fresh.set_params(p0)
        fresh.set_bn_state(s0)
        fresh.set_batch(x)
        fresh.set_targets(y, w)
        fresh.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
        assert abs(loss - fresh.read_outputs(B)[2]) <= 1e-6 * max(1.0, abs(loss)), (B, loss, fresh.read_outputs(B)[2])
        gf = fresh.get_grads()
        assert np.abs(g - gf).max() <= 1e-6 * max(1e-6, np.abs(gf).max()), (B, np.abs(g - gf).max())
        fresh.close()
    eng.close()


def check_training_reduces_loss(lib, B=8, T=194, steps=8):
    om = mo.OracleModel("mixednet", DEF, T, seed=3, dtype=torch.float32)
    lay, eng = make_engine(lib, T, B, om)
    rng = np.random.default_rng(0)
    x = synth_x(rng, B, T)
    y = (np.arange(B) % 2).astype(np.float32)
    x[y > 0.5, 60:80, :] += 8.0
    eng.set_batch(x)
    eng.set_targets(y, np.ones(B, np.float32))
    losses = []
    for _ in range(steps):
        eng.train_step(B, 1e-3)
        losses.append(eng.read_outputs(B)[2])
    eng.close()
    assert losses[-1] < losses[0], losses
    return losses


# ------------------------------------------------------------------------------------------ inception
INC = dict(mo.INCEPTION_DEFAULTS)
# every option away from its default: two stem layers, dilation, sub-spectral groups inside the blocks
INC_VARIANT = dict(cnn1_filters="16,24", cnn1_kernel_sizes="3,5", cnn1_subspectral_groups="2,4", cnn2_filters1="12,16",
                   cnn2_filters2="10,16", cnn2_kernel_sizes="3,5", cnn2_subspectral_groups="2,1", cnn2_dilation="2,1", dropout=0.3)


def perturbed_inception_oracle(T, flags, seed=42, dtype=torch.float64):
    om = mo.OracleModel("inception", flags, T, seed=seed, dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    ws = []
    for v, w in zip(om.vars, om.get_weights()):
        if v.name.endswith(("bias", "beta", "moving_mean")):
            w = w + rng.normal(0, 0.1, w.shape).astype(np.float32)
        if v.name.endswith(("gamma", "moving_variance")):
            w = w + np.abs(rng.normal(0, 0.2, w.shape)).astype(np.float32)
        ws.append(w)
    om.set_weights(ws)
    return om


def make_inception_engine(lib, T, max_batch, om, flags, fuse_heads=True):
    from microwakeword_amd.layout import InceptionLayout
    lay = InceptionLayout(flags, T, fuse_heads=fuse_heads)
    eng = native.Engine(lib=lib, **lay.engine_args(max_batch))
    p, s = lay.pack(om.get_weights())
    eng.set_params(p)
    eng.set_bn_state(s)
    return lay, eng


def check_inception_forward(lib, B=3, T=194, training=False, grid=None, flags=INC, fuse_heads=True):
    om = perturbed_inception_oracle(T, flags)
    lay, eng = make_inception_engine(lib, T, max(B, 2), om, flags, fuse_heads)
    if grid:
        for k in ("grid_graph", "grid_head"):
            eng.set_option(k, grid)
    rng = np.random.default_rng(7)
    x = synth_x(rng, B, T)
    eng.set_batch(x)
    # mww_forward(training=1) uses batch statistics but no dropout (Dropout belongs to the train step)
    eng.forward(B, training=training)
    pr, z, _ = eng.read_outputs(B, want_loss=False)
    taps = {}
    keep = np.ones((B, lay.t_last * lay.c_last), np.float32) * (1.0 - flags["dropout"])   # keep/(1-rate) == 1
    zo, _ = om.logits(x, training, dropout_mask=keep if training else None, taps=taps)
    po = torch.sigmoid(zo).numpy()
    for k, (members, op) in enumerate(zip(lay.op_members, lay.ops)):
        got = eng.debug_read("p%d" % (k + 1), B, B * op["tout"] * op["filters"]).reshape(B, op["tout"], op["filters"])
        # a fused op holds its Keras layers side by side along the channel axis
        ref = np.concatenate([taps[name + ".pre_bn"].detach().numpy() for name, _, _ in members], axis=2)
        assert got.shape == ref.shape, (members, got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (members, np.abs(got - ref).max())
    assert np.abs(pr - po).max() <= FWD_TOL, (pr, po)
    assert np.abs(z - zo.detach().numpy()).max() <= 1e-3 * max(1.0, np.abs(zo.detach().numpy()).max())
    eng.close()
    return float(np.abs(pr - po).max())


def check_inception_train_steps(lib, B=4, T=194, steps=2, grid=2, lr=1e-3, graphs=False, flags=INC, fuse_heads=True, options=None):
    om = perturbed_inception_oracle(T, flags)
    lay, eng = make_inception_engine(lib, T, B, om, flags, fuse_heads)
    if grid:
        for k in ("grid_graph", "grid_head"):
            eng.set_option(k, grid)
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    if graphs:
        eng.set_option("graphs", 1)
    rng = np.random.default_rng(11)
    l2s = []
    for s in range(steps):
        x = synth_x(rng, B, T)
        y = (rng.random(B) < 0.5).astype(np.float32)
        w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
        keep = (rng.random((B, lay.t_last * lay.c_last)) >= flags["dropout"]).astype(np.float32)
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.set_dropout_mask(keep)
        eng.train_step(B, lr)
        pr, z, loss = eng.read_outputs(B)
        # fp32 (engine) vs fp64 (oracle): a BN output within float32 rounding of zero can land on the other
        # side of the ReLU (a few per million activations).  The decisions the engine took are read back
        # (exact sign of fma(p, scale, shift)), checked to differ from the oracle's only at such near-zero
        # values, and then imposed on the oracle so that the gradient comparison is between identical graphs.
        taps = {}
        om.logits(x, True, dropout_mask=keep, taps=taps)
        masks, flips = {}, 0
        for k, (members, op) in enumerate(zip(lay.op_members, lay.ops)):
            n_el = B * op["tout"] * op["filters"]
            pk = eng.debug_read("p%d" % (k + 1), B, n_el).reshape(B, op["tout"], op["filters"]).astype(np.float64)
            bn = eng.debug_read("bn%d" % (k + 1), B, 9 * op["filters"]).reshape(9, op["filters"]).astype(np.float64)
            m_all = (pk * bn[0] + bn[1]) > 0
            for name, c0, cn in members:
                m = m_all[:, :, c0:c0 + cn]
                ref = taps[name + ".bn_out"].detach().numpy()
                diff = m != (ref > 0)
                flips += int(diff.sum())
                assert np.abs(ref[diff]).max(initial=0.0) <= 2e-5 * max(1.0, np.abs(ref).max()), (name, np.abs(ref[diff]).max())
                masks[name] = np.ascontiguousarray(m.transpose(0, 2, 1))
        n_units = sum(B * op["tout"] * op["filters"] for op in lay.ops)
        assert flips <= max(8, 2e-5 * n_units), (flips, n_units)   # 21 of 5e7 units at B = 1024, each within 2e-5 of zero
        lo, po, grads, _ = om.loss_and_grads(x, y, w, dropout_mask=keep, relu_masks=masks)
        g = eng.get_grads()
        gref = lay.pack([grads[n].numpy().astype(np.float32) if kind == "param" else np.zeros(shape, np.float32)
                         for n, shape, kind in lay.keras_vars])[0]
        assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo)), (loss, lo)
        assert np.abs(pr - po).max() <= FWD_TOL
        scale = max(1e-6, float(np.abs(gref).max()))
        off = 0
        for name, n in lay.segments():
            a, r = g[off:off + n], gref[off:off + n]
            off += n
            seg_scale = max(float(np.abs(r).max()), 1e-3 * scale)
            l2 = float(np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-3 * scale * np.sqrt(n)))
            # dense.bias and the BN / sub-spectral gamma, beta gradients are signed sums that may nearly cancel (and have as
            # few as 1-4 entries): float32 summation noise reaches a few 1e-4 of their norm in ~1 % of random topologies
            loose = name == "dense.bias" or name.endswith((".bn.gamma", ".bn.beta"))
            assert l2 <= (1e-3 if loose else 1e-4), (s, name, l2)
            assert np.abs(a - r).max() <= 1e-3 * seg_scale, (s, name, np.abs(a - r).max(), seg_scale)
            l2s.append(l2)
        om.train_step(x, y, w, lr, dropout_mask=keep, relu_masks=masks)
        p_ref, s_ref = lay.pack(om.get_weights())
        p_got, s_got = eng.get_params(), eng.get_bn_state()
        well = np.abs(gref) > 1e-4 * scale
        assert np.abs(p_got - p_ref)[well].max() <= 0.05 * lr, (s, np.abs(p_got - p_ref)[well].max())
        assert np.abs(p_got - p_ref).max() <= 2.0 * lr
        assert np.abs(s_got - s_ref).max() <= 1e-5 * max(1.0, np.abs(s_ref).max())
        eng.set_params(p_ref)
        eng.set_bn_state(s_ref)
    m = native.metrics_from_raw(eng.metrics_raw())
    r = om.metrics.result()
    for k in ("accuracy", "recall", "precision", "auc"):
        assert abs(m[k] - r[k]) < 1e-6, (k, m[k], r[k])
    for k in ("tp", "fp", "tn", "fn"):
        np.testing.assert_array_equal(m[k], r[k])
    assert not l2s or np.median(l2s) <= 2e-5
    eng.close()
    return max(l2s) if l2s else 0.0


def check_inception_generated_dropout(lib, B=4, T=194, flags=INC):
    """Built-in mask generator: the kept fraction matches the rate, the mask changes every step, the
    same seed reproduces it, and forward/backward use the same mask (gradient of a dropped input is 0)."""
    om = perturbed_inception_oracle(T, flags)
    lay, eng = make_inception_engine(lib, T, B, om, flags)
    rng = np.random.default_rng(5)
    x = synth_x(rng, B, T)
    y = (np.arange(B) % 2).astype(np.float32)
    eng.set_batch(x)
    eng.set_targets(y, np.ones(B, np.float32))
    n = lay.t_last * lay.c_last
    masks = []
    for rep in range(2):
        eng.set_option("dropout_seed", 1234)
        for step in range(2):
            eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
            k = eng.debug_read("keep", B, B * n).reshape(B, n)
            masks.append(k.copy())
            vals = np.unique(k)
            assert set(np.round(vals, 5)) <= {0.0, round(1.0 / (1.0 - flags["dropout"]), 5)}
            assert abs((k > 0).mean() - (1.0 - flags["dropout"])) < 0.04
            lo, po, grads, _ = om.loss_and_grads(x, y, np.ones(B), dropout_mask=(k > 0).astype(np.float32))
            _, _, loss = eng.read_outputs(B)
            assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo))
    assert not np.array_equal(masks[0], masks[1])
    np.testing.assert_array_equal(masks[0], masks[2])
    np.testing.assert_array_equal(masks[1], masks[3])
    eng.close()


# ------------------------------------------------------------------------------------------ validation
class _HostOnly:
    """Hides the device fast path of a FeatureHandler so validate_nonstreaming takes the reference's
    get_data -> evaluate route."""

    def __init__(self, fh):
        self._fh = fh

    def get_data(self, *a, **k):
        return self._fh.get_data(*a, **k)

    def get_mode_size(self, m):
        return self._fh.get_mode_size(m)

    def get_mode_duration(self, m):
        return self._fh.get_mode_duration(m)


def check_validation_on_device(lib, gold, tag="u16"):
    """validate_nonstreaming (train.py:41-163) with the windows kept in HBM gives the same numbers as the
    host-array route, consumes the same RNG draws, and its counters match the oracle's Keras-metric
    restatement fed with the oracle model's own predictions."""
    from microwakeword_amd import train as tr
    from microwakeword_amd.model import Model
    T = 194
    om = perturbed_oracle(T)
    model = Model(DEF, (T, 40), 16, lib=lib, max_batch=16)
    model.set_weights(om.get_weights())
    cfg = dict(golden_config(gold, tag), batch_size=16, spectrogram_length=T)
    fh = FeatureHandler(cfg, engine=model.engine)
    random.seed(5)
    np.random.seed(5)
    fast = tr.validate_nonstreaming(cfg, fh, model, "validation")
    tail_fast = np.random.random()
    fast_counts = {k: model.evaluation_results()[k].numpy().copy() for k in ("tp", "fp", "tn", "fn")}
    random.seed(5)
    np.random.seed(5)
    slow = tr.validate_nonstreaming(cfg, _HostOnly(fh), model, "validation")
    assert tail_fast == np.random.random()
    assert set(fast) == set(slow)
    for k in fast:
        assert np.allclose(fast[k], slow[k], rtol=0, atol=1e-12), (k, fast[k], slow[k])
    for k, v in fast_counts.items():
        np.testing.assert_array_equal(v, model.evaluation_results()[k].numpy())
    # oracle: same windows through the CPU restatement + Keras-metric bucketing
    np.random.seed(5)
    xv, yv, _ = fh.get_data("validation", 16, T, "truncate_start")
    xa, ya, _ = fh.get_data("validation_ambient", 16, T, "split")
    met = mo.Metrics()
    met.update(*om.predict_with_logits(xv)[:1], yv, om.predict_with_logits(xv)[1])
    met.update(*om.predict_with_logits(xa)[:1], ya, om.predict_with_logits(xa)[1])
    r = met.result()
    for k in ("tp", "fp", "tn", "fn"):
        np.testing.assert_array_equal(fast_counts[k], r[k])
    assert abs(fast["auc"] - r["auc"]) < 1e-6 and abs(fast["loss"] - r["loss"]) < 1e-5
    model.engine.close()
    return fast


# ------------------------------------------------------------------------------------------ the whole loop
def learnable_config(n=64, seed=0, T=60):
    """Two providers whose windows differ by a band of raised energy: a tiny separable wake-word task."""
    rng = np.random.default_rng(seed)

    def samples(count, positive, lo, hi):
        out = []
        for _ in range(count):
            L = int(rng.integers(lo, hi))
            s = rng.integers(0, 200, size=(L, 40)).astype(np.uint16)
            if positive:
                s[-40:-20, 8:24] += 400
            out.append(s)
        return out

    pos = {"training": [samples(n, True, T + 5, T + 40)], "validation": [samples(n // 2, True, T + 5, T + 40)]}
    neg = {"training": [samples(n, False, T + 5, T + 40)], "validation": [samples(n // 2, False, T + 5, T + 40)],
           "validation_ambient": [samples(4, False, 4 * T, 6 * T)]}
    return {"stride": 1, "window_step_ms": 10, "features": [
        dict(type="mmap", stores=pos, truth=True, sampling_weight=1.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
        dict(type="mmap", stores=neg, truth=False, sampling_weight=1.0, penalty_weight=1.0, truncation_strategy="truncate_start")]}


def check_train_loop_end_to_end(lib, tmp_path, B=16, steps=12, kind="mixednet", min_val_accuracy=None):
    """model_train_eval's objects through microwakeword_amd.train.train: schedule, on-device batches, periodic
    validation (device-resident), best-weights rule, checkpoint + restore — and the model learns the task."""
    from microwakeword_amd import inception, mixednet
    from microwakeword_amd import train as tr
    T = 60
    flags = DEF if kind == "mixednet" else dict(INC, dropout=0.1)
    module = mixednet if kind == "mixednet" else inception
    cfg = dict(learnable_config(T=T), train_dir=str(tmp_path / "run"), summaries_dir=str(tmp_path / "run" / "logs"), batch_size=B,
               spectrogram_length=T, training_steps=[steps // 2, steps - steps // 2], learning_rates=[0.01, 0.003],
               time_mask_max_size=[0], time_mask_count=[0], freq_mask_max_size=[0], freq_mask_count=[0],
               positive_class_weight=[1.0], negative_class_weight=[1.0], eval_step_interval=steps // 3, target_minimization=0.9,
               minimization_metric=None, maximization_metric="accuracy")
    random.seed(1)
    np.random.seed(1)
    model = module.model(flags, (T, 40), B, lib=lib, seed=7, max_batch=64)
    fh = FeatureHandler(cfg, engine=model.engine)
    out = tr.train(model, cfg, fh, verbose=False)
    if min_val_accuracy is not None:
        # validation runs in inference mode on the BN moving averages (momentum 0.99): they need a few hundred
        # steps to converge, so the learning claim is only made for long enough runs
        assert out["best_maximization"] >= min_val_accuracy, out
    run = tmp_path / "run"
    for f in ("best_weights.weights.h5.npz", "last_weights.weights.h5.npz", "restore/ckpt.weights.npz", "restore/ckpt.opt.npz",
              "logs/train/scalars.jsonl", "logs/validation/scalars.jsonl"):
        assert (run / f).exists(), f
    # restore: a fresh model picks up weights + Adam state and keeps the validation quality
    w_last = model.get_weights()
    m2 = module.model(flags, (T, 40), B, lib=lib, seed=99, max_batch=64)
    m2.load_weights(str(run / "last_weights.weights.h5"))
    for a, b in zip(w_last, m2.get_weights()):
        np.testing.assert_array_equal(a, b)
    m2.load_optimizer_state(str(run / "restore" / "ckpt.opt.npz"))
    assert m2.engine.get_opt_state()[2] == steps
    fh2 = FeatureHandler(cfg, engine=m2.engine)
    nm = tr.validate_nonstreaming(dict(cfg), fh2, m2, "validation")
    assert 0.0 <= nm["ambient_false_positives_per_hour"]
    if min_val_accuracy is not None:
        assert nm["accuracy"] >= min_val_accuracy - 0.05, nm
    model.engine.close()
    m2.engine.close()
    return out


# ------------------------------------------------------------------------------------------ generic MixedNet
# flag combinations outside the specialised block kernels: odd filter counts, a block without depthwise,
# repeat_in_block 2, multi-kernel groups, strided 5x1 first conv — and one without a first conv at all
GRAPH_MIXEDNET = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0", pointwise_filters="24,32,40", repeat_in_block="1,2,1",
                      mixconv_kernel_sizes="[3],[1],[3,5]", first_conv_filters=16, first_conv_kernel_size=5, stride=2)
GRAPH_MIXEDNET_RESIDUAL = dict(mo.MIXEDNET_DEFAULTS, residual_connection="1,0,1", pointwise_filters="24,24,32", repeat_in_block="2,1,1",
                               mixconv_kernel_sizes="[5],[3,7],[1]", first_conv_filters=16, first_conv_kernel_size=3, stride=1)
GRAPH_MIXEDNET_HEADS = [dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0", pointwise_filters="24,32", repeat_in_block="1,1",
                             mixconv_kernel_sizes="[5],[7]", first_conv_filters=16, spatial_attention=sa, pooled=po, max_pool=mp)
                        for sa, po, mp in ((1, 0, 0), (0, 1, 0), (0, 1, 1), (1, 1, 0), (1, 1, 1))]
# residual branches + the spatial-attention gate + the pooled head in one model (the default widths and kernels)
GRAPH_MIXEDNET_FULL = dict(mo.MIXEDNET_DEFAULTS, residual_connection="1,0,1,0", spatial_attention=1, pooled=1)
GRAPH_MIXEDNET_NOCONV1 = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0", pointwise_filters="16,24", repeat_in_block="1,1",
                              mixconv_kernel_sizes="[5],[7,9]", first_conv_filters=0)


def check_graph_mixednet(lib, flags=GRAPH_MIXEDNET, B=3, T=100, steps=1, grid=2, graphs=False, lr=1e-3, bn_inline=None, options=None):
    """Forward intermediates and train step of a MixedNet running on the generic conv/BN graph kernels.  bn_inline: None =
    the engine's default (graphs of convolutions + BN and depthwise ops + bias hand their statistics over; residual /
    attention / pooled graphs use finalize launches), 0 = finalize launches everywhere."""
    from microwakeword_amd.layout import GraphMixedNetLayout
    om = perturbed_oracle(T, flags=flags)
    lay = GraphMixedNetLayout(flags, T)
    assert [n for n, _, _ in lay.keras_vars] == [v.name for v in om.vars]
    eng = native.Engine(lib=lib, **lay.engine_args(B))
    eng.set_grad_mask(lay.grad_mask())
    p, s = lay.pack(om.get_weights())
    eng.set_params(p)
    eng.set_bn_state(s)
    if grid:
        for k in ("grid_graph", "grid_head"):
            eng.set_option(k, grid)
    if graphs:
        eng.set_option("graphs", 1)
    if bn_inline is not None:
        eng.set_option("bn_inline", bn_inline)
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    rng = np.random.default_rng(13)
    wts = dict(zip([n for n, _, _ in lay.keras_vars], om.get_weights()))
    for training in (False, True):
        x = synth_x(rng, B, T)
        eng.set_batch(x)
        eng.forward(B, training=training)
        pr, z, _ = eng.read_outputs(B, want_loss=False)
        taps = {}
        zo, _ = om.logits(x, training, taps=taps)
        for k, (name, op) in enumerate(zip(lay.op_names, lay.ops)):
            got = eng.debug_read("p%d" % (k + 1), B, B * op["tout"] * op["filters"]).reshape(B, op["tout"], op["filters"])
            if name == "conv1":
                got, ref = np.maximum(got, 0), taps["conv1"].detach().numpy()
            elif name.endswith(".res"):
                continue   # residual branch: checked through the block outputs it is added to
            elif name not in ("conv1",) and not name.endswith((".dw", ".pw")):
                continue
            elif name.endswith(".dw"):
                it = lay.items[k]
                bias = np.concatenate([wts[lay.keras_vars[vi + 1][0]] for vi, _, _ in it["groups"]])
                got, ref = got + bias, taps[name].detach().numpy()
            else:
                ref = taps[name[:-3] + ".pre_bn"].detach().numpy()
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())
        assert np.abs(pr - torch.sigmoid(zo).numpy()).max() <= FWD_TOL
    l2s = []
    for st in range(steps):
        x = synth_x(rng, B, T)
        y = (rng.random(B) < 0.5).astype(np.float32)
        w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
        # float32 resolves 1 - p only to 6e-8, so the probability-form BCE of the engine (and of the reference's float32
        # graph) carries an absolute error of up to 6e-8 * e^|z| per window on top of ordinary rounding
        step_taps = {}
        zabs = om.logits(x, True, taps=step_taps)[0].abs().detach().numpy()
        loss_slack = float(np.sum(w * 1.2e-7 * np.exp(np.minimum(zabs, 16.0))) / B)
        # A unit whose pre-activation is within float32 rounding of zero may take the other side of its ReLU in the float32
        # engine than in the float64 oracle; with these tiny batches one such unit moves every upstream gradient by ~1 %
        # (seen for 2 of 200 random topologies).  Such a step is only held to the looser bound.
        near_zero = 0
        for key, tap in step_taps.items():
            if key == "conv1" or key.endswith(".bn_out"):
                v = np.abs(tap.detach().numpy())
                near_zero += int((v < 4e-6 * max(1.0, float(v.max()))).sum())
        if flags.get("spatial_attention"):
            # the attention gate takes a max over the channels of every frame: a near tie (seen at 1.4e-7 and 3.4e-7 of the
            # tensor's maximum, tools/emu_fuzz.py cases 3468 and 1331) lets float32 pick the other channel and the gradient
            # takes the other route - the same kind of one-unit decision as a ReLU at zero
            last = [k for k in step_taps if k.endswith(".bn_out")][-1]
            act = np.maximum(step_taps[last].detach().numpy(), 0.0)
            top2 = np.sort(act, axis=2)[:, :, -2:]
            near_zero += int(((top2[:, :, 1] - top2[:, :, 0]) < 4e-6 * max(1.0, float(act.max()))).sum())
        grad_tol = 1e-3 if near_zero == 0 else 5e-2
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.train_step(B, lr)
        pr, z, loss = eng.read_outputs(B)
        lo, po, grads, _ = om.loss_and_grads(x, y, w)
        g = eng.get_grads()
        gref = lay.pack([grads[n].numpy().astype(np.float32) if kind == "param" else np.zeros(shape, np.float32)
                         for n, shape, kind in lay.keras_vars])[0]
        assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo)) + loss_slack, (loss, lo, loss_slack)
        scale = max(1e-6, float(np.abs(gref).max()))
        off = 0
        for name, n in lay.segments():
            a, r = g[off:off + n], gref[off:off + n]
            off += n
            if name.endswith(".dw.bias"):   # followed by a BatchNorm: the true gradient is zero, what remains is cancellation noise
                assert np.abs(a - r).max() <= 2e-3 * scale, (st, name, np.abs(a - r).max())
                continue
            l2 = float(np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-3 * scale * np.sqrt(n)))
            assert l2 <= grad_tol, (st, name, l2, near_zero)
            if near_zero == 0:
                l2s.append(l2)
        # structural zero taps of fused MixConv groups stay exactly zero in the gradient
        assert np.all(g[lay.grad_mask() == 0] == 0)
        om.train_step(x, y, w, lr)
        p_ref, s_ref = lay.pack(om.get_weights())
        well = np.abs(gref) > (1e-4 if near_zero == 0 else 0.2) * scale
        # every step starts from the oracle's weights, moving statistics AND Adam moments (installed below), so a step in
        # which a one-unit decision went the other way does not loosen the steps after it: the update is always held to
        # 5 % of an Adam step (a flagged step itself: only where the gradient is well away from the flipped unit's share)
        assert np.abs(eng.get_params() - p_ref)[well].max() <= 0.05 * lr, (st, near_zero)
        assert np.abs(eng.get_bn_state() - s_ref).max() <= 1e-5 * max(1.0, np.abs(s_ref).max())
        eng.set_params(p_ref)
        eng.set_bn_state(s_ref)
        tr = [v.name for v in om.vars if v.trainable]
        for slot in ("m", "v"):
            d = dict(zip(tr, getattr(om.adam, slot)))
            flat = lay.pack([d[n].numpy().astype(np.float32) if kind == "param" else np.zeros(shape, np.float32)
                             for n, shape, kind in lay.keras_vars])[0]
            if slot == "m":
                m_flat = flat
        eng.set_opt_state(m_flat, flat, om.adam.t)
    assert not l2s or np.median(l2s) <= 2e-5
    eng.close()
    return max(l2s) if l2s else 0.0


# ------------------------------------------------------------------------------------------ data fuzz
def random_data_case(seed):
    """A random feature-set + augmentation policy: 2-3 providers, uint16 or float32 stores, every training
    truncation strategy, fixed_right_cutoffs, mask sizes 0..12 and counts 0..3, short and long samples."""
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.choice([60, 194]))
    dtype = np.uint16 if rng.random() < 0.6 else np.float32
    provs = []
    for pi in range(int(rng.integers(2, 4))):
        n = int(rng.integers(5, 30))
        lens = rng.integers(max(8, T - 40), T + 120, size=n)
        if dtype == np.uint16:
            store = [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16) for l in lens]
        else:
            store = [(rng.random((int(l), 40), dtype=np.float32) * np.float32(26.0)) for l in lens]
        strat = str(rng.choice(["truncate_start", "truncate_end", "random", "fixed_right_cutoff"]))
        cut = [int(c) for c in rng.integers(0, 8, size=int(rng.integers(1, 3)))] if strat == "fixed_right_cutoff" else [0]
        provs.append(dict(store=store, truth=bool(pi == 0 or rng.random() < 0.3), sampling_weight=float(rng.choice([0.5, 1.0, 2.0, 10.0])),
                          penalty_weight=float(rng.choice([0.5, 1.0, 1.5])), truncation_strategy=strat, fixed_right_cutoffs=cut))
    policy = dict(freq_mix_prob=0.0, time_mask_max_size=int(rng.choice([0, 1, 5, 12])), time_mask_count=int(rng.integers(0, 4)),
                  freq_mask_max_size=int(rng.choice([0, 1, 5, 12])), freq_mask_count=int(rng.integers(0, 4)))
    # a cutoff larger than the spare frames raises in the reference: keep the lengths compatible
    for p in provs:
        if p["truncation_strategy"] == "fixed_right_cutoff":
            p["store"] = [s if (s.shape[0] <= T or s.shape[0] - T >= max(p["fixed_right_cutoffs"])) else s[:T] for s in p["store"]]
    return T, provs, policy, int(rng.integers(4, 40)), str(rng.choice(["default", "default", "truncate_start", "random"]))


def check_data_fuzz(lib, cases=12, first=0):
    """Engine (host sampler + HIP assembly) against the oracle on random feature sets, policies and strategies:
    identical windows, masks, labels, weights, shuffle order and RNG tails."""
    for case in range(first, first + cases):
        T, provs, policy, B, strategy = random_data_case(case)
        _, eng = make_engine(lib, T, B) if T == 194 else (None, native.Engine(lib=lib, **MixedNetLayout(DEF, T).engine_args(B)))
        cfg = {"stride": 1, "window_step_ms": 10, "features": [
            dict(type="mmap", stores={"training": [p["store"]]}, truth=p["truth"], sampling_weight=p["sampling_weight"],
                 penalty_weight=p["penalty_weight"], truncation_strategy=p["truncation_strategy"],
                 fixed_right_cutoffs=p["fixed_right_cutoffs"]) for p in provs]}
        random.seed(case)
        np.random.seed(case)
        fh = FeatureHandler(cfg, engine=eng)
        got = [fh.get_data("training", B, T, strategy, policy) for _ in range(2)]
        tail = (random.random(), np.random.random())
        random.seed(case)
        np.random.seed(case)
        op = [do.index_provider({"training": [p["store"]]}, p["truth"], p["sampling_weight"], p["penalty_weight"],
                                p["truncation_strategy"], 1, 0.01, p["fixed_right_cutoffs"]) for p in provs]
        for (x, y, w) in got:
            xo, yo, wo, _, _ = do.get_data(op, "training", B, T, strategy, policy)
            np.testing.assert_array_equal(x, xo, err_msg="case %d" % case)
            np.testing.assert_array_equal(y, yo)
            np.testing.assert_array_equal(w, wo)
        assert tail == (random.random(), np.random.random()), case
        eng.close()


# ------------------------------------------------------------------------------------------ topology fuzz
def random_mixednet_flags(seed):
    """A random MixedNet flag set inside the limits of the graph kernels."""
    rng = np.random.default_rng(5000 + seed)
    nb = int(rng.integers(1, 4))
    widths = [8, 12, 16, 24, 32, 40, 48, 64]
    pf = [int(rng.choice(widths)) for _ in range(nb)]
    ks = []
    for _ in range(nb):
        n = int(rng.choice([1, 1, 2]))
        k = sorted(int(v) for v in rng.choice([1, 3, 5, 7, 9], size=n, replace=False))
        ks.append(k)
    f0 = int(rng.choice([0, 8, 16, 32]))
    flags = dict(mo.MIXEDNET_DEFAULTS, pointwise_filters=",".join(map(str, pf)) if nb > 1 else str(pf[0]),
                 repeat_in_block=",".join(str(int(rng.integers(1, 3))) for _ in range(nb)) if nb > 1 else str(int(rng.integers(1, 3))),
                 mixconv_kernel_sizes=",".join(str(k) for k in ks) if nb > 1 else str(ks[0]) + ",",
                 residual_connection=",".join(str(int(rng.random() < 0.4)) for _ in range(nb)) if nb > 1 else str(int(rng.random() < 0.4)),
                 first_conv_filters=f0, first_conv_kernel_size=int(rng.choice([3, 5])), stride=int(rng.choice([1, 1, 2, 3])) if f0 else 1,
                 spatial_attention=int(rng.random() < 0.3), pooled=int(rng.random() < 0.3), max_pool=int(rng.random() < 0.5))
    return flags


def check_topology_fuzz(lib, cases=6, first=0, B=3):
    done = 0
    for case in range(first, first + 4 * cases):
        flags = random_mixednet_flags(case)
        T = 70
        try:
            check_graph_mixednet(lib, flags, B=B, T=T, steps=1, grid=2)
        except ValueError as e:            # too short for this kernel stack / channel split: a legitimate refusal, draw again
            if "too short" in str(e) or "at least 4 frames" in str(e):
                continue
            raise
        done += 1
        if done == cases:
            return
    raise AssertionError("too few valid random topologies")


# ------------------------------------------------------------------------------------------ frozen oracle outputs
def check_against_frozen_oracle(lib, golden_dir):
    """The engine against the committed fixture tests/golden/model_oracle_golden.npz (outputs of the oracle frozen by
    tests/golden/make_golden_model.py): eval / train probabilities and the loss of the default MixedNet, the notebook
    topology and the default Inception on the fixture's inputs."""
    import importlib.util
    from microwakeword_amd.layout import InceptionLayout
    spec = importlib.util.spec_from_file_location("make_golden_model", os.path.join(golden_dir, "make_golden_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(os.path.join(golden_dir, "model_oracle_golden.npz"))
    for name in ("mixednet_default", "mixednet_notebook", "inception_default"):
        kind, flags, T = mod.CASES[name]
        om = mo.OracleModel(kind, flags, T, seed=42)          # only its initial weights are used (numpy, no forward)
        lay = MixedNetLayout(flags, T) if kind == "mixednet" else InceptionLayout(flags, T)
        eng = native.Engine(lib=lib, **lay.engine_args(4))
        eng.set_grad_mask(lay.grad_mask())
        p, s = lay.pack(om.get_weights())
        eng.set_params(p)
        eng.set_bn_state(s)
        rng = np.random.default_rng(7)
        x = (rng.integers(0, 667, size=(4, T, 40)).astype(np.float32) * SCALE).astype(np.float32)
        y = np.array([1, 0, 0, 1], np.float32)
        w = np.array([1.0, 0.5, 2.0, 1.0], np.float32)
        eng.set_batch(x)
        eng.forward(4, training=False)
        assert np.abs(eng.read_outputs(4, want_loss=False)[0] - gold[name + "/p_eval"]).max() <= FWD_TOL, name
        if kind == "inception":
            n = lay.t_last * lay.c_last
            eng.set_dropout_mask((rng.random((4, n)) >= flags["dropout"]).astype(np.float32))
        eng.set_targets(y, w)
        eng.train_step(4, 1e-3, flags=native.STEP_NO_APPLY)
        pr, _, loss = eng.read_outputs(4)
        assert np.abs(pr - gold[name + "/p_train"]).max() <= FWD_TOL, name
        assert abs(loss - float(gold[name + "/loss"])) <= 1e-5 * max(1.0, abs(float(gold[name + "/loss"]))), name
        eng.close()


def check_against_reference_graph_fixture(lib, golden_dir, names=None):
    """The engine, through the package's drop-in builders (``mixednet.model`` / ``inception.model``), against
    tests/golden/ref_graph_golden.npz: what the REFERENCE'S OWN ``mixednet.py`` / ``inception.py`` compute (executed by
    oracle/ref_model_shim.py over float64 stand-ins of the Keras layer primitives, frozen by tests/golden/make_golden_ref_graph.py)
    on the fixture's inputs - inference and training probabilities, loss, the gradient of every trainable variable, the BN
    moving statistics after the step.  The reference's variable order is the order ``set_weights`` takes."""
    import importlib.util
    from microwakeword_amd import inception as amd_inception, mixednet as amd_mixednet
    spec = importlib.util.spec_from_file_location("make_golden_ref_graph", os.path.join(golden_dir, "make_golden_ref_graph.py"))
    sys.path.insert(0, golden_dir)
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(golden_dir)
    gold = np.load(os.path.join(golden_dir, "ref_graph_golden.npz"))
    report = {}
    for name in (names or mod.CASES):
        kind, flags, T = mod.CASES[name]
        x, y, w = gold[name + "/x"], gold[name + "/y"], gold[name + "/w"]
        B = x.shape[0]
        trainable = gold[name + "/trainable"]
        values = [gold["%s/value/%03d" % (name, i)] for i in range(len(trainable))]
        m = (amd_mixednet if kind == "mixednet" else amd_inception).model(flags, (T, 40), B, lib=lib, max_batch=B)
        lay, eng = m.layout, m.engine
        assert [tuple(sh) for _, sh, _ in lay.keras_vars] == [v.shape for v in values], name   # the reference's creation order and shapes
        m.set_weights(values)
        eng.set_batch(x)
        eng.forward(B, training=False)
        assert np.abs(eng.read_outputs(B, want_loss=False)[0] - gold[name + "/p_eval"]).max() <= FWD_TOL, name
        if name + "/keep" in gold.files:
            eng.set_dropout_mask(gold[name + "/keep"])
        eng.set_targets(y, w)
        eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
        pr, z, loss = eng.read_outputs(B)
        ref_loss = float(gold[name + "/loss"])
        assert np.abs(pr - gold[name + "/p_train"]).max() <= FWD_TOL, name
        assert np.abs(z - gold[name + "/z_train"]).max() <= 1e-4 * max(1.0, np.abs(gold[name + "/z_train"]).max()), name
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (name, loss, ref_loss)
        arrs = [gold["%s/grad/%03d" % (name, i)].astype(np.float32) if t else np.zeros(v.shape, np.float32)
                for i, (t, v) in enumerate(zip(trainable, values))]
        gref = lay.pack(arrs)[0] * lay.grad_mask() if hasattr(lay, "grad_mask") and lay.grad_mask() is not None else lay.pack(arrs)[0]
        g = eng.get_grads()
        # float32 engine vs the float64 graph with these tiny batches: a unit within rounding of zero may sit on the other side of
        # its ReLU (moves upstream gradients by ~1/sqrt(units)); the bound is the one the un-imposed oracle checks use
        l2 = float(np.linalg.norm(g - gref) / max(np.linalg.norm(gref), 1e-30))
        assert l2 <= 1e-2, (name, l2)
        stats = lay.pack([gold["%s/moving/%03d" % (name, i)].astype(np.float32) if ("%s/moving/%03d" % (name, i)) in gold.files
                          else np.zeros(v.shape, np.float32) for i, v in enumerate(values)])[1]
        got = eng.get_bn_state()
        assert np.abs(got - stats).max() <= 1e-5 * max(1.0, np.abs(stats).max()), (name, np.abs(got - stats).max())
        report[name] = l2
        eng.close()
    return report


def check_head_frame_limit_is_refused_at_creation(lib, T=436, B=2):
    """64 channels x 390 final frames (T = 436 on the default kernels): more than the classifier head kernel holds (24 frames per
    frame group: 384 at 64 channels).  The block kernels must refuse the shape when the model is created, not at the first
    forward (tools/gpu_x6_fuzz.py case 460); windows this long do not fit the graph kernels' LDS tiles either, so
    ``mixednet.model`` ends in MWW_ERR_UNSUPPORTED - at creation, naming both reasons."""
    import logging
    from microwakeword_amd import mixednet
    flags = dict(DEF, pointwise_filters="48,48,48,64")
    assert mixednet.kernel_family(flags, T - 6, lib=lib)[0] == "block"        # 384 final frames: the widest head instantiation
    fam, why = mixednet.kernel_family(flags, T, lib=lib)
    assert fam == "graph" and "final frames" in why, (fam, why)
    eng = None
    try:
        lay = MixedNetLayout(flags, T)
        eng = native.Engine(lib=lib, **lay.engine_args(B))
    except native.NativeError as e:
        assert "error -3" in str(e) and "final frames" in str(e), e
    assert eng is None, "the block kernels accepted a head they have no kernel for"
    logging.disable(logging.WARNING)
    try:
        mixednet.model(flags, (T, 40), B, lib=lib, max_batch=B)
        raise AssertionError("expected MWW_ERR_UNSUPPORTED at creation")
    except native.NativeError as e:
        assert "error -3" in str(e), e
    finally:
        logging.disable(logging.NOTSET)


def check_prefetched_training_matches_synchronous(lib, B=8, T=60, steps=7):
    """Batches drawn ahead by the prefetcher's worker thread (native.Prefetcher, csrc/sampler.cpp) against the synchronous
    sampler on the launching thread: the same private streams give the same windows / masks / labels / weights in the same
    order, so parameters and outputs after a few train steps are bit-identical - also across a change of the augmentation
    policy and of the class weights (the prefetcher is rebuilt from the stream positions of the last batch handed out) and
    across a synchronous draw in between (the streams go back to the synchronous sampler and return)."""
    from microwakeword_amd import mixednet
    pol_a = dict(time_mask_max_size=4, time_mask_count=2, freq_mask_max_size=4, freq_mask_count=2)
    pol_b = dict(time_mask_max_size=6, time_mask_count=1, freq_mask_max_size=3, freq_mask_count=3)
    results = []
    for depth in (0, 1, 3):
        random.seed(3)
        np.random.seed(3)
        model = mixednet.model(DEF, (T, 40), B, lib=lib, seed=11, max_batch=B)
        eng = model.engine
        fh = FeatureHandler(learnable_config(T=T), engine=eng)
        fh.use_private_rng(prefetch=depth)
        seen, targets = [], []
        for k in range(steps):
            pol, cw = (pol_a, (1.0, 1.0)) if k < 3 else (pol_b, (0.5, 2.0))
            got = fh.next_training_batch_on_device(B, T, "default", pol, class_weights=cw, want_targets=True)
            assert (fh._pf is not None) == (depth > 0)
            if depth:   # (labels, per-sample weights) as they went to the device
                targets.append(np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in got]))
            eng.train_step(B, 1e-2)
            if k == 4:   # a synchronous draw between two prefetched batches
                d = fh.draw_training_batch(B, T, "default", pol_b)
                seen.append(np.asarray(d["masks"], np.float32).reshape(-1))
        seen.append(eng.read_outputs(B)[0].copy())
        seen.append(eng.get_batch(B).copy())
        results.append((eng.get_params().copy(), seen, targets))
        fh._drop_prefetcher()
        eng.close()
    for r in results[1:]:
        np.testing.assert_array_equal(results[0][0], r[0])
        for a, b in zip(results[0][1], r[1]):   # the synchronous draw's masks, last outputs, last batch
            np.testing.assert_array_equal(a, b)
    assert len(results[1][2]) == steps
    for a, b in zip(results[1][2], results[2][2]):
        np.testing.assert_array_equal(a, b)


def check_train_loop_prefetch_is_schedule_only(lib, tmp_path, B=8, T=60, steps=7):
    """train.train with the batches drawn on a worker thread from private RNG streams (prefetch_batches 2, the default)
    against prefetch_batches 0 (every draw on the launching thread, in the global generators, as the reference does):
    bit-identical weights ACROSS validation passes, whose shuffles (data.py:593-595) advance the global numpy stream the
    next training draws continue from, and with a provider whose 'random' truncation draws from that stream - the loop
    hands the streams back around every validation.  The global generators end at the same point too."""
    from microwakeword_amd import mixednet
    from microwakeword_amd import train as tr
    outs = []
    for depth in (0, 2):
        cfg = learnable_config(T=T)
        cfg["features"][1]["truncation_strategy"] = "random"   # consumes numpy.random per drawn sample
        cfg = dict(cfg, train_dir=str(tmp_path / ("run%d" % depth)), summaries_dir=str(tmp_path / ("run%d" % depth) / "logs"),
                   batch_size=B, spectrogram_length=T, training_steps=[steps], learning_rates=[0.01], time_mask_max_size=[4],
                   time_mask_count=[2], freq_mask_max_size=[4], freq_mask_count=[1], positive_class_weight=[1.0],
                   negative_class_weight=[1.0], eval_step_interval=max(2, steps // 2), target_minimization=0.9, minimization_metric=None,
                   maximization_metric="accuracy", prefetch_batches=depth)
        random.seed(2)
        np.random.seed(2)
        model = mixednet.model(DEF, (T, 40), B, lib=lib, seed=5, max_batch=64)
        fh = FeatureHandler(cfg, engine=model.engine)
        tr.train(model, cfg, fh, verbose=False)
        if depth:
            fh.release_private_rng()
        outs.append((model.engine.get_params().copy(), model.engine.get_bn_state().copy(), random.random(), float(np.random.random())))
        model.engine.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][2:] == outs[1][2:]


def check_train_loop_data_parallel_world1(lib, tmp_path, backend):
    """microwakeword_amd.train.train as the single rank of a ``torch.distributed`` job (config ``data_parallel``): sharding,
    the communicator (the library's own RCCL one on the GPU), the weight / optimizer-state broadcast, the gradient
    exchange in every step, the sharded validation's counter all-reduce and the BN-state averaging all run through the
    collective path; with one rank the run must equal the plain loop bit for bit."""
    import socket

    import torch
    import torch.distributed as dist
    from microwakeword_amd import mixednet
    from microwakeword_amd import train as tr
    T, B = 60, 16

    def run(tag, **extra):
        cfg = dict(learnable_config(T=T), train_dir=str(tmp_path / tag), summaries_dir=str(tmp_path / tag / "logs"), batch_size=B,
                   spectrogram_length=T, training_steps=[6], learning_rates=[0.01], time_mask_max_size=[3], time_mask_count=[1],
                   freq_mask_max_size=[3], freq_mask_count=[1], positive_class_weight=[1.0], negative_class_weight=[1.0],
                   eval_step_interval=3, target_minimization=0.9, minimization_metric=None, maximization_metric="accuracy", **extra)
        random.seed(9)
        np.random.seed(9)
        model = mixednet.model(DEF, (T, 40), B, lib=lib, seed=3, max_batch=64)
        fh = FeatureHandler(cfg, engine=model.engine)
        if not extra:
            # the plain loop on what rank 0 of a one-rank job draws from: canonical sample order, streams seeded 0 (data_parallel_seed 0)
            for p in fh.feature_providers:
                p.feature_sets["training"] = sorted(p.feature_sets["training"])
            random.seed(0)
            np.random.seed(0)
        out = tr.train(model, cfg, fh, verbose=False)
        res = (model.engine.get_params().copy(), model.engine.get_bn_state().copy(), out, model.data_parallel)
        fh._drop_prefetcher()
        model.engine.close()
        return res

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
    try:
        p1, s1, o1, dp = run("dp", data_parallel=True, data_parallel_seed=0)   # streams seeded (0 * W + rank) * 1000003 + step 0 = 0
        assert dp is not None and dp.world == 1
    finally:
        dist.destroy_process_group()
    p0, s0, o0, none = run("plain")
    assert none is None
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(s0, s1)
    assert o0 == o1
    return dp


# ------------------------------------------------------------------------------------------ statistics hand-over
def check_bn_inline_matches_finalize(lib, B=12, T=194, steps=3, flags=DEF):
    """"bn_inline" (BN sums in replicated fp64 accumulator rows, folded by their first consumer) against the
    finalize-launch path: same arithmetic on the same sums, so parameters, moving statistics and outputs agree to
    fp32 rounding of a differently ordered fp64 sum (identical in practice; tolerance 1e-6 relative).  Also
    through captured graphs (the accumulator parity is part of the graph key) and with a training-mode forward
    between steps (flips the forward parity only)."""
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 667, size=(steps + 1, B, T, 40)).astype(np.float32) * SCALE).astype(np.float32)
    y = (rng.random((steps + 1, B)) < 0.4).astype(np.float32)
    w = np.ones(B, np.float32)
    om = mo.OracleModel("mixednet", flags, T, seed=42)
    lay = MixedNetLayout(flags, T)
    p0, s0 = lay.pack(om.get_weights())
    outs = []
    for inline, graphs, tail in ((0, 0, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1)):
        eng = native.Engine(lib=lib, **lay.engine_args(B))
        eng.set_grad_mask(lay.grad_mask())
        eng.set_params(p0)
        eng.set_bn_state(s0)
        eng.set_option("bn_inline", inline)
        eng.set_option("tail_roles", tail)   # dense-weight gradient + metrics inside the gradient-reduction launch
        eng.set_option("graphs", graphs)
        probs = []
        for k in range(steps):
            eng.set_batch(x[k])
            eng.set_targets(y[k], w)
            eng.train_step(B, 1e-2)
            probs.append(eng.read_outputs(B)[0].copy())
            if k == 0:
                eng.set_batch(x[steps])
                eng.forward(B, training=True)
                probs.append(eng.read_outputs(B, want_loss=False)[0].copy())
        m = native.metrics_from_raw(eng.metrics_raw())
        counts = np.concatenate([np.asarray(m[k], np.float64).ravel() for k in ("tp", "fp", "fn", "tn")])
        outs.append((eng.get_params().copy(), eng.get_bn_state().copy(), np.concatenate(probs), eng.get_grads().copy(), counts))
        eng.close()
    ref = outs[0]
    for got in outs[1:]:
        for a, b in zip(ref, got):
            np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-7)


def check_inception_static_shapes_are_schedule_only(lib, B=9, lengths=(100, 194, 208, 212), steps=3, grid=0, combos=None):
    """The static-shape instantiations of the graph kernels (kernels_graph.hip.h GShape: the default Inception ops) against
    the run-time-shape kernels ("graph_static_shapes" 0): the same convolutions in the same order - the shapes fold index
    computations - so parameters, moving statistics, probabilities and gradients agree over several steps; to float32
    rounding, since the stem's static form sums its BN statistics per accumulator column instead of per row group
    (gconv_body "DIRECT": pre-BN outputs identical, the batch statistics differ in the last bit).  Among the static
    combinations (planar tensors, captured graphs) and among the run-time ones the results are bit-identical.
    Window lengths on both sides of the static path's limit (208 frames), a batch smaller than the grid in between, eager
    and through captured graphs."""
    rng = np.random.default_rng(17)
    for T in lengths:
        om = perturbed_inception_oracle(T, INC)
        x = (rng.integers(0, 667, size=(steps, B, T, 40)).astype(np.float32) * SCALE).astype(np.float32)
        y = (rng.random((steps, B)) < 0.4).astype(np.float32)
        w = rng.choice([0.5, 1.0, 2.0], size=B).astype(np.float32)
        outs, kinds, evals = [], [], []
        # ... and so is the planar layout of the fused branch heads' tensors ("graph_planar": one plane per consumer slice)
        for static, planar, graphs in (combos or ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1))):
            kinds.append(static)
            lay, eng = make_inception_engine(lib, T, B, om, INC)
            eng.set_option("graph_static_shapes", static)
            eng.set_option("graph_planar", planar)
            eng.set_option("graphs", graphs)
            if grid:
                eng.set_option("grid_graph", grid)
            # inference-mode forward first (moving statistics: no batch sums in the way): the two families run the same
            # convolutions in the same order, so the logits must be BIT-identical - a dropped boundary row or column of a static
            # shape shows here, where no ReLU-flip tolerance hides it (round-5 advisor finding)
            eng.set_batch(x[0])
            eng.forward(B, training=False)
            evals.append(eng.read_outputs(B, want_loss=False)[1].copy())
            got = []
            for k in range(steps):
                nb = B if k != 1 else min(B, 2)
                eng.set_batch(x[k][:nb])
                eng.set_targets(y[k][:nb], w[:nb])
                eng.set_dropout_mask(np.ones((nb, eng_dense_inputs(lay)), np.uint8))
                eng.train_step(nb, 1e-2)
                got.append(eng.read_outputs(nb)[0].copy())
                got.append(eng.get_grads().copy())
            got += [eng.get_params().copy(), eng.get_bn_state().copy()]
            outs.append(got)
            eng.close()
        for ev in evals[1:]:
            np.testing.assert_array_equal(evals[0], ev, err_msg="inference logits of the static / run-time-shape kernels at T = %d" % T)
        for kind, other in zip(kinds[1:], outs[1:]):
            same = outs[kinds.index(kind)]   # first run on the same kernel family
            for a, b in zip(same, other):
                np.testing.assert_array_equal(a, b)
            # across the families only the first step is comparable: probabilities to rounding, gradients within what one ReLU
            # unit flipped by the last bit of a batch statistic moves (~1e-3).  Later steps are not: the same run-time-shape
            # kernels on two grids (other partial sums) are 0.5 % apart in the second step's gradient and 26 % in the third's
            # at T = 100 (tools/gpu_static_diag.py, profiles/round5_inception_stem_ab.txt) - the middle step's two-window
            # batch amplifies rounding.  A wrong row count or a dropped tile is O(1) in the first step already.
            for idx, tol in ((0, 1e-5), (1, 1e-2)):
                a, b = outs[0][idx], other[idx]
                err = np.linalg.norm(b.astype(np.float64) - a) / max(np.linalg.norm(a.astype(np.float64)), 1e-30)
                assert err <= tol, (T, kind, idx, err)


def check_inception_bn_inline_matches_finalize(lib, B=7, T=150, steps=3, flags=INC, fuse_heads=True):
    """The statistics hand-over of the conv/BN graph kernels (accumulator rows folded by the first consumer launch, no
    finalize launches) against the finalize-launch path on the Inception graph: same sums, same arithmetic, so parameters,
    moving statistics, probabilities, gradients agree to rounding (1e-6 relative); through captured graphs, with a
    training-mode forward between steps, with a batch smaller than the accumulator row count in between (stale rows),
    and the profile of a step lists no finalize launch."""
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 667, size=(steps + 1, B, T, 40)).astype(np.float32) * SCALE).astype(np.float32)
    y = (rng.random((steps + 1, B)) < 0.4).astype(np.float32)
    w = np.ones(B, np.float32)
    om = perturbed_inception_oracle(T, flags)
    outs = []
    for inline, graphs in ((0, 0), (1, 0), (1, 1)):
        lay, eng = make_inception_engine(lib, T, B, om, flags, fuse_heads)
        eng.set_option("bn_inline", inline)
        eng.set_option("graphs", graphs)
        eng.set_option("graph_role_split", 0)   # same workgroups per role => same summation order of the weight-gradient partials
        probs = []
        for k in range(steps):
            nb = B if k != 1 else min(B, 3)     # fewer workgroups than accumulator rows in the middle step
            eng.set_batch(x[k][:nb])
            eng.set_targets(y[k][:nb], w[:nb])
            eng.set_dropout_mask(np.ones((nb, eng_dense_inputs(lay)), np.uint8))
            eng.train_step(nb, 1e-2)
            probs.append(eng.read_outputs(nb)[0].copy())
            if k == 0:
                eng.set_batch(x[steps])
                eng.forward(B, training=True)
                probs.append(eng.read_outputs(B, want_loss=False)[0].copy())
        outs.append((eng.get_params().copy(), eng.get_bn_state().copy(), np.concatenate(probs), eng.get_grads().copy()))
        if not graphs:
            eng.set_option("profile", 1)
            eng.set_batch(x[0])
            eng.set_targets(y[0], w)
            eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
            eng.train_step(B, 1e-2, flags=native.STEP_NO_APPLY)
            names = [nm for nm, _ in eng.profile_read()]
            eng.set_option("profile", 0)
            assert any("finalize" in nm for nm in names) == (inline == 0), names
        eng.close()
    ref = outs[0]
    for got in outs[1:]:
        for a, b in zip(ref, got):
            np.testing.assert_allclose(b, a, rtol=2e-6, atol=1e-7)
    # multi-role launches that divide their workgroups between the roles (the default with the hand-over): other partial
    # sums, same gradient up to float32 summation order
    grads = []
    for split, share in ((0, 50), (1, 50), (1, 30), (1, 70)):   # share: unequal weight- / data-gradient halves
        lay, eng = make_inception_engine(lib, T, B, om, flags, fuse_heads)
        eng.set_option("graph_role_split", split)
        eng.set_option("graph_dgrad_share", share)
        eng.set_batch(x[0])
        eng.set_targets(y[0], w)
        eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
        eng.train_step(B, 1e-2, flags=native.STEP_NO_APPLY)
        grads.append(eng.get_grads().copy())
        eng.close()
    # (another summation order of the BN sums can flip a ReLU unit that sits within rounding of zero, which moves every
    # upstream gradient by ~1e-3: seen on the GPU for some (batch, workgroups) pairs, tools/gpu_split_diag.py; a wrong row
    # count or a dropped role is O(1))
    for g in grads[1:]:
        assert np.linalg.norm(g - grads[0]) <= 1e-2 * np.linalg.norm(grads[0])


def check_graph_grid_options(lib, B=6, T=120, flags=INC):
    """The grid knobs of the conv/BN graph launches (workgroups per CU forward / backward, one fixed grid) only change how the
    windows are dealt to workgroups: same gradients up to float32 summation order; values outside their ranges are refused."""
    rng = np.random.default_rng(29)
    x = synth_x(rng, B, T)
    y = (rng.random(B) < 0.4).astype(np.float32)
    w = np.ones(B, np.float32)
    om = perturbed_inception_oracle(T, flags)
    grads = []
    for opts in ({}, {"graph_fwd_wg_per_cu": 1, "graph_bwd_wg_per_cu": 1}, {"grid_graph": 3, "graph_role_split": 0}):
        lay, eng = make_inception_engine(lib, T, B, om, flags, True)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_batch(x)
        eng.set_targets(y, w)
        eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
        eng.train_step(B, 1e-2, flags=native.STEP_NO_APPLY)
        grads.append(eng.get_grads().copy())
        if not opts:
            for k, v in (("graph_fwd_wg_per_cu", 0), ("graph_bwd_wg_per_cu", 9), ("graph_dgrad_share", 5), ("graph_dgrad_share", 95), ("grid_graph", -1)):
                try:
                    eng.set_option(k, v)
                except Exception:
                    continue
                raise AssertionError("%s = %d was accepted" % (k, v))
        eng.close()
    for g in grads[1:]:
        assert np.linalg.norm(g - grads[0]) <= 1e-2 * np.linalg.norm(grads[0])


def eng_dense_inputs(lay):
    last = lay.engine_args(1)["conv_ops"][-1]
    return last["tout"] * last["filters"]


# ------------------------------------------------------------------------------------------ fused input
def check_fused_input(lib, B=8, T=60, steps=4, dtype="u16", graphs=False):
    """"fused_input" (descriptor-only batches: the first block's kernels gather, scale and mask their rows straight from
    the feature stores) against the materialised x: the gathered values are the same floats, so parameters, outputs and
    the batch read back afterwards are bit-identical.  Covers a second step on the same batch (the descriptors' mailbox
    slot is no longer current: x is materialised), an evaluation forward, and labels set after the batch."""
    from microwakeword_amd import mixednet
    policy = dict(time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)
    cfg = learnable_config(T=T)
    if dtype == "f32":
        for prov in cfg["features"]:
            for mode, sets in prov["stores"].items():
                prov["stores"][mode] = [[(s.astype(np.float32) * SCALE).astype(np.float32) for s in group] for group in sets]
    results = []
    for fused in (0, 1):
        random.seed(4)
        np.random.seed(4)
        model = mixednet.model(DEF, (T, 40), B, lib=lib, seed=13, max_batch=B)
        eng = model.engine
        eng.set_option("fused_input", fused)
        eng.set_option("graphs", 1 if (graphs and fused) else 0)   # captured steps bake the mailbox slot of the descriptors
        fh = FeatureHandler(cfg, engine=eng)
        seen = []
        for k in range(steps):
            fh.next_training_batch_on_device(B, T, "default", policy)
            eng.train_step(B, 1e-2)
            seen.append(eng.read_outputs(B)[0].copy())
            if k == 0:   # same batch again: its mailbox slot has moved on
                eng.train_step(B, 1e-2)
                seen.append(eng.read_outputs(B)[0].copy())
            if k == 1:   # evaluation forward on a fresh batch, then the batch itself
                fh.next_training_batch_on_device(B, T, "default", policy)
                eng.forward(B, training=False, update_metrics=True)
                seen.append(eng.read_outputs(B, want_loss=False)[0].copy())
                seen.append(eng.get_batch(B).copy())
            if k == 2:   # labels replaced after the batch was described
                eng.set_targets(np.ones(B, np.float32), np.full(B, 0.5, np.float32))
                eng.train_step(B, 1e-2)
                seen.append(eng.read_outputs(B)[0].copy())
        m = native.metrics_from_raw(eng.metrics_raw())
        seen.append(np.concatenate([np.asarray(m["tp"], np.float64), np.asarray(m["fp"], np.float64), np.asarray(m["fn"], np.float64)]))
        results.append((eng.get_params().copy(), eng.get_bn_state().copy(), seen))
        eng.close()
    np.testing.assert_array_equal(results[0][0], results[1][0])
    np.testing.assert_array_equal(results[0][1], results[1][1])
    for a, b in zip(results[0][2], results[1][2]):
        np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------------------------------ gather fuzz
def check_gather_fuzz(lib, cases=8, first=0, notebook=False):
    """The gather stage inside the first block's kernels (fused_input) against the materialised x on random feature
    sets / policies / strategies (random_data_case: ragged lengths around T, uint16 and float32 stores, 0..3 masks of
    0..12 frames / bins per kind): evaluation outputs and the flat gradient of a train step are bit-identical."""
    flags = NOTEBOOK if notebook else DEF
    for case in range(first, first + cases):
        T, provs, policy, B, strategy = random_data_case(case)
        if notebook:
            T = 204
        cfg = {"stride": 1, "window_step_ms": 10, "features": [
            dict(type="mmap", stores={"training": [p["store"]]}, truth=p["truth"], sampling_weight=p["sampling_weight"],
                 penalty_weight=p["penalty_weight"], truncation_strategy=p["truncation_strategy"],
                 fixed_right_cutoffs=p["fixed_right_cutoffs"]) for p in provs]}
        lay = MixedNetLayout(flags, T)
        om = mo.OracleModel("mixednet", flags, T, seed=case)
        p0, s0 = lay.pack(om.get_weights())
        outs = []
        for fused in (0, 1):
            random.seed(case)
            np.random.seed(case)
            eng = native.Engine(lib=lib, **lay.engine_args(B))
            eng.set_grad_mask(lay.grad_mask())
            eng.set_params(p0)
            eng.set_bn_state(s0)
            eng.set_option("fused_input", fused)
            fh = FeatureHandler(cfg, engine=eng)
            got = []
            for _ in range(2):
                fh.next_training_batch_on_device(B, T, strategy, policy)
                eng.forward(B, training=False)
                got.append(eng.read_outputs(B, want_loss=False)[0].copy())
                fh.next_training_batch_on_device(B, T, strategy, policy)
                eng.train_step(B, 1e-3, flags=native.STEP_NO_APPLY)
                got.append(eng.get_grads().copy())
                got.append(eng.read_outputs(B)[0].copy())
            outs.append(got)
            eng.close()
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b, err_msg="case %d" % case)


def check_inception_gathered_stem(lib, cases=3, first=0, B=6, lengths=(194, 150, 200, 201), graphs=(0,), grid=0, rounds=3):
    """The Inception stem reading a descriptor-only batch in place (kernels_graph.hip.h XG instantiations of the forward
    convolution and the weight gradient: pad / truncate, uint16 scaling and SpecAugment masks applied while the window is
    staged) against the materialised x ("fused_input" 0): the gathered values are the same floats, so evaluation outputs,
    gradients, parameters after several steps and the batch read back afterwards are bit-identical.  Random feature sets /
    policies / strategies (random_data_case: ragged lengths around T, uint16 and float32 stores, 0..3 masks per kind);
    window lengths on both sides of the gather's limit (200 frames); a second step on the same batch (its mailbox slot has
    moved on: x is written out); optionally through captured graphs and with a grid that leaves a workgroup more windows than
    the gather keeps descriptors for (then x is written out by the launch itself)."""
    for case in range(first, first + cases):
        _, provs, policy, _, strategy = random_data_case(case)
        T = lengths[case % len(lengths)]
        for p in provs:   # (random_data_case's own rule, for this T: a cutoff larger than the spare frames raises in the reference)
            if p["truncation_strategy"] == "fixed_right_cutoff":
                p["store"] = [s if (s.shape[0] <= T or s.shape[0] - T >= max(p["fixed_right_cutoffs"])) else s[:T] for s in p["store"]]
        cfg = {"stride": 1, "window_step_ms": 10, "features": [
            dict(type="mmap", stores={"training": [p["store"]]}, truth=p["truth"], sampling_weight=p["sampling_weight"],
                 penalty_weight=p["penalty_weight"], truncation_strategy=p["truncation_strategy"],
                 fixed_right_cutoffs=p["fixed_right_cutoffs"]) for p in provs]}
        om = perturbed_inception_oracle(T, INC)
        outs = []
        for fused, use_graphs in [(0, 0)] + [(1, g) for g in graphs]:
            random.seed(case)
            np.random.seed(case)
            lay, eng = make_inception_engine(lib, T, B, om, INC)
            eng.set_option("fused_input", fused)
            eng.set_option("graphs", use_graphs)
            if grid:
                eng.set_option("grid_graph", grid)
            fh = FeatureHandler(cfg, engine=eng)
            got = []
            for k in range(rounds):
                fh.next_training_batch_on_device(B, T, strategy, policy)
                eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
                eng.forward(B, training=False)
                got.append(eng.read_outputs(B, want_loss=False)[0].copy())
                fh.next_training_batch_on_device(B, T, strategy, policy)
                eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
                eng.train_step(B, 1e-2)
                got.append(eng.get_grads().copy())
                got.append(eng.read_outputs(B)[0].copy())
                if k == 0:   # same batch again
                    eng.train_step(B, 1e-2)
                    got.append(eng.get_grads().copy())
                if k == rounds - 1:
                    got.append(eng.get_batch(B).copy())
            got += [eng.get_params().copy(), eng.get_bn_state().copy()]
            outs.append(got)
            if not use_graphs:   # the assembly launch runs exactly when the stem cannot gather
                eng.set_option("profile", 1)
                fh.next_training_batch_on_device(B, T, strategy, policy)
                eng.set_dropout_mask(np.ones((B, eng_dense_inputs(lay)), np.uint8))
                eng.train_step(B, 1e-2, flags=native.STEP_NO_APPLY)
                names = [nm for nm, _ in eng.profile_read()]
                gathers = bool(fused) and T <= 200 and (not grid or -(-B // grid) <= 8)
                assert any("assemble" in nm for nm in names) == (not gathers), (fused, T, names)
            eng.close()
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                np.testing.assert_array_equal(a, b, err_msg="case %d T %d" % (case, T))



# ------------------------------------------------------------------------------------------ inception topology fuzz
def random_inception_flags(seed):
    """A random Inception flag set inside the widths the graph kernels instantiate: 1-2 stem layers, 1-3 blocks,
    kernel sizes 3/5/7, dilation 1/2, sub-spectral groups that divide the filters, dropout 0.1..0.4."""
    rng = np.random.default_rng(9000 + seed)
    ns, nb = int(rng.integers(1, 3)), int(rng.integers(1, 4))

    def groups_for(f):
        return int(rng.choice([g for g in (1, 1, 2, 4) if f % g == 0]))

    stem_f = [int(rng.choice([8, 16, 24, 32])) for _ in range(ns)]
    f1 = [int(rng.choice([8, 10, 12, 16])) for _ in range(nb)]
    f2 = [int(rng.choice([8, 10, 12, 16, 20, 24])) for _ in range(nb)]
    return dict(cnn1_filters=",".join(map(str, stem_f)), cnn1_kernel_sizes=",".join(str(int(rng.choice([3, 5]))) for _ in range(ns)),
                cnn1_subspectral_groups=",".join(str(groups_for(f)) for f in stem_f),
                cnn2_filters1=",".join(map(str, f1)), cnn2_filters2=",".join(map(str, f2)),
                cnn2_kernel_sizes=",".join(str(int(rng.choice([3, 5, 7]))) for _ in range(nb)),
                cnn2_subspectral_groups=",".join(str(int(rng.choice([g for g in (1, 1, 2) if a % g == 0 and b % g == 0]))) for a, b in zip(f1, f2)),
                cnn2_dilation=",".join(str(int(rng.choice([1, 1, 2]))) for _ in range(nb)), dropout=float(rng.choice([0.1, 0.2, 0.4])))


def check_inception_topology_fuzz(lib, cases=4, first=0, B=3, T=150):
    for case in range(first, first + cases):
        flags = random_inception_flags(case)
        try:
            check_inception_train_steps(lib, B=B, T=T, steps=1, grid=2, flags=flags)
        except AssertionError as e:
            raise AssertionError("case %d %s: %s" % (case, flags, e))


# ------------------------------------------------------------------------------------------ shape fuzz
def check_first_conv_tail_rows(lib, B=5, grid=2, lengths=(194, 197, 200, 203, 204, 206, 207, 209), topologies=None):
    """Strided first convolutions (the notebook's 5x1 stride 3): windows whose a0 length is TT + 1 ... TT + K - 1 rows run as ONE
    tile with the rows behind the 64 MFMA rows computed on the VALU (fwd_first_body.inc "tail rows"); lengths on both sides of
    that range (exactly one tile, two tiles) take the ordinary paths.  Forward taps and a train step each, two topologies."""
    for T in lengths:
        for flags in (topologies or (NOTEBOOK, CROSSED[3])):
            check_forward_parity(lib, B=B, T=T, training=True, grid=grid, flags=flags)
            check_train_steps(lib, B=B, T=T, steps=1, grid=grid, flags=flags)


def check_shape_fuzz(lib, cases=10, first=0):
    """Random (frames, batch, grid) sizes through the specialised MixedNet kernels, default and notebook topologies
    (partial time tiles, fewer windows than workgroups and the reverse, windows per workgroup 1..48)."""
    for case in range(first, first + cases):
        rng = np.random.default_rng(7000 + case)
        nb = rng.random() < 0.35
        flags = NOTEBOOK if nb else DEF
        T = int(rng.integers(110, 300)) if nb else int(rng.integers(52, 300))
        B = int(rng.integers(1, 48))
        grid = int(rng.choice([0, 1, 2, 3, 5, 8]))
        try:
            check_train_steps(lib, B=B, T=T, steps=1, grid=grid, flags=flags)
        except ValueError as e:
            if "too short" not in str(e):
                raise
        except AssertionError as e:
            raise AssertionError("case %d (notebook=%s, T=%d, B=%d, grid=%d): %s" % (case, nb, T, B, grid, e))
