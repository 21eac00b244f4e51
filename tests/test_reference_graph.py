"""the oracle's model half against the reference's own graph code.

``oracle/ref_model_shim.py`` executes ``/root/reference/microwakeword/mixednet.py`` / ``inception.py`` (with ``layers/stream.py``,
``strided_drop.py``, ``sub_spectral_normalization.py`` ...) unmodified over float64 stand-ins of the Keras layer primitives.
What the reference's files decide - MixConv split and right alignment, residual placement inside repeated blocks, attention
and pooling heads, Stream's padding, the sub-spectral reshape, Flatten order, the order variables are created in - is thereby
compared by execution with both restatements under ``oracle/`` (container only; ``/root/reference`` does not exist on the GPU
box).  The frozen outputs (``tests/golden/ref_graph_golden.npz``) are what the oracle is held to everywhere and what
``tests/test_engine_gpu.py::test_against_the_reference_graph_fixture`` holds the HIP kernels to on the MI355X.
TensorFlow itself stays absent: the primitives (a convolution, BatchNormalization's documented formula) are restated, see the
shim's header."""
import os
import sys

import numpy as np
import pytest

import engine_checks as ec
from oracle import model_oracle as mo
from oracle import model_oracle_np as mnp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden_ref_graph as mg  # noqa: E402

BASE = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0")
INC = dict(mo.INCEPTION_DEFAULTS)
TOPOLOGIES = [
    ("mixednet", BASE, 60, 4),
    ("mixednet", ec.NOTEBOOK, 204, 3),                                                     # stride 3, MixConv groups [7,11] / [9,15]
    ("mixednet", dict(BASE, pointwise_filters="64,64,64,64", residual_connection="0,1,1,1", first_conv_kernel_size=5, stride=3,
                      mixconv_kernel_sizes="[5],[7,11],[9,15],[23]"), 204, 3),             # the notebook's flags incl. its residuals
    ("mixednet", dict(BASE, residual_connection="1,1,0,1", repeat_in_block="2,1,2,1", mixconv_kernel_sizes="[5],[3,5,7],[9],[5,9]"), 70, 4),
    ("mixednet", dict(BASE, spatial_attention=1), 60, 4),
    ("mixednet", dict(BASE, spatial_attention=1, pooled=1), 60, 4),
    ("mixednet", dict(BASE, pooled=1, max_pool=1), 60, 4),
    ("mixednet", dict(BASE, first_conv_filters=0, pointwise_filters="32,32,40,48"), 60, 4),
    ("mixednet", dict(BASE, stride=2, mixconv_kernel_sizes="[5],[9],[1],[7]"), 120, 4),   # a block without a depthwise stage
    ("inception", INC, 49, 4),
    ("inception", dict(INC, cnn1_filters="24,16", cnn1_kernel_sizes="5,3", cnn1_subspectral_groups="4,2", cnn2_subspectral_groups="2,1,2",
                       cnn2_dilation="1,2,1", dropout=0.0), 70, 3),
]


def _shim():
    from oracle import ref_model_shim as rm
    if not rm.available():
        pytest.skip("reference tree not present")
    return rm


def _compare(rm, kind, flags, T, B, seed):
    """-> worst discrepancy between the reference's graph and the oracle over probabilities (both modes), loss, every
    gradient (relative to the largest gradient entry of the model) and the BN moving statistics."""
    rng = np.random.default_rng(seed)
    om = (ec.perturbed_oracle(T, seed=seed, flags=flags) if kind == "mixednet" else ec.perturbed_inception_oracle(T, flags, seed=seed))
    values = om.get_weights()
    x = ec.synth_x(rng, B, T)
    y = (rng.random(B) < 0.5).astype(np.float64)
    w = rng.choice([0.5, 1.0, 2.0], size=B)
    keep = None
    if kind == "inception" and flags.get("dropout", 0) > 0:
        keep = (rng.random((B, values[-2].shape[0])) >= flags["dropout"]).astype(np.float64)
    loss_o, p_o, g_o, stats_o = om.loss_and_grads(x, y, w, dropout_mask=keep)
    loss_r, p_r, g_r, run = rm.reference_loss_and_grads(kind, flags, x, y, w, values, dropout_mask=keep, loss_fn=mo.weighted_loss)
    assert not [k for k in ("tensorflow", "microwakeword.mixednet", "microwakeword.inception", "microwakeword.layers.stream") if k in sys.modules]   # nothing outlives the import
    assert len(run.variables) == len(om.vars)                      # (shapes were checked position by position when they were created)
    assert [v.trainable for v in run.variables] == [v.trainable for v in om.vars]
    worst = max(abs(loss_o - loss_r), float(np.abs(p_o - p_r).max()))
    gscale = max(float(g.abs().max()) for g in g_o.values())
    for v, rv, gr in zip(om.vars, run.variables, g_r):
        if v.trainable:
            worst = max(worst, float(np.abs(g_o[v.name].numpy() - gr.numpy().reshape(v.value.shape)).max()) / gscale)
        elif rv.updated is not None:
            worst = max(worst, float(np.abs(stats_o[v.name].numpy() - rv.updated.numpy()).max()))
    ev = rm.run_reference_model(kind, flags, x, values, training=False)
    z, _ = om.logits(x, training=False)
    worst = max(worst, float(np.abs(ev.logits.detach().reshape(-1).numpy() - z.detach().numpy()).max()))
    return worst


@pytest.mark.reference
@pytest.mark.parametrize("case", range(len(TOPOLOGIES)))
def test_reference_builders_agree_with_the_oracle(case):
    rm = _shim()
    kind, flags, T, B = TOPOLOGIES[case]
    assert _compare(rm, kind, flags, T, B, seed=40 + case) <= 1e-11


@pytest.mark.reference
def test_reference_builders_agree_with_the_oracle_on_random_topologies():
    """the flag generators of the GPU fuzz tests (engine_checks.random_mixednet_flags / random_inception_flags)"""
    rm = _shim()
    done = 0
    for seed in range(24):
        try:
            flags = ec.random_mixednet_flags(seed)
            mo.mixednet_build(flags, 150)
        except ValueError:
            continue
        assert _compare(rm, "mixednet", flags, 150, 3, seed) <= 1e-11, flags
        done += 1
    for seed in range(6):
        flags = dict(ec.random_inception_flags(seed), dropout=0.0 if seed % 2 else 0.2)
        assert _compare(rm, "inception", flags, 150, 3, seed) <= 1e-11, flags
        done += 1
    assert done >= 16


@pytest.mark.reference
def test_reference_builders_raise_where_the_oracle_does():
    """mixednet.py:298-305: lists of different lengths (the reference's own default --residual_connection has five entries)"""
    rm = _shim()
    x = np.zeros((2, 60, 40), np.float32)
    with pytest.raises(ValueError, match="same length"):
        rm.run_reference_model("mixednet", dict(mo.MIXEDNET_DEFAULTS), x)
    with pytest.raises(ValueError, match="same length"):
        mo.mixednet_build(dict(mo.MIXEDNET_DEFAULTS), 60)
    # a variable list in another order is refused at the first position whose shape differs
    om = mo.OracleModel("mixednet", BASE, 60)
    values = om.get_weights()
    values[1], values[3] = values[3], values[1]
    with pytest.raises(ValueError, match="variable #1"):
        rm.run_reference_model("mixednet", BASE, x, values)


@pytest.mark.reference
@pytest.mark.parametrize("kind,flags,T", [("inception", INC, 60), ("mixednet", dict(BASE, residual_connection="1,0,1,1", repeat_in_block="1,2,1,1"), 70)])
def test_weight_conversion_tools_recover_the_creation_order(kind, flags, T):
    """tools/keras_creation_order.py (used by keras_weights_to_npz.py / npz_to_keras_weights.py where Keras is installed): the log
    of constructed layers gives back the order the reference's builder creates its variables in, whatever order the model lists
    its weights in (a Keras functional model sorts layers by graph depth: not the creation order for residual blocks / Inception)."""
    rm = _shim()
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    try:
        import keras_creation_order as kco
    finally:
        sys.path.pop(0)
    x = np.zeros((2, T, 40), np.float32)
    with kco.layer_creation_log(rm.Layer) as created:
        run = rm.run_reference_model(kind, flags, x, training=False)
    listed = sorted(run.variables, key=lambda v: (v.name.split("/")[1], v.name))          # some other listing of the same variables
    assert listed != run.variables
    perm = kco.creation_permutation(listed, created)
    assert [listed[i] for i in perm] == run.variables
    with pytest.raises(RuntimeError, match="covers"):
        kco.creation_permutation(listed, created[:-1])                                   # a weight no logged layer owns is an error
    assert rm.Layer.__init__.__name__ == "__init__"                                      # the patch is undone


@pytest.mark.reference
def test_ref_graph_fixture_is_current():
    """tests/golden/ref_graph_golden.npz is what make_golden_ref_graph.py produces from the reference tree today
    (MWW_WRITE_REF_GRAPH=1 rewrites it)."""
    _shim()
    blob = mg.build()
    if os.environ.get("MWW_WRITE_REF_GRAPH") == "1":
        np.savez_compressed(mg.FIXTURE, **blob)
    gold = mg.load()
    assert sorted(gold.files) == sorted(blob)
    for k, v in blob.items():
        if v.dtype.kind in "US" or v.dtype == bool:
            assert (gold[k] == v).all(), k
        else:
            assert np.abs(gold[k].astype(np.float64) - v).max() <= 1e-12 * max(1.0, float(np.abs(v).max())), k


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_both_restatements_match_the_reference_graph_fixture(name):
    """runs everywhere: oracle/model_oracle.py (torch autograd) and oracle/model_oracle_np.py (hand-derived backward) against what
    the reference's graph returned on the fixture's inputs"""
    gold = mg.load()
    kind, flags, T, values, x, y, w, keep = mg.case_inputs(name)
    trainable = gold[name + "/trainable"]
    assert len(values) == len(trainable)
    for i, v in enumerate(values):                                  # the inputs are reproducible from the seeds
        assert (gold["%s/value/%03d" % (name, i)] == v).all()
    assert (gold[name + "/x"] == x).all() and (gold[name + "/y"] == y).all() and (gold[name + "/w"] == w).all()
    om = mo.OracleModel(kind, flags, T, seed=42)
    om.set_weights(values)
    assert np.abs(om.predict(x) - gold[name + "/p_eval"]).max() <= 1e-12
    loss, p, grads, stats = om.loss_and_grads(x, y, w, dropout_mask=keep)
    assert abs(loss - float(gold[name + "/loss"])) <= 1e-12 and np.abs(p - gold[name + "/p_train"]).max() <= 1e-12
    gscale = max(float(np.abs(gold[k]).max()) for k in gold.files if k.startswith(name + "/grad/"))
    for i, v in enumerate(om.vars):
        if v.trainable:
            assert np.abs(grads[v.name].numpy() - gold["%s/grad/%03d" % (name, i)]).max() <= 1e-11 * gscale, v.name
        else:
            assert np.abs(stats[v.name].numpy() - gold["%s/moving/%03d" % (name, i)]).max() <= 1e-12, v.name
    # the numpy twin (its own graph objects, forward and hand-written backward; sequential MixedNets and Inception)
    try:
        twin = mnp.NumpyModel(kind, flags, T)
    except NotImplementedError:
        return
    twin.set_weights({v.name: val for v, val in zip(om.vars, values)})
    assert np.abs(twin.logits(x, False) - np.log(gold[name + "/p_eval"] / (1 - gold[name + "/p_eval"]))).max() <= 1e-9
    loss_n, p_n, grads_n, stats_n = twin.loss_and_grads(x, y, w, keep=keep)
    assert abs(loss_n - float(gold[name + "/loss"])) <= 1e-11 and np.abs(p_n - gold[name + "/p_train"]).max() <= 1e-11
    for i, v in enumerate(om.vars):
        if v.trainable:
            assert np.abs(np.asarray(grads_n[v.name]).reshape(v.value.shape) - gold["%s/grad/%03d" % (name, i)]).max() <= 1e-10 * gscale, v.name
        else:
            assert np.abs(np.asarray(stats_n[v.name]) - gold["%s/moving/%03d" % (name, i)]).max() <= 1e-11, v.name
