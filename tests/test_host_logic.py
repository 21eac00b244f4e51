"""Host-side logic that needs no GPU: flag parsing / shape derivation, Keras<->native weight
mapping (incl. MixConv zero-tap fusion), config loading, the train-loop schedule and best-model
rule, validate_nonstreaming arithmetic — against the oracle / reference semantics."""
import argparse
import os
import types

import numpy as np
import pytest
import yaml

from microwakeword_amd import layout as lay
from microwakeword_amd import mixednet, model_train_eval, synthetic, train as tr
from microwakeword_amd.model import initial_weights
from oracle import data_oracle as do
from oracle import model_oracle as mo

DEF = synthetic.DEFAULT_MIXEDNET_FLAGS


def test_argparse_defaults_match_reference_and_do_not_construct():
    p = argparse.ArgumentParser()
    mixednet.model_parameters(p)
    flags = p.parse_args([])
    for k, v in mo.MIXEDNET_DEFAULTS.items():
        assert str(getattr(flags, k)) == str(v), k
    with pytest.raises(ValueError, match="same length"):
        lay.MixedNetLayout(flags, 194)  # five residual entries vs four blocks (mixednet.py:52-57,298-305)
    flags.residual_connection = "0,0,0,0"
    L = lay.MixedNetLayout(flags, 194)
    assert L.keras_param_counts() == (22561, 22177) and L.n_params == 22177 and L.n_state == 384
    assert mixednet.spectrogram_slices_dropped(flags) == 46


def test_unsupported_options_raise_loudly():
    for k, v in (("residual_connection", "0,1,0,0"), ("repeat_in_block", "1,2,1,1"), ("spatial_attention", 1), ("pooled", 1), ("first_conv_filters", 0)):
        with pytest.raises(NotImplementedError):
            lay.MixedNetLayout(dict(DEF, **{k: v}), 194)


def test_pack_unpack_roundtrip_and_oracle_order():
    L = lay.MixedNetLayout(DEF, 194)
    om = mo.OracleModel("mixednet", DEF, 194, seed=1)
    ws = [w + 0.01 * (i + 1) for i, w in enumerate(om.get_weights())]
    assert [tuple(s) for _, s, _ in L.keras_vars] == [w.shape for w in ws]
    p, s = L.pack(ws)
    back = L.unpack(p, s)
    for a, b in zip(ws, back):
        np.testing.assert_array_equal(a.astype(np.float32), b)
    assert L.grad_mask().min() == 1.0


def test_mixconv_fusion_matches_oracle_right_alignment():
    """Multi-kernel MixConv groups fused into one [K_last, C] depthwise with zero leading taps give the
    same block output as the reference's split / StridedDrop / concat (checked through the oracle)."""
    flags = dict(DEF, mixconv_kernel_sizes="[5],[7,11],[9,15],[23]", pointwise_filters="48,48,48,48")
    T = 110
    L = lay.MixedNetLayout(flags, T)
    assert [b.k for b in L.blocks] == [5, 11, 15, 23] and L.blocks[1].group_channels == (24, 24)
    om = mo.OracleModel("mixednet", flags, T, seed=2)
    assert [tuple(s) for _, s, _ in L.keras_vars] == [w.shape for w in om.get_weights()]
    ws = om.get_weights()
    p, s = L.pack(ws)
    m = L.grad_mask()
    assert m.sum() == L.keras_param_counts()[1]
    # fused single-kernel model with the packed weights == multi-kernel oracle
    fused_flags = dict(flags, mixconv_kernel_sizes="[5],[11],[15],[23]")
    Lf = lay.MixedNetLayout(fused_flags, T)
    omf = mo.OracleModel("mixednet", fused_flags, T, seed=0)
    omf.set_weights(Lf.unpack(p, s))
    x = np.random.default_rng(0).random((2, T, 40)) * 5
    np.testing.assert_allclose(omf.predict(x, True), om.predict(x, True), rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        lay.MixedNetLayout(dict(DEF, mixconv_kernel_sizes="[5],[11,7],[13],[21]"), 194)


def test_initial_weights_follow_keras_defaults():
    L = lay.MixedNetLayout(DEF, 194)
    ws = initial_weights(L, seed=0)
    for (name, shape, _), w in zip(L.keras_vars, ws):
        assert w.shape == tuple(shape)
        if name.endswith(("gamma", "moving_variance")):
            assert (w == 1).all()
        elif name.endswith(("bias", "beta", "moving_mean")):
            assert (w == 0).all()
    k = ws[0]
    assert np.abs(k).max() <= np.sqrt(6.0 / (3 * 40 + 3 * 32)) + 1e-6


def test_load_config_shape_derivation(tmp_path):
    cfg = dict(window_step_ms=10, train_dir=str(tmp_path / "m"), features=[], training_steps=[10], batch_size=8, clip_duration_ms=1500,
               eval_step_interval=5, target_minimization=0.9, minimization_metric=None, maximization_metric="average_viable_recall")
    f = tmp_path / "c.yaml"
    f.write_text(yaml.dump(cfg))
    flags = model_train_eval.build_parser().parse_args(["--training_config", str(f), "mixednet", "--residual_connection", "0,0,0,0"])
    c = model_train_eval.load_config(flags, mixednet)
    assert (c["spectrogram_length_final_layer"], c["spectrogram_length"], c["stride"]) == (148, 194, 1)
    assert c["training_input_shape"] == (194, 40) and c["summaries_dir"].endswith("logs/")
    assert (c["spectrogram_length_final_layer"], c["spectrogram_length"]) == mo.spectrogram_length(1500, 10, 1, 46)


class _FakeModel:
    """Duck-typed model recording what the loop does."""

    def __init__(self):
        self.optimizer = types.SimpleNamespace(learning_rate=types.SimpleNamespace(assign=self._lr))
        self.lrs, self.weights_seen, self.saved, self.resets, self.evals = [], [], [], 0, 0

    def _lr(self, v):
        self.lrs.append(v)

    def compile(self, **k):
        pass

    def make_train_function(self):
        pass

    def train_on_batch(self, x, y, sample_weight=None):
        self.weights_seen.append(np.asarray(sample_weight).copy())
        return [0.5, 0.9, 0.8, 0.7, None, None, None, None, 0.95, 0.4]

    def reset_metrics(self):
        self.resets += 1

    def evaluate(self, x, y, **k):
        self.evals += 1
        c = types.SimpleNamespace
        tp = np.linspace(10, 0, 101)
        return dict(accuracy=0.9, recall=0.8, precision=0.7, auc=0.9, loss=0.3, tp=c(numpy=lambda: tp), fp=c(numpy=lambda: np.zeros(101)),
                    tn=c(numpy=lambda: np.zeros(101)), fn=c(numpy=lambda: 10 - tp))

    def save_weights(self, path):
        self.saved.append(os.path.basename(path))


class _FakeData:
    def __init__(self):
        self.policies = []

    def get_data(self, mode, batch_size, features_length, truncation_strategy="default", augmentation_policy=None):
        if mode == "training":
            self.policies.append(dict(augmentation_policy))
            y = np.array([1.0, 0.0, 1.0, 0.0])
            return np.zeros((4, features_length, 40), np.float32), y, np.array([1.0, 2.0, 3.0, 4.0])
        return np.zeros((3, features_length, 40), np.float32), np.array([1.0, 0.0, 1.0]), np.ones(3)

    def get_mode_size(self, mode):
        return 0

    def get_mode_duration(self, mode):
        return 0.0


def test_train_loop_schedule_and_weights(tmp_path):
    config = dict(train_dir=str(tmp_path / "run"), summaries_dir=str(tmp_path / "run" / "logs"), batch_size=4, spectrogram_length=50,
                  training_steps=[2, 3], learning_rates=[0.01, 0.001], time_mask_max_size=[5, 0], time_mask_count=[2],
                  positive_class_weight=[1.0, 2.0], negative_class_weight=[20.0], eval_step_interval=2, target_minimization=0.9,
                  minimization_metric=None, maximization_metric="accuracy")
    m, d = _FakeModel(), _FakeData()
    out = tr.train(m, config, d, verbose=False)
    assert m.lrs == [0.01, 0.01, 0.001, 0.001, 0.001]                       # piecewise constant by cumulative steps
    assert [p["time_mask_max_size"] for p in d.policies] == [5, 5, 0, 0, 0]
    assert [p["time_mask_count"] for p in d.policies] == [2] * 5             # padded with the last entry
    # the default reading of train.py:288-293 is the reference's arithmetic (keras_last_axis): penalty_j * mean_i cw(y_i)
    np.testing.assert_allclose(m.weights_seen[0], np.array([1.0, 2.0, 3.0, 4.0]) * np.mean([1.0, 20.0, 1.0, 20.0]))
    np.testing.assert_allclose(m.weights_seen[4], np.array([1.0, 2.0, 3.0, 4.0]) * np.mean([2.0, 20.0, 2.0, 20.0]))
    m2, d2 = _FakeModel(), _FakeData()
    tr.train(m2, dict(config, train_dir=str(tmp_path / "run2"), summaries_dir=str(tmp_path / "run2" / "logs"), sample_weight_broadcast="per_sample"),
             d2, verbose=False)
    np.testing.assert_array_equal(m2.weights_seen[0], [1.0, 40.0, 3.0, 80.0])  # opt-in: penalty * class weight per sample
    np.testing.assert_array_equal(m2.weights_seen[4], [2.0, 40.0, 6.0, 80.0])
    assert m.evals == 3 and "best_weights.weights.h5" in m.saved and m.saved.count("last_weights.weights.h5") == 4
    assert "100000000_weights_2.weights.h5" in m.saved                      # previous best (10000) embedded in the name
    assert out["best_maximization"] == 0.9


def test_sample_weight_broadcast_modes():
    """train.py:288-293's [B,B] weight matrix W[i,j] = penalty_j * cw(y_i): the two readings of Keras' reduction are its
    column / row means, "per_sample" its diagonal (the intended per-sample weight); the vector form (combine_weights) agrees
    with the matrix form; per_sample equals keras_last_axis when the class weights are uniform and keras_first_axis when the
    penalties are.  ONE default for both entry points (round 6): the package loop's vector form and the drop-in
    train_on_batch's matrix form reduce a non-uniform batch to the same weights - the reference's arithmetic."""
    from microwakeword_amd.model import Model, combine_weights
    rng = np.random.default_rng(0)
    B = 7
    y = (rng.random(B) < 0.5).astype(np.float32)
    pen = rng.choice([0.5, 1.0, 2.0], size=B)
    cw = np.where(y > 0.5, 3.0, 0.25)
    W = pen[None, :] * cw[:, None]            # what the reference's broadcast produces
    m = Model.__new__(Model)                  # the reduction is host arithmetic: no engine needed
    for mode, want in (("per_sample", pen * cw), ("keras_last_axis", pen * cw.mean()), ("keras_first_axis", cw * pen.mean())):
        m.sample_weight_broadcast = mode
        np.testing.assert_allclose(m._per_sample_weights(W, B), want, rtol=1e-6)
        np.testing.assert_allclose(combine_weights(pen, y, 0.25, 3.0, mode), want, rtol=1e-6)
    np.testing.assert_allclose(combine_weights(pen, y, 2.0, 2.0, "keras_last_axis"), combine_weights(pen, y, 2.0, 2.0, "per_sample"))
    np.testing.assert_allclose(combine_weights(np.ones(B), y, 0.25, 3.0, "keras_first_axis"), combine_weights(np.ones(B), y, 0.25, 3.0, "per_sample"))
    with pytest.raises(ValueError):
        combine_weights(pen, y, 1.0, 1.0, "diagonal")
    # both entry points agree on a batch where class AND penalty weights vary: a [B,B] matrix through the drop-in
    # train_on_batch (its caller is the reference's train.py) and the vector form of the package's own loop
    from microwakeword_amd import model as model_mod
    assert model_mod.MATRIX_WEIGHT_BROADCAST == model_mod.DEFAULT_WEIGHT_BROADCAST == "keras_last_axis"
    assert model_mod.readings_differ(pen, y, 0.25, 3.0) and not model_mod.readings_differ(pen, y, 2.0, 2.0)
    fresh = Model.__new__(Model)
    fresh.sample_weight_broadcast = model_mod.MATRIX_WEIGHT_BROADCAST          # what Model.__init__ sets
    np.testing.assert_allclose(fresh._per_sample_weights(W, B), combine_weights(pen, y, 0.25, 3.0), rtol=1e-6)   # default mode of the vector form
    np.testing.assert_allclose(fresh._per_sample_weights(W, B), pen * cw.mean(), rtol=1e-6)
    np.testing.assert_allclose(fresh._per_sample_weights(pen * cw, B), pen * cw, rtol=1e-6)   # a plain vector is taken as it is


def test_bench_global_batch_is_split_or_refused():
    """bench.py --global-batch (strong scaling, BASELINE configs[4]): 4096 / N windows per GPU, refused when N does not divide it."""
    import bench
    assert bench.per_gpu_batch(1024, 0, 8) == 1024
    assert [bench.per_gpu_batch(1024, 4096, n) for n in (1, 2, 4, 8)] == [4096, 2048, 1024, 512]
    with pytest.raises(SystemExit, match="not divisible"):
        bench.per_gpu_batch(1024, 4096, 3)


def test_best_model_rule():
    f = tr._is_better
    assert f(0.5, 0.1, 10000, 0.0, 0.9)        # first time under target
    assert f(0.5, 0.3, 0.5, 0.2, 0.9)          # under target, accuracy improved
    assert not f(0.5, 0.1, 0.5, 0.2, 0.9)
    assert f(1.5, 0.0, 2.0, 0.5, 0.9)          # above target but decreased
    assert not f(2.5, 0.9, 2.0, 0.5, 0.9)


def test_synthetic_generators_agree_with_oracle_copy():
    a = synthetic.synthetic_stores(16, 1234)
    b = do.synthetic_stores(16, 1234)
    for sa, sb in zip(a, b):
        for x, y in zip(sa, sb):
            np.testing.assert_array_equal(x, y)


def test_inception_flags_layout_and_oracle_order(tmp_path):
    from microwakeword_amd import inception
    p = argparse.ArgumentParser()
    inception.model_parameters(p)
    flags = p.parse_args([])
    for k, v in mo.INCEPTION_DEFAULTS.items():
        assert str(getattr(flags, k)) == str(v), k
    assert inception.spectrogram_slices_dropped(flags) == mo.inception_slices_dropped(flags) == 28
    om = mo.OracleModel("inception", vars(flags), 194, seed=1)
    ws = [w + 0.01 * (i + 1) for i, w in enumerate(om.get_weights())]
    for fuse in (False, True):
        L = lay.InceptionLayout(flags, 194, fuse_heads=fuse)
        assert [n for n, _, _ in L.keras_vars] == [v.name for v in om.vars]
        assert [tuple(s) for _, s, _ in L.keras_vars] == [v.value.shape for v in om.vars]
        assert L.keras_param_counts() == om.n_params()
        assert (L.t_last, L.c_last) == (166, 16) and len(L.ops) == (16 if fuse else 22)
        # StridedDrop alignment of the reduce conv: branch1 / branch2 lose their leading 8 / 4 frames
        red = L.ops[5 if fuse else 7]
        assert red["drop"] == [8, 4, 0] and red["cin"] == 30
        if fuse:   # the three 1x1 branch heads run as one 24->30 convolution; consumers name their slices
            assert L.ops[1]["filters"] == 30 and red["src"] == [1, 2, 4] and red["slice"][0] == (0, 10)
            assert L.ops[2]["src"] == [1] and L.ops[2]["slice"] == [(10, 10)] and L.ops[3]["slice"] == [(20, 10)]
            k = ws[[n for n, _, _ in L.keras_vars].index("i0.b2a.kernel")]
            off = dict((n, sum(m for _, m in L.segments()[:i])) for i, (n, _) in enumerate(L.segments()))["i0.b1+i0.b2a+i0.b3a.kernel"]
            fused = L.pack(ws)[0][off:off + 24 * 30].reshape(24, 30)
            np.testing.assert_array_equal(fused[:, 10:20], k[0, 0].astype(np.float32))
        else:
            assert red["src"] == [1, 4, 6]   # native op order: the three heads first, then b2b, b3b, b3c
        pv, sv = L.pack(ws)
        assert pv.size == L.n_params and sv.size == L.n_state
        for a, b in zip(ws, L.unpack(pv, sv)):
            np.testing.assert_array_equal(a.astype(np.float32), b)
    # sub-spectral groups inside a block keep the heads separate (their BN slots interleave per branch)
    assert len(lay.InceptionLayout(dict(mo.INCEPTION_DEFAULTS, cnn2_subspectral_groups="2,1,1"), 194).ops) == 18
    # sub-spectral groups must divide the filters (sub_spectral_normalization.py:41-45)
    with pytest.raises(ValueError, match="divisible"):
        lay.InceptionLayout(dict(mo.INCEPTION_DEFAULTS, cnn1_subspectral_groups="5"), 194)
    with pytest.raises(ValueError, match="too short"):
        lay.InceptionLayout(mo.INCEPTION_DEFAULTS, 20)
    # config derivation through the CLI surface
    cfg = dict(window_step_ms=10, train_dir=str(tmp_path / "m"), features=[], training_steps=[10], batch_size=8, clip_duration_ms=1500,
               eval_step_interval=5, target_minimization=0.9, minimization_metric=None, maximization_metric="average_viable_recall")
    f = tmp_path / "c.yaml"
    f.write_text(yaml.dump(cfg))
    fl = model_train_eval.build_parser().parse_args(["--training_config", str(f), "inception"])
    c = model_train_eval.load_config(fl, inception)
    assert (c["spectrogram_length_final_layer"], c["spectrogram_length"]) == (148, 176)


def test_package_and_bench_main_path_never_import_the_oracle():
    """oracle/ is test infrastructure: the package must not reference it, bench.py only inside the cpu_baseline leg
    (cpu_baseline() and the _reference_loader() it calls)."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "microwakeword_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, f)).read())
            for node in ast.walk(tree):
                names = [a.name for a in node.names] if isinstance(node, ast.Import) else \
                    ([node.module or ""] if isinstance(node, ast.ImportFrom) else [])
                assert not any(n.split(".")[0] == "oracle" for n in names), (f, names)
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                assert fn.name in ("cpu_baseline", "_reference_loader"), fn.name   # the baseline leg and its loader helper
    assert synthetic.DEFAULT_INCEPTION_FLAGS == mo.INCEPTION_DEFAULTS


def test_ragged_store_reader_refuses_anything_but_the_exact_layout(tmp_path):
    """microwakeword_amd/ragged.py: the recalled mmap_ninja layout is accepted only if every invariant holds; the
    flat export written by tools/export_ragged_to_flat.py (on a machine with the real library) takes precedence."""
    import shutil

    from microwakeword_amd.ragged import FLAT_EXPORT, RaggedStoreReader, flatten_store, write_ragged_store
    rng = np.random.default_rng(0)
    samples = [rng.integers(0, 667, size=(t, 40)).astype(np.uint16) for t in (150, 7, 400)]
    d = tmp_path / "a_mmap"
    write_ragged_store(str(d), samples)
    r = RaggedStoreReader(str(d))
    assert len(r) == 3 and r.dtype == np.uint16
    for i, s in enumerate(samples):
        np.testing.assert_array_equal(r[i], s)
    flat, starts, lens = flatten_store(r)
    assert lens.tolist() == [150, 7, 400] and starts.tolist() == [0, 6000, 6280] and flat.size == 557 * 40

    def broken(name, mutate):
        b = tmp_path / name
        shutil.copytree(d, b)
        mutate(b)
        with pytest.raises(ValueError, match="unrecognised ragged store layout"):
            RaggedStoreReader(str(b))

    broken("no_shapes", lambda b: shutil.rmtree(b / "shapes"))
    broken("fortran", lambda b: (b / "data" / "order.ninja").write_text("F"))
    broken("gap", lambda b: np.array([0, 6000, 6400], np.int64).tofile(b / "starts" / "data.ninja"))
    broken("short_data", lambda b: ((b / "data" / "shape.ninja").write_text("22000"), np.zeros(22000, np.uint16).tofile(b / "data" / "data.ninja")))
    broken("wrong_bins", lambda b: np.array([150, 41, 7, 40, 400, 40], np.int64).tofile(b / "shapes" / "data.ninja"))
    broken("float64", lambda b: ((b / "data" / "dtype.ninja").write_text("float64"), np.zeros(22280, np.float64).tofile(b / "data" / "data.ninja")))

    # the verified interchange wins over whatever else is in the directory
    e = tmp_path / "b_mmap"
    e.mkdir()
    sizes = np.array([150, 7, 400]) * 40
    np.savez(e / FLAT_EXPORT, data=np.concatenate([s.reshape(-1) for s in samples]), starts=np.cumsum(sizes) - sizes, lens=np.array([150, 7, 400], np.int32))
    r2 = RaggedStoreReader(str(e))
    for i, s in enumerate(samples):
        np.testing.assert_array_equal(r2[i], s)


def test_bench_refuses_to_report_a_smaller_job_under_a_bigger_label():
    """`python bench.py --gpus N` outside a launcher becomes the launcher of N ranks; where fewer GPUs are visible it must
    fail loudly instead of printing a line with another n_gpus (round-1 verdict: `--gpus 8` ran one rank), and without the
    HIP library / a GPU the single-rank run must fail as well (no CPU fallback behind the product path)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "GPU(s) are visible" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_traffic_figure_is_tied_to_the_library_it_was_measured_on(tmp_path):
    """roofline.traffic comes from a committed PMC summary; bench.py reports it only when the summary's stamp
    (`# library sha256_16=...`, written by tools/pmc_summary.py) is the sha of the library this process loads."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib = tmp_path / "libfake.so"
    lib.write_bytes(b"library build A")
    sha = bench.library_sha16(str(lib))
    body = ("## counters pmc3\nbwd_block_kernel<48, 48, 21, true>   n=8   us=52.0 FETCH_SIZE=3.2e+04\n"
            "## counters pmc4\nbwd_block_kernel<48, 48, 21, true>   n=8   us=52.0 WRITE_SIZE=4e+04\n")
    good = tmp_path / "good.txt"
    good.write_text("# library sha256_16=%s\n%s" % (sha, body))
    nbytes, src = bench.pmc_traffic("bwd_block4", "mixednet", path=str(good), library=str(lib))
    assert nbytes == int(2 * 3.2e4 * 1024 + 4e4 * 1024) and sha in src
    lib.write_bytes(b"library build B")                       # the library moved on, the profile did not
    nbytes, src = bench.pmc_traffic("bwd_block4", "mixednet", path=str(good), library=str(lib))
    assert nbytes is None and src.startswith("stale: profile sha %s != library sha" % sha)
    # Inception: the two stem launches have kernels of their own; the other ops share symbols and report nothing
    lib.write_bytes(b"library build A")
    inc = tmp_path / "inc.txt"
    inc.write_text("# library sha256_16=%s\n## counters pmc3\ngconv_wgrad_xg_kernel<24, GShape<5, 1, 40, 40, 0, 0, 0, 0> > n=8 us=50 FETCH_SIZE=2.6e+04\n"
                   "## counters pmc4\ngconv_wgrad_xg_kernel<24, GShape<5, 1, 40, 40, 0, 0, 0, 0> > n=8 us=50 WRITE_SIZE=9600\n" % sha)
    nbytes, src = bench.pmc_traffic("conv_wgrad1", "inception", path=str(inc), library=str(lib))
    assert nbytes == int(2 * 2.6e4 * 1024 + 9600 * 1024) and sha in src
    assert bench.pmc_traffic("conv_bwd5", "inception", path=str(inc), library=str(lib)) == (None, None)
    assert bench.pmc_traffic("bwd_block4", "notebook", path=str(inc), library=str(lib)) == (None, None)
    lib.write_bytes(b"library build B")
    unstamped = tmp_path / "old.txt"
    unstamped.write_text(body)                                 # a summary from before the stamp existed
    nbytes, src = bench.pmc_traffic("bwd_block4", "mixednet", path=str(unstamped), library=str(lib))
    assert nbytes is None and src.startswith("stale: profile sha None")
    # the summariser writes the stamp of the library in the tree as its first line
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py")], capture_output=True, text=True)
    first = r.stdout.splitlines()[0]
    assert first.startswith("# library sha256_16=")
    if os.path.isfile(bench.LIBRARY):
        from microwakeword_amd import build_native
        assert first == "# library sha256_16=%s source_sha16=%s" % (bench.library_sha16(), build_native.library_source_sha16() or "unknown")
    # a library REBUILT from the same source set (hipcc output is not bit-identical across build paths) is accepted by its
    # source stamp, and the source string says that it was
    rebuilt = tmp_path / "librebuilt.so"
    rebuilt.write_bytes(b"other bytes, same kernels: mww-hip 0.1 (gfx950) src=0123456789abcdef\0")
    stamped = tmp_path / "stamped.txt"
    stamped.write_text("# library sha256_16=%s source_sha16=0123456789abcdef\n%s" % ("f" * 16, body))
    nbytes, src = bench.pmc_traffic("bwd_block4", "mixednet", path=str(stamped), library=str(rebuilt))
    assert nbytes == int(2 * 3.2e4 * 1024 + 4e4 * 1024) and "same source set 0123456789abcdef" in src
    stamped.write_text("# library sha256_16=%s source_sha16=fedcba9876543210\n%s" % ("f" * 16, body))
    assert bench.pmc_traffic("bwd_block4", "mixednet", path=str(stamped), library=str(rebuilt))[0] is None


def test_train_model_claims_the_directory_like_the_reference(tmp_path, monkeypatch):
    """model_train_eval.py:99-128: a fresh train_dir is created, an existing one is an error unless restore_checkpoint."""
    calls = []
    monkeypatch.setattr(model_train_eval.train_mod, "train", lambda model, config, dp: calls.append("train") or "done")
    monkeypatch.setattr(model_train_eval, "save_model_summary", lambda model, path, file_name="model_summary.txt": calls.append("summary"))
    cfg = {"train_dir": str(tmp_path / "run"), "summaries_dir": str(tmp_path / "run" / "logs"), "features": []}
    assert model_train_eval.train_model(cfg, object(), object(), 0) == "done"
    assert os.path.isfile(os.path.join(cfg["train_dir"], "training_config.yaml")) and os.path.isdir(cfg["summaries_dir"])
    with pytest.raises(ValueError, match="model already exists"):
        model_train_eval.train_model(cfg, object(), object(), 0)
    assert model_train_eval.train_model(cfg, object(), object(), 1) == "done"
    assert calls == ["summary", "train", "summary", "train"]


def test_source_stamp_of_the_library(tmp_path, monkeypatch):
    """build_native: the stamp compiled into the library (mww_version()) is the sha256 of csrc/* + include/mww.h;
    __graft_entry__.build() rebuilds exactly when the library's stamp differs from the tree's."""
    import importlib
    from microwakeword_amd import build_native as bn
    tree = bn.source_sha16()
    assert len(tree) == 16 and tree == bn.source_sha16()
    fake = tmp_path / "lib.so"
    fake.write_bytes(b"\x7fELF....mww-hip 0.1 (gfx950) src=%s\0...." % tree.encode())
    assert bn.library_source_sha16(str(fake)) == tree
    fake.write_bytes(b"\x7fELF....mww-hip 0.1 (gfx950)\0....")     # a library from before the stamp
    assert bn.library_source_sha16(str(fake)) is None
    assert bn.library_source_sha16(str(tmp_path / "missing.so")) is None
    # both branches of build(): up to date -> nothing compiled; stamp differs -> build_library is called
    ge = importlib.import_module("__graft_entry__")
    built = []
    monkeypatch.setattr(bn, "build_library", lambda out, **kw: built.append(out) or 1)
    state = {"sha": tree}
    monkeypatch.setattr(bn, "library_source_sha16", lambda path=bn.LIB: state["sha"])
    monkeypatch.setattr(bn, "library_sha16", lambda path=bn.LIB: "0" * 16)
    monkeypatch.setattr(ge.os.path, "getmtime", lambda p: 0.0)
    import microwakeword_amd.native as native

    class _NL:
        def __init__(self, path): pass
        def version(self): return "stub"
    monkeypatch.setattr(native, "NativeLib", _NL)
    monkeypatch.setattr("builtins.open", lambda *a, **k: (_ for _ in ()).throw(OSError("no build_info in this test")) if str(a[0]).endswith("build_info.json") else open.__wrapped__(*a, **k) if hasattr(open, "__wrapped__") else __import__("io").open(*a, **k))
    ge.build()
    assert built == []
    state["sha"] = "f" * 16
    with pytest.raises(RuntimeError, match="does not carry"):   # the stubbed build changes nothing, which build() must notice
        ge.build()
    assert built == [ge.LIB]


def test_kernel_family_of_a_grid_of_flag_sets():
    """Which MixedNet flag sets land on the specialised MFMA block kernels and which on the graph kernels (2.5x the step
    time): answered from the build-time shape table through the C ABI, no GPU (mww_block_kernels_cover)."""
    from microwakeword_amd import build_native, native
    if not os.path.isfile(build_native.LIB):
        pytest.skip("libmww_hip.so not built")
    nl = native.NativeLib(build_native.LIB)
    base = dict(mo_defaults(), residual_connection="0,0,0,0")

    def fam(T=194, **kw):
        return mixednet.kernel_family(dict(base, **kw), T, lib=nl)[0]

    # the two documented topologies and everything between them
    assert fam() == "block"
    assert fam(T=204, first_conv_kernel_size=5, stride=3, pointwise_filters="64,64,64,64", mixconv_kernel_sizes="[5],[7,11],[9,15],[23]") == "block"
    for widths in ("32,32,32,32", "48,48,48,48", "64,64,64,64", "32,48,64,64", "64,48,32,32", "48,64,48,64"):
        for ks in ("[5],[9],[13],[21]", "[3],[7],[17],[19]", "[7],[11,15],[3,5,23],[9]", "[5],[5],[5],[5]"):
            for k1, st in ((3, 1), (5, 1), (3, 2), (5, 3)):
                assert fam(T=230, pointwise_filters=widths, mixconv_kernel_sizes=ks, first_conv_kernel_size=k1, stride=st) == "block", (widths, ks, k1, st)
    # 2..6 blocks
    for nb in (2, 3, 5, 6):
        lists = dict(pointwise_filters=",".join(["48"] * nb), mixconv_kernel_sizes=",".join(["[5]"] + ["[9]"] * (nb - 1)),
                     repeat_in_block=",".join(["1"] * nb), residual_connection=",".join(["0"] * nb))
        assert fam(**lists) == "block", nb
    one = dict(pointwise_filters="48", mixconv_kernel_sizes="[5]", repeat_in_block="1", residual_connection="0")
    assert fam(**one) == "graph"
    # off the table: other widths, even / long kernels, first-block kernels beyond 7, another first conv, the options
    assert fam(pointwise_filters="40,40,40,40") == "graph"
    assert fam(mixconv_kernel_sizes="[5],[10],[13],[21]") == "graph"
    assert fam(T=260, mixconv_kernel_sizes="[5],[9],[13],[25]") == "graph"
    assert fam(mixconv_kernel_sizes="[9],[9],[13],[21]") == "graph"
    assert fam(first_conv_filters=16) == "graph" and fam(first_conv_kernel_size=7) == "graph" and fam(stride=4) == "graph"
    assert fam(residual_connection="0,1,0,0") == "graph" and fam(repeat_in_block="1,2,1,1") == "graph"
    assert fam(spatial_attention=1) == "graph" and fam(pooled=1) == "graph" and fam(first_conv_filters=0) == "graph"
    # the classifier head holds a window's final frames in registers: 24 frames per frame group = 768 / 504 / 384 frames at
    # 32 / 48 / 64 channels; one more lands on the graph kernels at creation instead of failing at the first forward
    # (tools/gpu_x6_fuzz.py case 460: 64 channels x 390 frames)
    dropped = 2 + 4 + 8 + 12 + 20
    for width, limit in ((32, 768), (48, 504), (64, 384)):
        pf = "48,48,48,%d" % width
        assert fam(T=limit + dropped, pointwise_filters=pf) == "block", width
        fam_, why = mixednet.kernel_family(dict(base, pointwise_filters=pf), limit + 1 + dropped, lib=nl)
        assert fam_ == "graph" and "final frames" in why, (width, why)
    # the bf16 modes exist for the documented topologies and their crosses only
    assert mixednet.kernel_family(base, 194, lib=nl, bf16=True)[0] == "block"
    assert mixednet.kernel_family(dict(base, pointwise_filters="32,32,32,32"), 194, lib=nl, bf16=True)[0] == "graph"


def mo_defaults():
    from oracle import model_oracle as mo
    return dict(mo.MIXEDNET_DEFAULTS)
