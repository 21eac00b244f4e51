"""the reference's own train loop drives this package's objects.

``oracle/ref_train_shim.py`` executes ``/root/reference/microwakeword/train.py`` unmodified (stand-ins for the ``tensorflow`` names it
touches: constructors, ``tf_decorator.unwrap``, ``tf.train.Checkpoint``, ``tf.summary``) and its ``train(model, config,
data_processor)`` is called with ``microwakeword_amd``'s ``Model`` and ``FeatureHandler`` on the host-emulated library.  Container
only (``/root/reference`` does not exist on the GPU box); the recorded call / return trace is the committed fixture that
``tests/test_engine_gpu.py::test_reference_train_loop_trace_replay`` replays on the MI355X."""
import copy
import json
import os
import random
import sys

import numpy as np
import pytest

import engine_checks as ec
import ref_train_replay as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _class_weights_per_step(cfg):
    out = []
    for n, neg, pos in zip(cfg["training_steps"], cfg["negative_class_weight"], cfg["positive_class_weight"]):
        out += [[float(neg), float(pos)]] * int(n)
    return out


def _run_reference_loop(emu_lib, tmp_path, steps=24, B=16, evaluations=3):
    import ref_train_shim as shim
    if not shim.available():
        pytest.skip("reference tree not present")
    ref = shim.load_reference_train_module()
    assert "tensorflow" not in sys.modules            # the stand-ins do not outlive the import
    cfg = rr.run_config(ec, tmp_path, steps=steps, B=B)
    cfg["eval_step_interval"] = steps // evaluations
    model, data = rr.make_objects(ec, emu_lib, cfg)
    trace = shim.Trace(model, data)
    trace.config_before = {k: copy.deepcopy(v) for k, v in cfg.items() if k != "features"}   # train.py:191-204 pads the lists in place
    ref.train(trace.model, cfg, trace.data)            # train.py:166-462, every line of it
    return ref, shim, cfg, model, data, trace


@pytest.fixture(scope="module")
def reference_run(emu_lib, tmp_path_factory):
    """ONE 24-step run of the reference's loop for the tests below (it takes a minute on the emulated kernels)."""
    return _run_reference_loop(emu_lib, tmp_path_factory.mktemp("refrun"))


@pytest.mark.reference
def test_reference_train_loop_drives_model_and_feature_handler(emu_lib, tmp_path, reference_run):
    ref, shim, cfg, model, data, trace = reference_run
    steps = int(np.sum(cfg["training_steps"]))
    run = cfg["train_dir"]
    # what train.py writes (the .weights.h5 names carry the documented .npz twin): last / best weights, the per-evaluation
    # snapshots, the checkpoint
    for f in ("last_weights.weights.h5.npz", "best_weights.weights.h5.npz", "restore/ckpt.weights.npz", "restore/ckpt.opt.npz"):
        assert os.path.isfile(os.path.join(run, f)), f
    assert len([f for f in os.listdir(os.path.join(run, "train")) if f.endswith(".weights.h5.npz")]) == 3
    # the calls of SURVEY §8(b)'s model row, in the order train.py makes them
    names = [(o, m) for o, m, *_ in trace.calls]
    assert names[:3] == [("model", "compile"), ("model", "make_train_function"), ("model", "setattr:train_function")]
    assert names.count(("model", "train_on_batch")) == steps and names.count(("model", "optimizer.learning_rate.assign")) == steps
    assert names.count(("data", "get_data")) == steps + 2 * 3 and names.count(("model", "evaluate")) == 2 * 3
    assert names.count(("model", "setattr:reset_metrics")) == 2 * 3          # swapped for a no-op and back, once per validation
    assert [s["lr"] for s in trace.steps] == [0.01] * (steps // 2) + [0.003] * (steps - steps // 2)
    # train_on_batch really was handed the [B,B] matrix (train.py:288-293)
    tob = [c for c in trace.calls if c[1] == "train_on_batch"][0]
    assert tob[3]["sample_weight"]["__placeholder__"] == "ndarray[16, 16]:float64"
    # the summaries train.py writes through tf.summary
    tags = {(os.path.basename(d), n) for d, n, *_ in shim.Summaries.records}
    assert ("train", "loss") in tags and ("validation", "average_viable_recall") in tags
    assert any("So far the best minimization quantity" in ln for ln in shim.LogCapture.lines)
    # the package's own loop (microwakeword_amd.train.train: device-resident batches, ``sample_weight_broadcast: keras_last_axis`` =
    # the reference's arithmetic for its [B,B] weights) reads the same five numbers after every step under the same seeds: the two
    # loops feed the same windows (host batch vs gathered in the kernel: bit-identical) and the same weights (column means of the
    # matrix vs penalty * mean(class weight): equal to float64 rounding)
    from microwakeword_amd import train as tr
    cfg_p = dict(rr.run_config(ec, tmp_path / "pkg"), sample_weight_broadcast="keras_last_axis", prefetch_batches=0, progress_interval_steps=1)
    model_p, data_p = rr.make_objects(ec, emu_lib, cfg_p)
    tp = shim.Trace(model_p, data_p)
    tr.train(tp.model, cfg_p, tp.data, verbose=True)
    assert len(tp.steps) == len(trace.steps) == steps
    np.testing.assert_allclose(np.array([s_["result"] for s_ in trace.steps]), np.array([s_["result"] for s_ in tp.steps]), rtol=0, atol=2e-6)
    for wa, wb in zip(model.get_weights(), model_p.get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=1e-6)
    model_p.engine.close()
    # restore is unconditional (train.py:232-233): a second call continues from the checkpoint's optimizer step
    model2, data2 = rr.make_objects(ec, emu_lib, cfg, seed=99)
    tr2 = shim.Trace(model2, data2)
    cfg2 = dict(cfg, training_steps=[2], learning_rates=[0.001], eval_step_interval=2)
    ref.train(tr2.model, cfg2, tr2.data)
    assert model2.engine.get_opt_state()[2] == steps + 2
    model2.engine.close()


@pytest.mark.reference
def test_reference_cli_main_runs_on_the_package_modules(emu_lib, tmp_path, monkeypatch):
    """The reference's OWN command line - ``model_train_eval.py`` executed as ``__main__`` (argparse, ``load_config``,
    ``input_data.FeatureHandler(config)`` built BEFORE the model, ``mixednet.model(flags, shape, batch_size)``, ``train_model`` ->
    its own ``train.train``; model_train_eval.py:45-128,277-439) - with this package's ``data`` / ``mixednet`` / ``inception`` modules
    bound to the names it imports: stores read from ``<features_dir>/<mode>/*_mmap`` on disk, the handler attaches to the model's
    engine at its first use, and the run leaves what the reference's run leaves."""
    import yaml

    import ref_train_shim as shim
    from microwakeword_amd import ragged
    if not shim.available():
        pytest.skip("reference tree not present")
    rng = np.random.default_rng(0)
    for prov, positive in (("wake", True), ("background", False)):
        for mode, n in (("training", 12), ("validation", 6), ("validation_ambient", 2)):
            if mode == "validation_ambient" and positive:
                continue
            lo, hi = (200, 260) if mode == "validation_ambient" else (62, 90)
            samples = []
            for _ in range(n):
                s_ = rng.integers(0, 200, size=(int(rng.integers(lo, hi)), 40)).astype(np.uint16)
                if positive:
                    s_[-30:-10, 8:24] += 400
                samples.append(s_)
            ragged.write_ragged_store(str(tmp_path / prov / mode / ("%s_mmap" % mode)), samples)
    cfg = dict(window_step_ms=10, train_dir=str(tmp_path / "trained"), clip_duration_ms=160, batch_size=4, training_steps=[4],
               learning_rates=[0.001], eval_step_interval=2, target_minimization=0.9, minimization_metric=None,
               maximization_metric="average_viable_recall", time_mask_max_size=[3], time_mask_count=[1], freq_mask_max_size=[3],
               freq_mask_count=[1], positive_class_weight=[1], negative_class_weight=[2],
               features=[dict(features_dir=str(tmp_path / "wake"), sampling_weight=1.0, penalty_weight=1.0, truth=True,
                              truncation_strategy="truncate_start", type="mmap"),
                         dict(features_dir=str(tmp_path / "background"), sampling_weight=2.0, penalty_weight=1.0, truth=False,
                              truncation_strategy="random", type="mmap")])
    (tmp_path / "cfg.yaml").write_text(yaml.dump(cfg))
    monkeypatch.setenv("MWW_HIP_LIB", emu_lib.path)       # the CLI builds its own engine: point it at the emulator build
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    random.seed(3)
    np.random.seed(3)
    # (--test_tflite_streaming_quantized defaults to 1: the TFLite export / evaluation stays with the reference and TensorFlow)
    argv = ["--training_config", str(tmp_path / "cfg.yaml"), "--verbosity", "ERROR", "--test_tflite_streaming_quantized", "0",
            "mixednet", "--residual_connection", "0,0,0,0"]
    shim.run_reference_cli(argv)
    run = tmp_path / "trained"
    for f in ("training_config.yaml", "model_summary.txt", "best_weights.weights.h5.npz", "last_weights.weights.h5.npz",
              "restore/ckpt.weights.npz", "restore/ckpt.opt.npz"):
        assert (run / f).exists(), f
    saved = yaml.load((run / "training_config.yaml").read_text(), yaml.Loader)
    assert saved["spectrogram_length"] == 60 and saved["spectrogram_length_final_layer"] == 14 and saved["training_input_shape"] == (60, 40)
    assert "Total params" in (run / "model_summary.txt").read_text()
    assert int(np.load(run / "restore" / "ckpt.opt.npz")["step"]) == 4
    with pytest.raises(ValueError, match="model already exists"):
        shim.run_reference_cli(argv)                                   # model_train_eval.py:113-120
    shim.run_reference_cli(argv[:2] + ["--restore_checkpoint", "1"] + argv[2:])
    assert int(np.load(run / "restore" / "ckpt.opt.npz")["step"]) == 8   # restored (train.py:232-233) + 4 new optimizer steps


@pytest.mark.reference
def test_reference_loop_learns_the_task(emu_lib, tmp_path):
    """Long enough for the BN moving averages (validation runs in inference mode): the reference's loop reaches >= 95 % validation
    accuracy on the separable task with this package's objects."""
    ref, shim, cfg, model, data, trace = _run_reference_loop(emu_lib, tmp_path, steps=72, B=8)
    val = trace.evals[-2]          # the last pass over the validation set (the ambient pass accumulates behind it)
    assert val["accuracy"] >= 0.95, val["accuracy"]
    model.engine.close()


@pytest.mark.reference
def test_reference_trace_fixture_is_current(reference_run):
    """The committed fixture is what the reference's loop does today on the emulated library (regenerate with
    ``MWW_WRITE_TRACE=1 python -m pytest tests/test_reference_train_loop.py -k fixture``)."""
    ref, shim, cfg, model, data, trace = reference_run
    fx = rr.trace_to_fixture(trace, dict(trace.config_before, train_dir=cfg["train_dir"]), _class_weights_per_step(cfg), None)
    if os.environ.get("MWW_WRITE_TRACE") == "1":
        with open(rr.FIXTURE, "w") as fh:
            json.dump(fx, fh, indent=0, separators=(",", ":"))
    have = rr.load_fixture()
    assert have["calls"] == json.loads(json.dumps(fx["calls"]))
    assert have["config"] == json.loads(json.dumps(fx["config"]))
    np.testing.assert_allclose(np.array([s["result"] for s in have["steps"]]), np.array([s["result"] for s in fx["steps"]]), atol=1e-6)


def test_reference_trace_replays_on_the_emulated_library(emu_lib, tmp_path):
    """The replayer used on the GPU box (tests/ref_train_replay.py) against the library the fixture was recorded on: the five
    numbers train.py reads after every step, and every evaluate result, come back as recorded."""
    import ref_train_shim as shim
    if shim.available() and os.environ.get("MWW_REPLAY_ALWAYS") != "1":
        pytest.skip("the reference tree is here: test_reference_trace_fixture_is_current re-records the trace live and compares it with the "
                    "fixture; the replayer itself runs in the -m gpu suite (MWW_REPLAY_ALWAYS=1 forces this test)")
    fx = rr.load_fixture()
    worst, evals, cfg = rr.replay(fx, ec, emu_lib, tmp_path)
    assert worst.max() <= 1e-6, worst
    for got, want in zip(evals, fx["evals"]):
        for k in ("accuracy", "recall", "precision", "auc", "loss"):
            assert abs(got[k] - want[k]) <= 1e-6, k
        for k in ("tp", "fp", "tn", "fn"):
            np.testing.assert_array_equal(got[k], np.array(want[k], np.float32))
    assert os.path.isfile(os.path.join(cfg["train_dir"], "best_weights.weights.h5.npz"))
