"""TEST INFRASTRUCTURE: the run of the reference's own ``microwakeword/train.py`` against this package's ``Model`` +
``FeatureHandler`` (``oracle/ref_train_shim.py``, build container only) leaves a call / return trace; this module turns it into a
JSON fixture (``tests/golden/ref_train_trace.json``) and replays it where ``/root/reference`` does not exist (the GPU box).

The fixture holds what the reference's loop DID — the ordered calls on the two objects with their simple arguments — and what it
GOT BACK on the host-emulated library — the five ``train_on_batch`` entries it reads per step, every ``evaluate`` result — plus the
configuration of the run.  The spectrogram batches themselves are not stored: the data half is pinned bit-exactly to the
reference (tests/golden/data_golden.npz), so the same seeds reproduce them.

Replay rules (how data flows between the recorded calls; this is the only knowledge of train.py the replayer has):
  data.get_data(...)               -> the current (x, y, w)
  model.train_on_batch             -> x, y.reshape(-1, 1), sample_weight = w * vectorize(class_weights.get)(y[:, None]) with the
                                      step's recorded class weights: the [B, B] matrix of train.py:288-293
  model.evaluate                   -> x, y.reshape(-1, 1) of the current batch, the recorded keyword arguments
  setattr:train_function           -> the attribute is assigned (train.py:227)
  setattr:reset_metrics            -> "noop" swaps in a no-op, "restore" puts the original back (train.py:89 swap_attribute)
  model.save_weights / checkpoint  -> the recorded path relative to the run directory
  anything else                    -> called with the recorded simple arguments
"""
import json
import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_train_trace.json")


def run_config(ec, tmp, B=16, T=60, steps=24):
    """The run both sides use: the separable task of engine_checks.learnable_config, two schedule phases, non-uniform class AND
    penalty weights (so the [B,B] weight matrix of train.py:288-293 matters), SpecAugment on, validation every third of the run."""
    cfg = dict(ec.learnable_config(T=T), train_dir=os.path.join(str(tmp), "run"), summaries_dir=os.path.join(str(tmp), "run", "logs"),
               batch_size=B, spectrogram_length=T, training_steps=[steps // 2, steps - steps // 2], learning_rates=[0.01, 0.003],
               time_mask_max_size=[3], time_mask_count=[1], freq_mask_max_size=[3], freq_mask_count=[1],
               positive_class_weight=[1.0, 2.0], negative_class_weight=[3.0, 1.0], eval_step_interval=steps // 3, target_minimization=0.9,
               minimization_metric=None, maximization_metric="accuracy", clip_duration_ms=600)
    cfg["features"][0]["penalty_weight"] = 2.0
    return cfg


def make_objects(ec, lib, cfg, seed=7):
    import random

    from microwakeword_amd import mixednet
    from microwakeword_amd.data import FeatureHandler
    random.seed(1)
    np.random.seed(1)
    model = mixednet.model(ec.DEF, (cfg["spectrogram_length"], 40), cfg["batch_size"], lib=lib, seed=seed, max_batch=64)
    return model, FeatureHandler(cfg, engine=model.engine)


def _rel(path, root):
    path = str(path)
    return os.path.relpath(path, root) if path.startswith(str(root)) else path


def trace_to_fixture(trace, cfg, class_weights_per_step, result):
    root = cfg["train_dir"]
    calls = []
    for obj, method, args, kwargs, ret in trace.calls:
        if method == "setattr:reset_metrics":
            calls.append([obj, method, ["noop" if "<lambda>" in str(args[0]) else "restore"], {}])   # (train.py:89's lambda, or the method back)
        else:
            args = [(_rel(a, root) if isinstance(a, str) else a) for a in args]
            calls.append([obj, method, args, kwargs])
    plain = {k: v for k, v in cfg.items() if k not in ("features", "train_dir", "summaries_dir")}
    evals = [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in e.items()} for e in trace.evals]
    return dict(config=plain, calls=calls, steps=trace.steps, class_weights=class_weights_per_step, evals=evals,
                result={k: float(v) for k, v in result.items()} if result else None)


def _simple(args, kwargs):
    """the recorded simple arguments; array / object arguments ({"__placeholder__": ...}) are supplied by the replay rules"""
    ph = lambda v: isinstance(v, dict) and "__placeholder__" in v
    return [a for a in args if not ph(a)], {k: v for k, v in kwargs.items() if not ph(v)}


def replay(fx, ec, lib, tmp, verbose=False):
    """Walks the recorded call list on fresh objects over ``lib``; returns (worst |step result - recorded| per entry, the
    evaluate results, the run's config)."""
    cfg = run_config(ec, tmp, B=fx["config"]["batch_size"], T=fx["config"]["spectrogram_length"], steps=int(np.sum(fx["config"]["training_steps"])))
    for k, v in fx["config"].items():
        assert cfg[k] == v, (k, cfg[k], v)
    model, data = make_objects(ec, lib, cfg)
    objs = {"model": model, "data": data}
    original_reset = model.reset_metrics
    cur = None
    step = 0
    worst = np.zeros(5)
    evals = []
    for obj, method, args, kwargs in fx["calls"]:
        o = objs[obj]
        a, kw = _simple(args, kwargs)
        if method == "setattr:reset_metrics":
            model.reset_metrics = (lambda: None) if args[0] == "noop" else original_reset
        elif method == "setattr:train_function":
            model.train_function = model.train_function   # train.py:227 stores the undecorated function back
        elif method == "get_data":
            cur = data.get_data(*a, **kw)
        elif method == "train_on_batch":
            x, y, w = cur
            y2 = y.reshape(-1, 1)
            neg, pos = fx["class_weights"][step]
            combined = w * np.vectorize({0: neg, 1: pos}.get)(y2)      # train.py:288-293: [B] * [B,1] -> [B,B]
            assert combined.shape == (y.size, y.size)
            out = model.train_on_batch(x, y2, sample_weight=combined)
            assert abs(float(model.optimizer.learning_rate.value) - fx["steps"][step]["lr"]) < 1e-12
            diff = np.abs(np.array([float(out[i]) for i in (1, 2, 3, 8, 9)]) - np.array(fx["steps"][step]["result"]))
            if verbose:
                print("step", step, diff)
            worst = np.maximum(worst, diff)
            step += 1
        elif method == "evaluate":
            x, y, _ = cur
            out = model.evaluate(x, y.reshape(-1, 1), **kw)
            evals.append({k: (np.array(v.numpy()) if hasattr(v, "numpy") else float(v)) for k, v in out.items()})
        elif method in ("save_weights", "save_optimizer_state", "load_weights", "load_optimizer_state"):
            p = os.path.join(cfg["train_dir"], a[0])
            if method.startswith("save"):
                os.makedirs(os.path.dirname(p), exist_ok=True)
            getattr(o, method)(p)
        elif method == "optimizer.learning_rate.assign":
            model.optimizer.learning_rate.assign(a[0])
        else:
            getattr(o, method)(*a, **kw)
    assert step == len(fx["steps"]) and len(evals) == len(fx["evals"])
    model.engine.close()
    return worst, evals, cfg


def load_fixture():
    with open(FIXTURE) as fh:
        return json.load(fh)
