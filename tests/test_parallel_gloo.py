"""world_size-2 gloo tests of the data-parallel host logic (sharding, RNG streams, gradient
all-reduce + averaged Adam, parameter broadcast).  The compute engine is stubbed with the CPU
oracle here (tests may use the oracle); the collectives and bookkeeping are the product code of
microwakeword_amd/parallel.py that runs unchanged over RCCL on the GPUs."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from microwakeword_amd import synthetic
from microwakeword_amd.parallel import DataParallel, shard_feature_handler
from oracle import model_oracle as mo

DEF = synthetic.DEFAULT_MIXEDNET_FLAGS
T = 60


class OracleEngine:
    """Minimal engine stand-in: flat gradient / parameter views + NO_APPLY step + apply."""

    def __init__(self, seed):
        self.om = mo.OracleModel("mixednet", DEF, T, seed=seed, dtype=torch.float64)
        self.tr = [v for v in self.om.vars if v.trainable]
        n = sum(v.value.size for v in self.tr)
        self.grad_view = torch.zeros(n, dtype=torch.float64)
        self.param_view = torch.zeros(n, dtype=torch.float64)
        self._pull()
        self.adam = mo.KerasAdam([(n,)], torch.float64)
        self.batch = None

    def _pull(self):
        self.param_view.copy_(torch.cat([torch.tensor(v.value, dtype=torch.float64).reshape(-1) for v in self.tr]))

    def _push(self):
        o = 0
        for v in self.tr:
            v.value = self.param_view[o:o + v.value.size].reshape(v.value.shape).numpy().astype(np.float32)
            o += v.value.size

    def synchronize(self):
        pass

    def train_step(self, B, lr, flags):
        assert flags & 1, "DP must request NO_APPLY"
        self._push()
        x, y, w = self.batch
        _, _, grads, _ = self.om.loss_and_grads(x, y, w)
        self.grad_view.copy_(torch.cat([grads[v.name].reshape(-1) for v in self.tr]))

    def apply_gradients(self, lr, scale):
        new = self.adam.apply([self.param_view.clone()], [self.grad_view * scale], lr)
        self.param_view.copy_(new[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    eng = OracleEngine(seed=100 + rank)               # ranks start from DIFFERENT weights
    dp = DataParallel(eng, eng.grad_view, eng.param_view)
    dp.broadcast_parameters(0)                        # ... and must agree after the broadcast
    rng = np.random.default_rng(7)
    xs = rng.random((2, 4, T, 40)) * 5
    ys = (rng.random((2, 4)) < 0.5).astype(np.float64)
    for step in range(2):
        eng.batch = (xs[rank] + step, ys[rank], np.ones(4))
        dp.train_step(4, 1e-3)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), eng.param_view.numpy())
    np.save(os.path.join(out_dir, "g%d.npy" % rank), eng.grad_view.numpy())
    dist.destroy_process_group()


def test_dp_two_ranks_allreduce_and_apply(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    np.testing.assert_array_equal(p0, p1)             # identical weights on every rank
    np.testing.assert_array_equal(np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy"))
    # single-process restatement: average of the two local gradients, Keras Adam, two steps
    ref = OracleEngine(seed=100)
    rng = np.random.default_rng(7)
    xs = rng.random((2, 4, T, 40)) * 5
    ys = (rng.random((2, 4)) < 0.5).astype(np.float64)
    for step in range(2):
        gs = []
        for r in range(2):
            ref.batch = (xs[r] + step, ys[r], np.ones(4))
            ref.train_step(4, 1e-3, 1)
            gs.append(ref.grad_view.clone())
        ref.grad_view.copy_(gs[0] + gs[1])
        ref.apply_gradients(1e-3, 0.5)
    np.testing.assert_allclose(p0, ref.param_view.numpy(), rtol=0, atol=1e-12)


class _Prov:
    def __init__(self, n):
        self.feature_sets = {"training": [(0, i) for i in range(n)]}
        self.stats = {"training": {"spectrogram_count": n}}


class _Handler:
    def __init__(self):
        self.feature_providers = [_Prov(10), _Prov(7)]
        self._sampler = "stale"
        self.private = None

    def use_private_rng(self):
        self.private = (random.random(), float(np.random.random()))


def test_sharding_partitions_every_provider_and_splits_rng_streams():
    seen, streams = [set(), set()], []
    for rank in range(3):
        h = _Handler()
        shard_feature_handler(h, rank, 3, seed=5)
        assert h._sampler is None
        for i, p in enumerate(h.feature_providers):
            mine = {s for _, s in p.feature_sets["training"]}
            assert not (mine & seen[i])
            seen[i] |= mine
        streams.append(h.private)
    assert seen[0] == set(range(10)) and seen[1] == set(range(7))
    assert len(set(streams)) == 3
    with pytest.raises(ValueError):
        h = _Handler()
        h.feature_providers = [_Prov(2)]
        shard_feature_handler(h, 2, 3, seed=0)


# ------------------------------------------------------------------------------------------ sync-BN
# Two ranks, each running the PRODUCT kernels (host-emulated, tests/hipemu) on half of a global batch with
# sync_bn=True, against the single-process oracle on the whole batch: W ranks x B/W windows must
# reproduce the single-device train step (loss normalisation, BN statistics, gradients, Adam, moving stats).
def _sync_worker(rank, world, port, out_dir, emu_path, kind, sync=True, buckets=2, tag="out"):
    import ctypes as C

    import engine_checks as ec
    from microwakeword_amd import native
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    lib = native.NativeLib(emu_path)
    Bl = 3
    z = np.load(os.path.join(out_dir, "inputs.npz"))
    if kind == "mixednet":
        om = ec.perturbed_oracle(T)
        lay, eng = ec.make_engine(lib, T, Bl, om)
    elif kind == "graph_mixednet":   # a MixedNet flag set that runs on the conv / depthwise graph kernels
        from microwakeword_amd.layout import GraphMixedNetLayout
        om = ec.perturbed_oracle(T, flags=ec.GRAPH_MIXEDNET)
        lay = GraphMixedNetLayout(ec.GRAPH_MIXEDNET, T)
        eng = native.Engine(lib=lib, **lay.engine_args(Bl))
        eng.set_grad_mask(lay.grad_mask())
        p0, s0 = lay.pack(om.get_weights())
        eng.set_params(p0)
        eng.set_bn_state(s0)
    else:
        om = ec.perturbed_inception_oracle(T, ec.INC)
        lay, eng = ec.make_inception_engine(lib, T, Bl, om, ec.INC)
        eng.set_dropout_mask(z["keep"][rank * Bl:(rank + 1) * Bl])

    def wrap(ptr, n):   # emulated "device" memory is host memory
        return torch.from_numpy(np.ctypeslib.as_array((C.c_float * n).from_address(ptr)))

    g = wrap(eng.device_ptr(native.BUF_GRADS), eng.n_params)
    p = wrap(eng.device_ptr(native.BUF_PARAMS), eng.n_params)
    dp = DataParallel(eng, g, p, None, sync_bn=sync, wrap=wrap, grad_buckets=buckets)
    assert dp.world == world
    eng.set_batch(z["x"][rank * Bl:(rank + 1) * Bl])
    eng.set_targets(z["y"][rank * Bl:(rank + 1) * Bl], z["w"][rank * Bl:(rank + 1) * Bl])
    dp.train_step(Bl, 1e-3)
    pr, _, loss = eng.read_outputs(Bl)
    np.savez(os.path.join(out_dir, "%s%d.npz" % (tag, rank)), grads=eng.get_grads(), params=eng.get_params(), state=eng.get_bn_state(),
             probs=pr, loss=loss, exchanges=np.asarray(dp.exchanges, np.int64))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["mixednet", "inception"])
def test_sync_bn_two_ranks_equal_single_device_step(tmp_path, kind):
    import conftest
    import engine_checks as ec
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W, Bl = 2, 3
    rng = np.random.default_rng(3)
    x = ec.synth_x(rng, W * Bl, T)
    y = (rng.random(W * Bl) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=W * Bl).astype(np.float32)
    if kind == "mixednet":
        om = ec.perturbed_oracle(T)
        from microwakeword_amd.layout import MixedNetLayout
        lay = MixedNetLayout(ec.DEF, T)
        keep = None
    else:
        om = ec.perturbed_inception_oracle(T, ec.INC)
        from microwakeword_amd.layout import InceptionLayout
        lay = InceptionLayout(ec.INC, T)
        keep = (rng.random((W * Bl, lay.t_last * lay.c_last)) >= ec.INC["dropout"]).astype(np.float32)
    np.savez(tmp_path / "inputs.npz", x=x, y=y, w=w, keep=keep if keep is not None else np.zeros(1))
    mp.spawn(_sync_worker, args=(W, _free_port(), str(tmp_path), emu, kind), nprocs=W, join=True)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(W)]
    # every rank holds the same reduced gradient, weights and moving statistics
    for k in ("grads", "params", "state"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
    lo, po, grads, _ = om.loss_and_grads(x, y, w, **({"dropout_mask": keep} if keep is not None else {}))
    if kind == "mixednet":
        gref = ec.oracle_grads_native_order(lay, om, grads)
    else:
        gref = lay.pack([grads[n].numpy().astype(np.float32) if kd == "param" else np.zeros(sh, np.float32)
                         for n, sh, kd in lay.keras_vars])[0]
    # rank losses are local means over B/W windows: their average is the global-batch loss
    assert abs(np.mean([float(o["loss"]) for o in outs]) - lo) <= 1e-5 * max(1.0, abs(lo))
    np.testing.assert_allclose(np.concatenate([o["probs"] for o in outs]), po, atol=1e-3)
    g = outs[0]["grads"] / W   # the buffer holds the SUM over ranks; Adam consumed it times 1/W
    scale = float(np.abs(gref).max())
    off = 0
    for name, n in lay.segments():
        a, r = g[off:off + n], gref[off:off + n]
        off += n
        tol = 2e-3 * scale if name.endswith("dw.bias") else 1e-3 * max(float(np.linalg.norm(r)), 1e-3 * scale * np.sqrt(n))
        assert np.linalg.norm(a - r) <= tol if not name.endswith("dw.bias") else np.abs(a - r).max() <= tol, (name, np.abs(a - r).max())
    om.train_step(x, y, w, 1e-3, **({"dropout_mask": keep} if keep is not None else {}))
    p_ref, s_ref = lay.pack(om.get_weights())
    well = np.abs(gref) > 1e-4 * scale
    assert np.abs(outs[0]["params"] - p_ref)[well].max() <= 0.05 * 1e-3
    assert np.abs(outs[0]["state"] - s_ref).max() <= 1e-5 * max(1.0, np.abs(s_ref).max())


def test_local_bn_two_ranks_with_product_kernels(tmp_path):
    """Throughput mode (what bench.py measures at N > 1): each rank normalises over its own half batch, the flat
    gradients are summed by the collective and Adam consumes the average.  Product kernels (host-emulated) on
    both ranks against two single-rank oracle passes averaged by hand."""
    import conftest
    import engine_checks as ec
    from microwakeword_amd.layout import MixedNetLayout
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W, Bl = 2, 3
    rng = np.random.default_rng(3)
    x = ec.synth_x(rng, W * Bl, T)
    y = (rng.random(W * Bl) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=W * Bl).astype(np.float32)
    np.savez(tmp_path / "inputs.npz", x=x, y=y, w=w, keep=np.zeros(1))
    mp.spawn(_sync_worker, args=(W, _free_port(), str(tmp_path), emu, "mixednet", False), nprocs=W, join=True)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(W)]
    # the gradient went in two buckets: [blocks 3, 4 + dense] deferred - it overlaps the backward kernels of
    # blocks 2 and 1 -, then the rest in stream order, then the flush that orders Adam behind the deferred bucket
    lay = MixedNetLayout(ec.DEF, T)
    from microwakeword_amd import native
    ex = outs[0]["exchanges"].tolist()
    segs = lay.segments()
    n_tail = sum(n for name, n in segs[[name for name, _ in segs].index("b2.dw.kernel"):])   # [blocks 3, 4 (0-based 2, 3) + dense]
    assert [f for _, f in ex] == [native.EXCHANGE_DEFERRED, native.EXCHANGE_IN_ORDER, native.EXCHANGE_FLUSH], ex
    assert ex[0][0] + ex[1][0] == lay.n_params and ex[2][0] == 0 and min(ex[0][0], ex[1][0]) > 0, ex
    assert ex[0][0] == n_tail, (ex, n_tail)
    # ... and the overlapped schedule is bit-identical to the single exchange after the backward pass
    mp.spawn(_sync_worker, args=(W, _free_port(), str(tmp_path), emu, "mixednet", False, 1, "one"), nprocs=W, join=True)
    ones = [np.load(tmp_path / ("one%d.npz" % r)) for r in range(W)]
    assert [f for _, f in ones[0]["exchanges"].tolist()] == [native.EXCHANGE_IN_ORDER]
    for k in ("grads", "params", "state"):
        np.testing.assert_array_equal(outs[0][k], ones[0][k])
    np.testing.assert_array_equal(outs[0]["grads"], outs[1]["grads"])      # the reduced gradient
    np.testing.assert_array_equal(outs[0]["params"], outs[1]["params"])    # hence identical weights
    assert np.abs(outs[0]["state"] - outs[1]["state"]).max() > 0           # but rank-local BN moving statistics
    gsum, states = 0.0, []
    for r in range(W):
        om = ec.perturbed_oracle(T)
        sl = slice(r * Bl, (r + 1) * Bl)
        _, _, grads, new_stats = om.loss_and_grads(x[sl], y[sl], w[sl])
        gsum = gsum + ec.oracle_grads_native_order(lay, om, grads).astype(np.float64)
        om.train_step(x[sl], y[sl], w[sl], 1e-3)
        states.append(lay.pack(om.get_weights())[1])
    g = outs[0]["grads"].astype(np.float64)
    assert np.linalg.norm(g - gsum) <= 2e-3 * np.linalg.norm(gsum)
    for r in range(W):
        assert np.abs(outs[r]["state"] - states[r]).max() <= 1e-5 * max(1.0, np.abs(states[r]).max())
    # Adam on the average: one Keras-Adam step from the common initial weights
    om = ec.perturbed_oracle(T)
    p0 = lay.pack(om.get_weights())[0].astype(np.float64)
    gavg = gsum / W
    m, v = 0.1 * gavg, 0.001 * gavg * gavg
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    p1 = p0 - alpha * m / (np.sqrt(v) + 1e-7)
    well = np.abs(gavg) > 1e-4 * np.abs(gavg).max()
    assert np.abs(outs[0]["params"] - p1)[well].max() <= 0.05 * 1e-3


def test_local_bn_two_ranks_inception_graph_kernels(tmp_path):
    """The conv/BN graph engine under the exchange hook with rank-local BatchNorm: the statistics hand-over (no finalize
    launches) and the gradient exchange in one step.  Two ranks x 3 windows of the default Inception graph (host-emulated
    product kernels) against two single-rank oracle passes summed by hand; identical reduced gradient and weights on both
    ranks, rank-local moving statistics."""
    import conftest
    import engine_checks as ec
    from microwakeword_amd.layout import InceptionLayout
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W, Bl = 2, 3
    rng = np.random.default_rng(4)
    x = ec.synth_x(rng, W * Bl, T)
    y = (rng.random(W * Bl) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=W * Bl).astype(np.float32)
    lay = InceptionLayout(ec.INC, T)
    last = lay.engine_args(1)["conv_ops"][-1]
    keep = (rng.random((W * Bl, last["tout"] * last["filters"])) > 0.2)
    np.savez(tmp_path / "inputs.npz", x=x, y=y, w=w, keep=keep)
    mp.spawn(_sync_worker, args=(W, _free_port(), str(tmp_path), emu, "inception", False), nprocs=W, join=True)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(W)]
    np.testing.assert_array_equal(outs[0]["grads"], outs[1]["grads"])
    np.testing.assert_array_equal(outs[0]["params"], outs[1]["params"])
    assert np.abs(outs[0]["state"] - outs[1]["state"]).max() > 0
    gsum = 0.0
    for r in range(W):
        om = ec.perturbed_inception_oracle(T, ec.INC)
        sl = slice(r * Bl, (r + 1) * Bl)
        _, _, grads, _ = om.loss_and_grads(x[sl], y[sl], w[sl], dropout_mask=keep[sl])
        gsum = gsum + lay.pack([grads[n].numpy().astype(np.float32) if k == "param" else np.zeros(sh, np.float32)
                                for n, sh, k in lay.keras_vars])[0].astype(np.float64)
        om.train_step(x[sl], y[sl], w[sl], 1e-3, dropout_mask=keep[sl])
        s_ref = lay.pack(om.get_weights())[1]
        assert np.abs(outs[r]["state"] - s_ref).max() <= 1e-5 * max(1.0, np.abs(s_ref).max())
    g = outs[0]["grads"].astype(np.float64)
    assert np.linalg.norm(g - gsum) <= 2e-3 * np.linalg.norm(gsum)


def test_local_bn_two_ranks_mixednet_graph_kernels(tmp_path):
    """As above for a MixedNet on the graph engine (convolution + BN and depthwise + bias ops): the hand-over through the
    depthwise kernels (forward fold as first consumer, backward sums added for the producing convolution, bias gradient
    folded by the weight-gradient launch) together with the gradient exchange."""
    import conftest
    import engine_checks as ec
    from microwakeword_amd.layout import GraphMixedNetLayout
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W, Bl = 2, 3
    rng = np.random.default_rng(5)
    x = ec.synth_x(rng, W * Bl, T)
    y = (rng.random(W * Bl) < 0.5).astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], size=W * Bl).astype(np.float32)
    lay = GraphMixedNetLayout(ec.GRAPH_MIXEDNET, T)
    np.savez(tmp_path / "inputs.npz", x=x, y=y, w=w, keep=np.zeros(1))
    mp.spawn(_sync_worker, args=(W, _free_port(), str(tmp_path), emu, "graph_mixednet", False), nprocs=W, join=True)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(W)]
    np.testing.assert_array_equal(outs[0]["grads"], outs[1]["grads"])
    np.testing.assert_array_equal(outs[0]["params"], outs[1]["params"])
    assert np.abs(outs[0]["state"] - outs[1]["state"]).max() > 0
    gsum = 0.0
    for r in range(W):
        om = ec.perturbed_oracle(T, flags=ec.GRAPH_MIXEDNET)
        sl = slice(r * Bl, (r + 1) * Bl)
        _, _, grads, _ = om.loss_and_grads(x[sl], y[sl], w[sl])
        gsum = gsum + lay.pack([grads[n].numpy().astype(np.float32) if k == "param" else np.zeros(sh, np.float32)
                                for n, sh, k in lay.keras_vars])[0].astype(np.float64)
        om.train_step(x[sl], y[sl], w[sl], 1e-3)
        s_ref = lay.pack(om.get_weights())[1]
        assert np.abs(outs[r]["state"] - s_ref).max() <= 1e-5 * max(1.0, np.abs(s_ref).max())
    g = outs[0]["grads"].astype(np.float64)
    assert np.linalg.norm(g - gsum) <= 2e-3 * np.linalg.norm(gsum)


# ------------------------------------------------------------------------------------------ the product train loop
# microwakeword_amd.train.train as one rank of a two-rank job: providers sharded, one gradient exchange per
# step inside the engine, validation sharded by window index with one all-reduce of the raw counters per result,
# rank 0 the only writer.  Product kernels (host-emulated) on both ranks.
_LOOP_T, _LOOP_B, _LOOP_STEPS = 60, 8, 4


def _loop_config(run_dir, n=24):
    import engine_checks as ec
    return dict(ec.learnable_config(n=n, T=_LOOP_T), train_dir=str(run_dir), summaries_dir=os.path.join(str(run_dir), "logs"),
                batch_size=_LOOP_B, spectrogram_length=_LOOP_T, training_steps=[2, 2], learning_rates=[0.01, 0.003],
                time_mask_max_size=[3], time_mask_count=[1], freq_mask_max_size=[3], freq_mask_count=[1],
                positive_class_weight=[1.0], negative_class_weight=[1.0], eval_step_interval=2, target_minimization=0.9,
                minimization_metric=None, maximization_metric="accuracy")


def _final_validation(tr, cfg, fh, model):
    nm = tr.validate_nonstreaming(cfg, fh, model, "validation")
    res = model.evaluation_results()   # validation + ambient windows accumulated (the reference's no-op reset swap)
    counts = {k: res[k].numpy().copy() for k in ("tp", "fp", "tn", "fn")}
    return nm, counts


def _train_loop_worker(rank, world, port, out_dir, emu_path, sync_bn, n=24):
    import engine_checks as ec
    from microwakeword_amd import mixednet, native
    from microwakeword_amd import train as tr
    from microwakeword_amd.data import FeatureHandler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    lib = native.NativeLib(emu_path)
    cfg = dict(_loop_config(os.path.join(out_dir, "run"), n=n), sync_bn=sync_bn)
    os.makedirs(cfg["train_dir"], exist_ok=True)
    # the ranks deliberately disagree on everything a single process would get from its seeds: initial weights and the
    # per-mode shuffles of the providers - the loop has to make them agree (broadcast, canonical shards)
    random.seed(100 + rank)
    np.random.seed(100 + rank)
    model = mixednet.model(ec.DEF, (_LOOP_T, 40), _LOOP_B // world, lib=lib, seed=7 + rank, max_batch=64)
    fh = FeatureHandler(cfg, engine=model.engine)
    n_train = [len(p.feature_sets["training"]) for p in fh.feature_providers]
    full_bytes = fh.resident_bytes
    train_bytes = sum(int(np.asarray(p.loaded_features[fi][sub]).nbytes) for p in fh.feature_providers for fi, sub in p.feature_sets["training"])
    out = tr.train(model, cfg, fh, verbose=False)
    shard = [sorted(p.feature_sets["training"]) for p in fh.feature_providers]
    assert all(n // world <= len(s) <= -(-n // world) for s, n in zip(shard, n_train))
    # SURVEY 8(e) "each rank uploads its shard to its own HBM": what stays resident is this rank's training samples (1 / W of
    # the training stores, to the raggedness of the sample lengths) plus the validation / ambient samples every rank scores
    mine = sum(int(np.asarray(p.loaded_features[fi][sub]).nbytes) for p in fh.feature_providers for fi, sub in p.feature_sets["training"])
    assert fh.resident_bytes == full_bytes - train_bytes + mine
    assert abs(mine - train_bytes / world) <= 0.25 * train_bytes / world, (mine, train_bytes, world)
    nm, counts = _final_validation(tr, cfg, fh, model)
    m, v, step = model.engine.get_opt_state()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), params=model.engine.get_params(), state=model.engine.get_bn_state(), m=m, v=v,
             step=step, best=np.array([out["best_minimization"], out["best_maximization"], out["best_no_faph_cutoff"]]),
             nm=np.array([nm[k] for k in sorted(nm)], np.float64), shard0=np.array(shard[0]), **counts)
    model.engine.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [False, True])
def test_train_loop_two_ranks_end_to_end(tmp_path, sync_bn):
    import conftest
    import engine_checks as ec
    from microwakeword_amd import mixednet, native
    from microwakeword_amd import train as tr
    from microwakeword_amd.data import FeatureHandler
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W = 2
    mp.spawn(_train_loop_worker, args=(W, _free_port(), str(tmp_path), emu, sync_bn), nprocs=W, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(W)]
    # identical model, optimizer state, decisions and validation metrics on both ranks
    for k in ("params", "state", "m", "v", "step", "best", "nm", "tp", "fp", "tn", "fn"):
        np.testing.assert_array_equal(r[0][k], r[1][k], err_msg=k)
    assert int(r[0]["step"]) == _LOOP_STEPS
    # the shards partition the provider
    assert not set(map(tuple, r[0]["shard0"])) & set(map(tuple, r[1]["shard0"]))
    assert len(r[0]["shard0"]) + len(r[1]["shard0"]) == 24
    # ONE set of files, written once: two validation passes -> two summary lines (two writers would leave four)
    run = tmp_path / "run"
    for f in ("best_weights.weights.h5.npz", "last_weights.weights.h5.npz", "restore/ckpt.weights.npz", "restore/ckpt.opt.npz"):
        assert (run / f).exists(), f
    assert len((run / "logs" / "validation" / "scalars.jsonl").read_text().splitlines()) == 2
    assert len((run / "logs" / "train" / "scalars.jsonl").read_text().splitlines()) == 2
    assert sorted(os.listdir(run / "train")) == sorted("%d_weights_%d.weights.h5.npz" % (b, s) for b, s in ((100000000, 2), (0, 4)))
    # the sharded validation's summed counters are the single-process counters of the same model
    lib = native.NativeLib(emu)
    cfg = _loop_config(tmp_path / "single")
    random.seed(5)
    np.random.seed(5)
    model = mixednet.model(ec.DEF, (_LOOP_T, 40), _LOOP_B, lib=lib, seed=1, max_batch=64)
    model.load_weights(str(run / "last_weights.weights.h5"))
    np.testing.assert_array_equal(model.engine.get_params(), r[0]["params"])
    np.testing.assert_array_equal(model.engine.get_bn_state(), r[0]["state"])
    fh = FeatureHandler(cfg, engine=model.engine)
    nm, counts = _final_validation(tr, cfg, fh, model)
    for k in ("tp", "fp", "tn", "fn"):
        np.testing.assert_array_equal(counts[k], r[0][k], err_msg=k)
    np.testing.assert_allclose(np.array([nm[k] for k in sorted(nm)], np.float64), r[0]["nm"], rtol=1e-6, atol=1e-9)
    model.engine.close()


def test_train_loop_four_ranks_uneven_shards(tmp_path):
    """The same loop on FOUR ranks with provider sizes that four does not divide (26 samples per provider: shards of 7, 7, 6, 6;
    two windows per rank and step): identical model / optimizer state / decisions on every rank, the shards a partition, one
    writer.  (Readiness for the first multi-GPU lease: nothing above two ranks had ever run.)"""
    import conftest
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    W, n = 4, 26
    mp.spawn(_train_loop_worker, args=(W, _free_port(), str(tmp_path), emu, False, n), nprocs=W, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(W)]
    for k in ("params", "state", "m", "v", "step", "best", "nm", "tp", "fp", "tn", "fn"):
        for j in range(1, W):
            np.testing.assert_array_equal(r[0][k], r[j][k], err_msg="%s rank %d" % (k, j))
    assert int(r[0]["step"]) == _LOOP_STEPS
    shards = [set(map(tuple, r[k]["shard0"])) for k in range(W)]
    assert sorted(len(s) for s in shards) == [6, 6, 7, 7]
    assert len(set().union(*shards)) == sum(len(s) for s in shards)   # pairwise disjoint
    run = tmp_path / "run"
    assert len((run / "logs" / "validation" / "scalars.jsonl").read_text().splitlines()) == 2


def _cli_worker(rank, world, port, emu_path, argv, expect_exists):
    from microwakeword_amd import model_train_eval
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),   # the emulator has one "device"
                      MWW_DIST_BACKEND="gloo", MWW_HIP_LIB=emu_path)
    torch.set_num_threads(1)
    if expect_exists:
        with pytest.raises(ValueError, match="already exists"):   # every rank raises, none hangs in a collective
            model_train_eval.main(argv)
    else:
        out = model_train_eval.main(argv)
        assert set(out) == {"best_minimization", "best_maximization", "best_no_faph_cutoff"}
    assert not dist.is_initialized()   # main() ends the process group it created, also when it raises


def test_cli_two_ranks_from_disk(tmp_path):
    """``python -m torch.distributed.run --nproc-per-node 2 -m microwakeword_amd.model_train_eval ...`` as the launcher
    sees it: two processes with RANK / LOCAL_RANK / WORLD_SIZE in the environment run ``main`` on stores read from disk;
    rank 0 owns train_dir; a second launch into the same folder raises on BOTH ranks; a third restores the checkpoint."""
    import yaml

    import conftest
    from microwakeword_amd import ragged
    emu = conftest.build_emulator_lib()
    if emu is None:
        pytest.skip("clang++ not available for the host-side emulator build")
    rng = np.random.default_rng(0)
    for prov, positive in (("wake", True), ("background", False)):
        for mode, n in (("training", 12), ("validation", 6), ("validation_ambient", 2)):
            if mode == "validation_ambient" and positive:
                continue
            lo, hi = (200, 260) if mode == "validation_ambient" else (62, 90)
            samples = [rng.integers(0, 200, size=(int(rng.integers(lo, hi)), 40)).astype(np.uint16) for _ in range(n)]
            ragged.write_ragged_store(str(tmp_path / prov / mode / ("%s_mmap" % mode)), samples)
    cfg = dict(window_step_ms=10, train_dir=str(tmp_path / "trained"), clip_duration_ms=160, batch_size=4, training_steps=[4],
               learning_rates=[0.001], eval_step_interval=2, target_minimization=0.9, minimization_metric=None,
               maximization_metric="average_viable_recall", time_mask_max_size=[3], time_mask_count=[1], freq_mask_max_size=[3],
               freq_mask_count=[1], positive_class_weight=[1], negative_class_weight=[1],
               features=[dict(features_dir=str(tmp_path / "wake"), sampling_weight=1.0, penalty_weight=1.0, truth=True,
                              truncation_strategy="truncate_start", type="mmap"),
                         dict(features_dir=str(tmp_path / "background"), sampling_weight=2.0, penalty_weight=1.0, truth=False,
                              truncation_strategy="random", type="mmap")])
    (tmp_path / "cfg.yaml").write_text(yaml.dump(cfg))
    argv = ["--training_config", str(tmp_path / "cfg.yaml"), "--verbosity", "ERROR", "mixednet", "--residual_connection", "0,0,0,0"]
    mp.spawn(_cli_worker, args=(2, _free_port(), emu, argv, False), nprocs=2, join=True)
    run = tmp_path / "trained"
    for f in ("training_config.yaml", "model_summary.txt", "best_weights.weights.h5.npz", "last_weights.weights.h5.npz",
              "restore/ckpt.weights.npz", "restore/ckpt.opt.npz", "logs/train/scalars.jsonl", "logs/validation/scalars.jsonl"):
        assert (run / f).exists(), f
    assert len((run / "logs" / "validation" / "scalars.jsonl").read_text().splitlines()) == 2
    assert int(np.load(run / "restore" / "ckpt.opt.npz")["step"]) == 4
    mp.spawn(_cli_worker, args=(2, _free_port(), emu, argv, True), nprocs=2, join=True)
    mp.spawn(_cli_worker, args=(2, _free_port(), emu, argv[:2] + ["--restore_checkpoint", "1"] + argv[2:], False), nprocs=2, join=True)
    assert int(np.load(run / "restore" / "ckpt.opt.npz")["step"]) == 8   # 4 restored (rank 0, broadcast) + 4 new steps
