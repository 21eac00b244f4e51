"""Data-parallel training over the GPUs of one node: one process per GPU, one exchange step per train step —
an all-reduce(sum) of the flat 22 177-float gradient (88.7 KB: latency-bound) over RCCL / xGMI.

The reference has no distributed code at all (SURVEY §2 "Parallelism strategies: none"); this is
new work defined by SURVEY §8(e):
  * each rank draws its own B/W windows per step from its shard of every provider
    (sample i -> rank i mod W) with a rank-distinct RNG stream (seed*W + rank);
  * local forward/backward -> flat gradient of the LOCAL mean loss;
  * all-reduce(sum), then Adam consumes grad/W (== gradient of the global-batch mean loss);
  * weights, Adam slots and the step counter stay bit-identical across ranks because every rank
    applies the same reduced gradient;
  * BatchNorm: ``sync_bn=False`` ("throughput mode", the default and what bench.py measures) normalises
    over the rank's own batch; ``sync_bn=True`` ("parity mode") exchanges the per-channel statistics
    sums of every BN layer in the forward and in the backward, so W ranks x B/W windows reproduce the
    single-device step on the global batch (no hipGraph replay).

Who issues the collective: on the GPU the LIBRARY does (``mww_allreduce_init``: ncclAllReduce from the launching
thread, on the engine's stream or on a library-owned side stream for a deferred bucket - no Python in the step);
``torch.distributed`` only carries the 128-byte communicator id to the ranks, the initial weight broadcast and the
barriers of the caller.  The callback form (``mww_set_allreduce_hook`` -> ``DataParallel._allreduce`` ->
``dist.all_reduce``) remains for process groups RCCL cannot serve (the two-rank gloo tests on the host-emulated kernels).
"""
from __future__ import annotations

import contextlib
import ctypes
import random
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import native


class _DeviceArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def wrap_device_floats(ptr: int, n: int, device) -> torch.Tensor:
    """A torch view of ``n`` floats of engine-owned HBM (no copy) so RCCL can reduce it in place."""
    return torch.as_tensor(_DeviceArray(ptr, n), device=device)


def host_floats(ptr: int, n: int) -> torch.Tensor:
    """A torch view of ``n`` floats of HOST memory (the "device" memory of a host-emulated build of the library, which a
    gloo group can reduce in place)."""
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr)))


def shard_feature_handler(handler, rank: int, world: int, seed: Optional[int] = None, prefetch: int = 4, epoch: int = 0):
    """Per provider keep training samples ``rank, rank+W, ...`` of the provider's list in CANONICAL (store, sample) order -
    a partition whatever per-rank shuffle produced the list (``MmapFeatureProvider`` shuffles with the global ``random``
    stream, which need not stand at the same point on every rank) - and give the rank its own RNG streams.  Validation
    windows are sharded by index in ``FeatureHandler.evaluate_on_device`` (``handler.eval_shard``).  SURVEY 8(e).

    The rank's streams are seeded from ``(base, rank, epoch)``: ``base`` is ``seed`` when the caller gives one, else 32 bits
    drawn from the global ``random`` stream as it stands (a run the user seeded stays a function of that seed, an unseeded
    run stays unseeded); ``epoch`` is the optimizer step the run resumes from, so a relaunch with ``--restore_checkpoint``
    continues with fresh draws instead of replaying the batches of step 1.  Idempotent: a handler already sharded for this
    (rank, world) keeps its lists (``train()`` called twice must not take a shard of a shard)."""
    if hasattr(handler, "shard_training_lists"):
        # FeatureHandler: the lists AND the HBM image (only this rank's training samples stay resident: 1 / W of the stores)
        handler.shard_training_lists(int(rank), int(world))
    elif getattr(handler, "_sharded_for", None) != (int(rank), int(world)):
        if getattr(handler, "_sharded_for", None) is not None:
            raise ValueError("feature handler already sharded for rank/world %r" % (handler._sharded_for,))
        for p in handler.feature_providers:
            p.feature_sets["training"] = sorted(p.feature_sets["training"])[rank::world]
            if p.stats["training"]["spectrogram_count"] and not p.feature_sets["training"]:
                raise ValueError("provider has fewer training samples than ranks")
        handler._sharded_for = (int(rank), int(world))
    handler._sampler = None
    handler.eval_shard = (int(rank), int(world))
    base = int(seed) if seed is not None else random.getrandbits(32)
    mixed = (base * world + rank) * 1000003 + int(epoch)
    random.seed(mixed)
    np.random.seed(mixed % (2 ** 32))
    try:
        handler.use_private_rng(prefetch=prefetch)
    except TypeError:   # duck-typed handlers without the prefetch argument
        handler.use_private_rng()


class DataParallel:
    """Wraps one engine per rank.  ``grad_view`` / ``param_view`` are torch tensors aliasing the
    engine's flat gradient / parameter vectors (device memory on the GPU path).

    ``for_engine`` (the GPU path) joins the ranks' engines into an RCCL communicator owned by the library:
    ``engine.train_step`` is then the complete data-parallel step and nothing of it runs in Python.
    With ``wrap`` and no library communicator the engine drives the exchange through the ``mww_set_allreduce_hook``
    callback into ``_allreduce`` (gloo groups, host-emulated kernels).  ``grad_buckets`` = 2 hands the gradient over in
    two buckets - [dense + the last two blocks] right after those blocks' backward kernels are enqueued, as a *deferred*
    exchange that runs on a side stream next to the remaining backward kernels, the rest after the first block's
    backward; it is bit-identical to the single exchange and 8 % slower at W = 1, unmeasured at W > 1, so the
    default is 1.  Without ``wrap`` (engines that only expose NO_APPLY + apply, e.g. the test stub) the single
    all-reduce is issued from here."""

    def __init__(self, engine, grad_view: torch.Tensor, param_view: torch.Tensor, state_view: Optional[torch.Tensor] = None,
                 group=None, sync_bn: bool = False, wrap=None, grad_buckets: int = 1, library_comm: bool = False):
        """``wrap(ptr, n) -> tensor`` turns a device address handed out by the engine into a tensor the
        process group can reduce in place (defaults to a zero-copy view of HBM on the engine's device)."""
        self.engine = engine
        self.grad_view, self.param_view, self.state_view = grad_view, param_view, state_view
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.sync_bn = bool(sync_bn)
        self._wrap = wrap
        self._views = {}
        self._pending = []
        self.exchanges = []   # (n, flags) of the hook calls of the last step (tests / diagnostics)
        self._step_open = False
        self._managed = False   # inside DataParallel.train_step (which opens and closes the record itself)
        self._grads_base = int(engine.device_ptr(native.BUF_GRADS)) if hasattr(engine, "device_ptr") else 0
        self.library_comm = bool(library_comm)
        if self.sync_bn and wrap is None and not library_comm:
            raise ValueError("sync_bn needs a wrap(ptr, n) function")
        self.engine_driven = wrap is not None or library_comm
        if library_comm:
            # the ranks decide TOGETHER whether the library-owned communicator is used (every step of the join is followed by
            # an all_reduce(MIN) of its success flag): a rank that cannot load librccl, or whose ncclCommInitRank fails, takes
            # every rank to the callback form over the caller's process group - or every rank raises
            err = self._join_library_communicator()
            if err is not None:
                if wrap is None:
                    raise err
                import warnings
                warnings.warn("library-owned RCCL communicator unavailable (%s): using torch.distributed through the callback" % err)
                self.library_comm = library_comm = False
        if library_comm:
            engine.set_option("grad_buckets", int(grad_buckets))
        elif self.engine_driven:
            # the engine calls back for every BN layer (sync-BN) and for the gradient buckets: the complete DP step
            engine.set_allreduce_hook(self._allreduce, self.world, sync_bn=self.sync_bn, reduce_grads=True)
            engine.set_option("grad_buckets", int(grad_buckets))

    @classmethod
    def for_engine(cls, engine: native.Engine, device=None, group=None, sync_bn: bool = False, grad_buckets: int = 1,
                   library_comm: bool = True):
        """The engine's own flat vectors as the views.  On the GPU (NCCL = RCCL process group) they are zero-copy views of HBM
        and the library joins its own communicator; a host-emulated build of the library (its "device" memory is host
        memory) is served by the callback over whatever group the caller initialised (gloo)."""
        if getattr(engine.nl, "host_emulated", False):
            wrap, library_comm = host_floats, False
        else:
            if dist.is_initialized() and dist.get_backend(group) != "nccl":
                raise ValueError("data-parallel training on the GPU needs an NCCL (RCCL) process group, not %r" % dist.get_backend(group))
            if device is None:
                device = torch.device("cuda", engine.device)

            def wrap(ptr, n):
                return wrap_device_floats(ptr, n, device)
        g = wrap(engine.device_ptr(native.BUF_GRADS), engine.n_params)
        p = wrap(engine.device_ptr(native.BUF_PARAMS), engine.n_params)
        s = wrap(engine.device_ptr(native.BUF_BN_STATE), engine.n_state)
        return cls(engine, g, p, s, group, sync_bn=sync_bn, wrap=wrap, grad_buckets=grad_buckets, library_comm=library_comm)

    # ---- small host-side collectives of the train loop (start-up, validation counters): never inside a step
    def agree(self, ok: bool) -> bool:
        """True iff ``ok`` on every rank (all_reduce(MIN) of the flag)."""
        if not dist.is_initialized() or self.world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.grad_view.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def allreduce_host(self, a: np.ndarray) -> np.ndarray:
        """Sum of a small float64 host vector over the ranks (every rank receives the same bits)."""
        a = np.ascontiguousarray(a, np.float64)
        if not dist.is_initialized() or self.world == 1:
            return a
        t = torch.from_numpy(a.copy()).to(self.grad_view.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def broadcast_host(self, a: np.ndarray, src: int = 0) -> np.ndarray:
        a = np.ascontiguousarray(a)
        if not dist.is_initialized() or self.world == 1:
            return a
        t = torch.from_numpy(a.copy()).to(self.grad_view.device)
        dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def barrier(self):
        if dist.is_initialized() and self.world > 1:
            self.agree(True)

    def _join_library_communicator(self):
        """Every rank probes the library's RCCL binding (``mww_allreduce_unique_id`` loads librccl; only rank 0's id is
        used), the process group carries rank 0's 128 bytes to the other ranks, every rank joins (``mww_allreduce_init`` is
        collective).  Returns None, or the error every rank reports after the ranks agreed that one of them failed."""
        dev = self.grad_view.device
        err, mine = None, None
        try:
            mine = self.engine.nl.allreduce_unique_id()
        except native.NativeError as e:
            err = e
        if not self.agree(err is None):
            return err or native.NativeError("another rank could not load librccl")
        uid = torch.zeros(native.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            uid.copy_(torch.from_numpy(mine))
        if dist.is_initialized() and self.world > 1:
            dist.broadcast(uid, src=0, group=self.group)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
        try:
            self.engine.allreduce_init(self.rank, self.world, uid.cpu().numpy(), sync_bn=self.sync_bn)
        except native.NativeError as e:
            err = e
        if not self.agree(err is None):
            if err is None:
                self.engine.allreduce_destroy()
            return err or native.NativeError("another rank could not join the RCCL communicator")
        return None

    def _engine_stream(self):
        """The callback's collectives are ordered against torch's CURRENT stream; the engine's kernels run on the engine's
        stream.  Entering that stream here makes the two the same whatever the caller's stream context is (an engine on a
        private stream, or a train_step called outside ``torch.cuda.stream(...)``, would otherwise race the all-reduce
        against the backward kernels and Adam)."""
        if self.grad_view.is_cuda:
            h = self.engine.device_ptr(native.BUF_STREAM)
            if h:
                return torch.cuda.stream(torch.cuda.ExternalStream(h, device=self.grad_view.device))
        return contextlib.nullcontext()

    def _allreduce(self, ptr: int, n: int, flags: int = native.EXCHANGE_IN_ORDER):
        """Hook target (include/mww.h mww_allreduce_fn).  With the engine's stream current, an in-order exchange is a
        plain ``dist.all_reduce`` (the NCCL backend orders it after, and the current stream behind, the collective); a
        deferred bucket is issued ``async_op`` - it starts once the kernels enqueued so far are done and runs on the
        communicator's stream - and its ``wait()`` is what the flush call enqueues."""
        if not self._step_open:      # a caller driving engine.train_step directly: keep one step's worth of records
            self.exchanges = []
            self._step_open = True
        self.exchanges.append((int(n), int(flags)))
        if not self._managed and flags != native.EXCHANGE_DEFERRED and (flags == native.EXCHANGE_FLUSH or ptr == self._grads_base):
            # the exchange that ends a step (the bucket that starts at the head of the flat gradient, or the flush behind the
            # deferred one): the next hook call belongs to the next step and starts a new record
            self._step_open = False
        with self._engine_stream():
            if flags == native.EXCHANGE_FLUSH:
                for w in self._pending:
                    w.wait()
                self._pending = []
                return
            t = self._views.get((ptr, n))
            if t is None:
                t = self._views[(ptr, n)] = self._wrap(ptr, n)
            if dist.is_initialized():
                if flags == native.EXCHANGE_DEFERRED:
                    self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_parameters(self, src=0, optimizer_state=False):
        """Weights and BN moving statistics of rank ``src`` to every rank; with ``optimizer_state`` also the Adam slots and
        the step counter (a resumed run: every rank continues from the checkpoint rank ``src`` restored)."""
        if dist.is_initialized():
            self.engine.synchronize()
            dist.broadcast(self.param_view, src=src, group=self.group)
            if self.state_view is not None:
                dist.broadcast(self.state_view, src=src, group=self.group)
            if self.grad_view.is_cuda:
                torch.cuda.synchronize(self.grad_view.device)   # the engine's stream is not torch's: order by completion
            if optimizer_state and hasattr(self.engine, "get_opt_state"):
                m, v, step = self.engine.get_opt_state()
                packed = np.concatenate([np.asarray(m, np.float64).reshape(-1), np.asarray(v, np.float64).reshape(-1), [float(step)]])
                packed = self.broadcast_host(packed, src=src)
                n = (packed.size - 1) // 2
                self.engine.set_opt_state(packed[:n].astype(np.float32), packed[n:2 * n].astype(np.float32), int(packed[-1]))

    def average_bn_state(self):
        """Rank-local BatchNorm (throughput mode) leaves every rank with its own moving statistics.  Before the model is
        validated or saved the ranks take their mean (one all-reduce of the small state vector), so that every rank scores -
        and rank 0 writes - the same model.  A no-op with sync-BN (identical already) and on one rank."""
        if self.state_view is None or self.sync_bn or not dist.is_initialized() or self.world == 1:
            return
        self.engine.synchronize()
        dist.all_reduce(self.state_view, op=dist.ReduceOp.SUM, group=self.group)
        self.state_view.mul_(1.0 / self.world)
        if self.grad_view.is_cuda:
            torch.cuda.synchronize(self.grad_view.device)

    def train_step(self, B, lr, flags=0, prefetch=None):
        """Local forward/backward, gradient all-reduce, Adam on the averaged gradient.  ``prefetch`` (optional
        callable, e.g. the draw of the next batch) runs after the step has been enqueued."""
        self.exchanges = []
        self._step_open = True
        self._managed = True
        try:
            if self.engine_driven:
                self.engine.train_step(B, lr, flags)   # statistics / gradient exchanges happen inside
                if prefetch is not None:
                    prefetch()
                return
            self.engine.train_step(B, lr, flags | native.STEP_NO_APPLY)
            work = None
            if dist.is_initialized():
                work = dist.all_reduce(self.grad_view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if prefetch is not None:
                prefetch()
            if work is not None:
                work.wait()   # orders the compute stream after the collective; no host block on RCCL
            self.engine.apply_gradients(lr, 1.0 / self.world)
        finally:
            self._step_open = False
            self._managed = False
