"""Data-parallel training over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" == RCCL over xGMI) for the single exchange step of the path — an all-reduce(sum)
of the flat 22 177-float gradient (88.7 KB: latency-bound), issued in two buckets so that the first one
([dense + blocks 4, 3], 54.5 KB) overlaps the backward kernels of blocks 2 and 1.

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
    sums of every BN layer in the forward and in the backward through ``mww_set_allreduce_hook``, so
    W ranks x B/W windows reproduce the single-device step on the global batch (no hipGraph replay).
"""
from __future__ import annotations

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


def shard_feature_handler(handler, rank: int, world: int, seed: int):
    """Per provider keep training samples ``rank, rank+W, ...`` of the (identically shuffled) list and
    give the rank its own RNG streams."""
    for p in handler.feature_providers:
        p.feature_sets["training"] = p.feature_sets["training"][rank::world]
        if p.stats["training"]["spectrogram_count"] and not p.feature_sets["training"]:
            raise ValueError("provider has fewer training samples than ranks")
    handler._sampler = None
    random.seed(seed * world + rank)
    np.random.seed(seed * world + rank)
    handler.use_private_rng()


class DataParallel:
    """Wraps one engine per rank.  ``grad_view`` / ``param_view`` are torch tensors aliasing the
    engine's flat gradient / parameter vectors (device memory on the GPU path).

    With ``wrap`` (the normal case: ``for_engine``) the engine drives the exchange through
    ``mww_set_allreduce_hook``: ``engine.train_step`` is the complete data-parallel step.  In throughput
    mode (local BatchNorm) the specialised MixedNet kernels hand the gradient over in TWO buckets - [dense +
    the last two blocks] right after those blocks' backward kernels are enqueued, as a *deferred* exchange
    that RCCL runs on its own stream next to the remaining backward kernels, and the rest after the first
    block's backward - then Adam consumes grad/W.  Without ``wrap`` (engines that only expose
    NO_APPLY + apply, e.g. the test stub) the single all-reduce is issued from here."""

    def __init__(self, engine, grad_view: torch.Tensor, param_view: torch.Tensor, state_view: Optional[torch.Tensor] = None,
                 group=None, sync_bn: bool = False, wrap=None, grad_buckets: int = 2):
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
        if self.sync_bn and wrap is None:
            raise ValueError("sync_bn needs a wrap(ptr, n) function")
        self.engine_driven = wrap is not None
        if self.engine_driven:
            # the engine calls back for every BN layer (sync-BN) and for the gradient buckets: the complete DP step
            engine.set_allreduce_hook(self._allreduce, self.world, sync_bn=self.sync_bn, reduce_grads=True)
            engine.set_option("grad_buckets", int(grad_buckets))

    @classmethod
    def for_engine(cls, engine: native.Engine, device, group=None, sync_bn: bool = False, grad_buckets: int = 2):
        g = wrap_device_floats(engine.device_ptr(native.BUF_GRADS), engine.n_params, device)
        p = wrap_device_floats(engine.device_ptr(native.BUF_PARAMS), engine.n_params, device)
        s = wrap_device_floats(engine.device_ptr(native.BUF_BN_STATE), engine.n_state, device)
        return cls(engine, g, p, s, group, sync_bn=sync_bn, wrap=lambda ptr, n: wrap_device_floats(ptr, n, device),
                   grad_buckets=grad_buckets)

    def _allreduce(self, ptr: int, n: int, flags: int = native.EXCHANGE_IN_ORDER):
        """Hook target (include/mww.h mww_allreduce_fn).  The engine lives on torch's current stream, so an in-order
        exchange is a plain ``dist.all_reduce`` (the NCCL backend orders it after, and the current stream behind, the
        collective); a deferred bucket is issued ``async_op`` - it starts once the kernels enqueued so far are done and
        runs on the communicator's stream - and its ``wait()`` is what the flush call enqueues."""
        self.exchanges.append((int(n), int(flags)))
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

    def broadcast_parameters(self, src=0):
        if dist.is_initialized():
            self.engine.synchronize()
            dist.broadcast(self.param_view, src=src, group=self.group)
            if self.state_view is not None:
                dist.broadcast(self.state_view, src=src, group=self.group)

    def train_step(self, B, lr, flags=0, prefetch=None):
        """Local forward/backward, gradient all-reduce (overlapped with the backward tail when the engine drives
        it), Adam on the averaged gradient.  ``prefetch`` (optional callable, e.g. the draw of the next batch) runs
        after the step has been enqueued."""
        self.exchanges = []
        if self.engine_driven:
            self.engine.train_step(B, lr, flags)   # statistics / gradient exchanges happen inside, via the hook
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
