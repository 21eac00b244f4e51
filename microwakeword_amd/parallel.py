"""Data-parallel training over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" == RCCL over xGMI) for the single exchange step of the path — an all-reduce(sum)
of the flat 22 177-float gradient (88.7 KB: latency-bound, one collective, no bucketing).

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
    engine's flat gradient / parameter vectors (device memory on the GPU path)."""

    def __init__(self, engine, grad_view: torch.Tensor, param_view: torch.Tensor, state_view: Optional[torch.Tensor] = None,
                 group=None, sync_bn: bool = False, wrap=None):
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
        if self.sync_bn:
            if wrap is None:
                raise ValueError("sync_bn needs a wrap(ptr, n) function")
            # the engine calls back for every BN layer (and for the gradient): the complete DP step
            engine.set_allreduce_hook(self._allreduce, self.world, sync_bn=True, reduce_grads=True)

    @classmethod
    def for_engine(cls, engine: native.Engine, device, group=None, sync_bn: bool = False):
        g = wrap_device_floats(engine.device_ptr(native.BUF_GRADS), engine.n_params, device)
        p = wrap_device_floats(engine.device_ptr(native.BUF_PARAMS), engine.n_params, device)
        s = wrap_device_floats(engine.device_ptr(native.BUF_BN_STATE), engine.n_state, device)
        return cls(engine, g, p, s, group, sync_bn=sync_bn, wrap=lambda ptr, n: wrap_device_floats(ptr, n, device))

    def _allreduce(self, ptr: int, n: int):
        """Hook target: enqueue sum-all-reduce of n floats at ptr (views are cached per buffer).  With
        the engine created on torch's current stream the collective is ordered with its kernels."""
        t = self._views.get((ptr, n))
        if t is None:
            t = self._views[(ptr, n)] = self._wrap(ptr, n)
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_parameters(self, src=0):
        if dist.is_initialized():
            self.engine.synchronize()
            dist.broadcast(self.param_view, src=src, group=self.group)
            if self.state_view is not None:
                dist.broadcast(self.state_view, src=src, group=self.group)

    def train_step(self, B, lr, flags=0, prefetch=None):
        """Local forward/backward, gradient all-reduce, Adam on the averaged gradient.

        ``prefetch`` (optional callable) is run between launching the all-reduce and waiting for it: the
        train loop passes the assembly of the NEXT batch, whose gather kernel then runs on the compute
        stream while RCCL moves the 88 KB gradient on its own stream (the step's backward has finished
        with the batch buffers by then, so the single set of buffers suffices)."""
        if self.sync_bn:
            self.engine.train_step(B, lr, flags)   # statistics and gradient exchanges happen inside, via the hook
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
