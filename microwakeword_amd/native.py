"""ctypes binding of ``libmww_hip.so`` (C ABI: ``include/mww.h``).

There is deliberately **no CPU fallback**: if the HIP library is missing, fails to load, or no
MI355X is visible, every entry point raises.  (The library path can be overridden — explicit ``path``
argument or the ``MWW_HIP_LIB`` environment variable, e.g. for a build in another directory; the
test-suite uses that to load the host-side kernel emulator it builds under ``tests/hipemu`` from the same
sources.  Nothing in this package contains or selects a CPU implementation.)
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libmww_hip.so")

MWW_MAX_BLOCKS = 8
MWW_MAX_STORES = 64
MAX_MASKS = 16
STEP_NO_APPLY = 1
STEP_NO_METRICS = 2
BUF_PARAMS, BUF_GRADS, BUF_BN_STATE, BUF_X, BUF_STREAM = 0, 1, 2, 3, 4
UNIQUE_ID_BYTES = 128
DTYPE_U16, DTYPE_F32 = 0, 1
STRATEGIES = {"random": 0, "truncate_start": 1, "truncate_end": 2, "fixed_right_cutoff": 3, "none": 4}


class MixedNetDesc(C.Structure):
    _fields_ = [("frames", C.c_int32), ("conv1_filters", C.c_int32), ("conv1_kernel", C.c_int32),
                ("conv1_stride", C.c_int32), ("n_blocks", C.c_int32), ("block_filters", C.c_int32 * MWW_MAX_BLOCKS),
                ("block_kernel", C.c_int32 * MWW_MAX_BLOCKS), ("max_batch", C.c_int32)]


OP_KINDS = {"conv": 0, "depthwise": 1}
NORMS = {"bn": 0, "bias": 1, "none": 2}
ACTS = {"relu": 0, "linear": 1}
MWW_MAX_GRAPH_OPS = 48
MWW_MAX_OP_SOURCES = 3


class ConvBnOp(C.Structure):
    _fields_ = [("n_src", C.c_int32), ("src", C.c_int32 * MWW_MAX_OP_SOURCES), ("src_drop", C.c_int32 * MWW_MAX_OP_SOURCES),
                ("src_c0", C.c_int32 * MWW_MAX_OP_SOURCES), ("src_cn", C.c_int32 * MWW_MAX_OP_SOURCES),
                ("kernel", C.c_int32), ("dilation", C.c_int32), ("filters", C.c_int32), ("bn_groups", C.c_int32),
                ("kind", C.c_int32), ("stride", C.c_int32), ("norm", C.c_int32), ("act", C.c_int32),
                ("residual", C.c_int32), ("residual_drop", C.c_int32)]


class ConvNetDesc(C.Structure):
    _fields_ = [("frames", C.c_int32), ("n_ops", C.c_int32), ("ops", ConvBnOp * MWW_MAX_GRAPH_OPS),
                ("dropout", C.c_float), ("max_batch", C.c_int32), ("head_attention", C.c_int32), ("head_pool", C.c_int32)]


class Window(C.Structure):
    _fields_ = [("store", C.c_int32), ("pad_rows", C.c_int32), ("copy_rows", C.c_int32), ("reserved", C.c_int32),
                ("src_elem", C.c_int64)]


WINDOW_DTYPE = np.dtype([("store", np.int32), ("pad_rows", np.int32), ("copy_rows", np.int32), ("reserved", np.int32),
                         ("src_elem", np.int64)])
assert WINDOW_DTYPE.itemsize == C.sizeof(Window)


class MetricsRaw(C.Structure):
    _fields_ = [("hist101", C.c_uint64 * 101 * 2), ("hist200", C.c_uint64 * 200 * 2), ("n", C.c_uint64),
                ("correct", C.c_uint64), ("tp5", C.c_uint64), ("fp5", C.c_uint64), ("fn5", C.c_uint64),
                ("pos", C.c_uint64), ("neg", C.c_uint64), ("bce_sum", C.c_double)]


class SamplerDesc(C.Structure):
    _fields_ = [("n_providers", C.c_int32), ("sampling_weight", C.POINTER(C.c_double)), ("strategy", C.POINTER(C.c_int32)),
                ("set_offsets", C.POINTER(C.c_int64)), ("set_store", C.POINTER(C.c_int32)),
                ("set_src_elem", C.POINTER(C.c_int64)), ("set_len", C.POINTER(C.c_int32)),
                ("cutoff_offsets", C.POINTER(C.c_int32)), ("cutoffs", C.POINTER(C.c_int32))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)   # mww_allreduce_fn
EXCHANGE_IN_ORDER, EXCHANGE_DEFERRED, EXCHANGE_FLUSH = 0, 1, 2


class NativeError(RuntimeError):
    pass


EXPORTS = [
    "mww_version", "mww_last_error", "mww_device_count", "mww_block_kernels_cover", "mww_allreduce_world", "mww_create", "mww_create_convnet", "mww_set_dropout_mask",
    "mww_set_allreduce_hook",
    "mww_destroy", "mww_synchronize",
    "mww_num_params", "mww_num_bn_state", "mww_set_params", "mww_get_params", "mww_set_bn_state", "mww_get_bn_state",
    "mww_set_grad_mask", "mww_set_opt_state", "mww_get_opt_state", "mww_get_grads", "mww_upload_store",
    "mww_assemble_batch", "mww_set_batch", "mww_get_batch", "mww_set_targets", "mww_train_step", "mww_apply_gradients",
    "mww_forward", "mww_read_outputs", "mww_metrics_read", "mww_metrics_reset", "mww_device_ptr", "mww_debug_read",
    "mww_set_option", "mww_profile_read", "mww_sample_training_batch", "mww_rng_selftest",
    "mww_prefetch_create", "mww_prefetch_create_weighted", "mww_prefetch_acquire", "mww_prefetch_release", "mww_prefetch_rng_state", "mww_prefetch_shape",
    "mww_prefetch_destroy", "mww_assemble_prefetched",
    "mww_allreduce_unique_id", "mww_allreduce_init", "mww_allreduce_destroy", "mww_evaluate_windows",
]


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class NativeLib:
    """Loads the shared library and declares the prototypes."""

    _instances = {}

    def __init__(self, path: Optional[str] = None):
        self.path = path or os.environ.get("MWW_HIP_LIB") or DEFAULT_LIB
        if not os.path.isfile(self.path):
            raise NativeError(
                "HIP library %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % self.path)
        try:
            self.lib = C.CDLL(self.path)
        except OSError as e:
            raise NativeError("cannot load %s: %s" % (self.path, e)) from None
        L = self.lib
        # a host-emulated build of the product sources (tests/hipemu): its "device" memory is host memory, so process groups
        # without device support (gloo) can reduce it in place - parallel.DataParallel.for_engine asks
        self.host_emulated = hasattr(L, "hipemu_host_memory")
        for name in EXPORTS:
            if not hasattr(L, name):
                raise NativeError("%s does not export %s (stale build?)" % (self.path, name))
        L.mww_version.restype = C.c_char_p
        L.mww_last_error.restype = C.c_char_p
        L.mww_create.argtypes = [C.POINTER(MixedNetDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mww_block_kernels_cover.argtypes = [C.POINTER(MixedNetDesc), C.c_int]
        L.mww_create_convnet.argtypes = [C.POINTER(ConvNetDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mww_set_dropout_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mww_set_allreduce_hook.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mww_destroy.argtypes = [C.c_void_p]
        L.mww_destroy.restype = None
        L.mww_synchronize.argtypes = [C.c_void_p]
        L.mww_num_params.argtypes = [C.c_void_p]
        L.mww_num_params.restype = C.c_int64
        L.mww_num_bn_state.argtypes = [C.c_void_p]
        L.mww_num_bn_state.restype = C.c_int64
        for f in (L.mww_set_params, L.mww_get_params, L.mww_set_bn_state, L.mww_get_bn_state, L.mww_set_grad_mask,
                  L.mww_get_grads):
            f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
        L.mww_set_opt_state.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int64, C.c_int64]
        L.mww_get_opt_state.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int64,
                                        C.POINTER(C.c_int64)]
        L.mww_upload_store.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int]
        L.mww_assemble_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mww_set_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.mww_get_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.mww_set_targets.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.mww_train_step.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.mww_apply_gradients.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.mww_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mww_read_outputs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]
        L.mww_metrics_read.argtypes = [C.c_void_p, C.POINTER(MetricsRaw)]
        L.mww_metrics_reset.argtypes = [C.c_void_p]
        L.mww_device_ptr.argtypes = [C.c_void_p, C.c_int]
        L.mww_device_ptr.restype = C.c_void_p
        L.mww_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int64]
        L.mww_debug_read.restype = C.c_int64
        L.mww_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.mww_profile_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int]
        L.mww_sample_training_batch.argtypes = [C.POINTER(SamplerDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.mww_rng_selftest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.mww_prefetch_create.argtypes = [C.POINTER(SamplerDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int, C.POINTER(C.c_void_p)]
        L.mww_prefetch_create_weighted.argtypes = [C.POINTER(SamplerDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int,
                                                   C.POINTER(C.c_void_p)]
        L.mww_prefetch_acquire.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 6
        L.mww_prefetch_release.argtypes = [C.c_void_p]
        L.mww_prefetch_rng_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mww_prefetch_rng_state.restype = C.c_int64
        L.mww_prefetch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mww_prefetch_destroy.argtypes = [C.c_void_p]
        L.mww_prefetch_destroy.restype = None
        L.mww_assemble_prefetched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mww_allreduce_unique_id.argtypes = [C.c_void_p, C.c_int]
        L.mww_allreduce_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.mww_allreduce_destroy.argtypes = [C.c_void_p]
        L.mww_evaluate_windows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int]

    @classmethod
    def get(cls, path: Optional[str] = None) -> "NativeLib":
        key = path or os.environ.get("MWW_HIP_LIB") or DEFAULT_LIB
        if key not in cls._instances:
            cls._instances[key] = NativeLib(key)
        return cls._instances[key]

    def check(self, rc: int):
        if rc < 0:
            raise NativeError("libmww error %d: %s" % (rc, self.lib.mww_last_error().decode()))
        return rc

    def version(self) -> str:
        return self.lib.mww_version().decode()

    def device_count(self) -> int:
        return int(self.lib.mww_device_count())

    def block_kernels_cover(self, frames, conv1_filters, conv1_kernel, conv1_stride, block_filters, block_kernel, max_batch=1,
                            bf16=False):
        """``mww_block_kernels_cover``: (True, "") if every block of this MixedNet has a specialised MFMA block kernel, else
        (False, reason) - the model then runs on the conv / depthwise graph kernels.  Needs no GPU (build-time shape table)."""
        if len(block_filters) != len(block_kernel) or len(block_filters) > MWW_MAX_BLOCKS:
            return False, "bad block lists"
        d = MixedNetDesc()
        d.frames, d.conv1_filters, d.conv1_kernel, d.conv1_stride = int(frames), int(conv1_filters), int(conv1_kernel), int(conv1_stride)
        d.n_blocks = len(block_filters)
        for i, (f, k) in enumerate(zip(block_filters, block_kernel)):
            d.block_filters[i], d.block_kernel[i] = int(f), int(k)
        d.max_batch = int(max_batch)
        rc = int(self.lib.mww_block_kernels_cover(C.byref(d), 1 if bf16 else 0))
        if rc < 0:
            self.check(rc)
        return (True, "") if rc == 1 else (False, self.lib.mww_last_error().decode())

    def allreduce_unique_id(self) -> np.ndarray:
        """ncclGetUniqueId through the library: 128 bytes rank 0 hands to every rank before ``Engine.allreduce_init``."""
        uid = np.zeros(UNIQUE_ID_BYTES, np.uint8)
        self.check(self.lib.mww_allreduce_unique_id(uid.ctypes.data_as(C.c_void_p), uid.size))
        return uid


class Prefetcher:
    """``mww_prefetcher``: a worker thread that draws the training batches of the next steps (csrc/sampler.cpp) from
    private copies of the two MT19937 streams.  ``py_state`` / ``np_state``: uint32[625] as exported by
    ``FeatureHandler._export_global_rng``.  Host-only (usable without a GPU)."""

    BROADCAST = {"per_sample": 0, "keras_last_axis": 1, "keras_first_axis": 2}

    def __init__(self, nl: NativeLib, desc: SamplerDesc, labels, weights, py_state, np_state, B, T, tmax, tcount, fmax, fcount,
                 default_strategy=-1, depth=2, class_weights=None, broadcast="per_sample"):
        """``weights``: per-provider sample weight (penalty x class weight), or - with ``class_weights`` (per provider) - the
        penalty weights alone, combined per batch as ``broadcast`` says (model.combine_weights, train.py:288-293)."""
        self.nl = nl
        self.B, self.nm = int(B), int(tcount) + int(fcount)
        lab = np.ascontiguousarray(labels, np.float32)
        wts = np.ascontiguousarray(weights, np.float32)
        py = np.ascontiguousarray(py_state, np.uint32)
        npst = np.ascontiguousarray(np_state, np.uint32)
        if py.size != 625 or npst.size != 625 or lab.size != desc.n_providers or wts.size != desc.n_providers:
            raise ValueError("bad prefetcher arguments")
        h = C.c_void_p()
        if class_weights is None:
            nl.check(nl.lib.mww_prefetch_create(C.byref(desc), lab.ctypes.data_as(C.c_void_p), wts.ctypes.data_as(C.c_void_p),
                                                py.ctypes.data_as(C.c_void_p), npst.ctypes.data_as(C.c_void_p), int(B), int(T), int(tmax),
                                                int(tcount), int(fmax), int(fcount), int(default_strategy), int(depth), C.byref(h)))
        else:
            cws = np.ascontiguousarray(class_weights, np.float32)
            if cws.size != desc.n_providers or broadcast not in self.BROADCAST:
                raise ValueError("bad prefetcher arguments")
            nl.check(nl.lib.mww_prefetch_create_weighted(C.byref(desc), lab.ctypes.data_as(C.c_void_p), wts.ctypes.data_as(C.c_void_p),
                                                         cws.ctypes.data_as(C.c_void_p), self.BROADCAST[broadcast],
                                                         py.ctypes.data_as(C.c_void_p), npst.ctypes.data_as(C.c_void_p), int(B), int(T),
                                                         int(tmax), int(tcount), int(fmax), int(fcount), int(default_strategy), int(depth),
                                                         C.byref(h)))
        self.h = h

    def acquire(self):
        """Next batch as numpy copies: dict(windows, masks, labels, weights, provider, sample).  (Tests; the train loop
        uses ``Engine.assemble_prefetched``, which never copies to Python.)"""
        ptrs = [C.c_void_p() for _ in range(6)]
        self.nl.check(self.nl.lib.mww_prefetch_acquire(self.h, *[C.byref(p) for p in ptrs]))
        B, nm = self.B, max(self.nm, 1)

        def arr(p, ctype, n, dtype):
            return np.frombuffer((ctype * n).from_address(p.value), dtype=dtype).copy()
        out = dict(windows=np.frombuffer((C.c_char * (B * WINDOW_DTYPE.itemsize)).from_address(ptrs[0].value), dtype=WINDOW_DTYPE).copy(),
                   masks=arr(ptrs[1], C.c_int32, B * nm * 2, np.int32).reshape(B, nm, 2)[:, :self.nm],
                   labels=arr(ptrs[2], C.c_float, B, np.float32), weights=arr(ptrs[3], C.c_float, B, np.float32),
                   provider=arr(ptrs[4], C.c_int32, B, np.int32), sample=arr(ptrs[5], C.c_int32, B, np.int32))
        self.nl.check(self.nl.lib.mww_prefetch_release(self.h))
        return out

    def rng_state(self):
        """(py_state, np_state, batches handed out): the streams as they stood after the last batch handed out."""
        py, npst = np.zeros(625, np.uint32), np.zeros(625, np.uint32)
        n = self.nl.lib.mww_prefetch_rng_state(self.h, py.ctypes.data_as(C.c_void_p), npst.ctypes.data_as(C.c_void_p))
        return py, npst, int(n)

    def close(self):
        if getattr(self, "h", None):
            self.nl.lib.mww_prefetch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One device context (``mww_ctx``): model weights, HBM-resident feature stores, the train step."""

    latest = None   # weakref to the engine created last: what a FeatureHandler built without one attaches to at first use

    def __init__(self, frames, conv1_filters=None, conv1_kernel=None, conv1_stride=1, block_filters=(), block_kernel=(),
                 max_batch=1024, device=0, stream=None, lib: Optional[NativeLib] = None, conv_ops=None, dropout=0.0,
                 head_attention=False, head_pool=0):
        """MixedNet topology from the ``conv1_*`` / ``block_*`` arguments, or — when ``conv_ops`` is given —
        a conv/BN graph: a list of dicts ``{src: [...], drop: [...], slice: [(c0, width), ...], kernel, dilation, filters, bn_groups}``
        (``mww_conv_bn_op``) with the classifier head on the last one."""
        self.nl = lib or NativeLib.get()
        self.max_batch = int(max_batch)
        self.frames = int(frames)
        self.device = int(device)
        if conv_ops is not None:
            d = ConvNetDesc()
            d.frames, d.n_ops, d.dropout, d.max_batch = int(frames), len(conv_ops), float(dropout), int(max_batch)
            d.head_attention, d.head_pool = int(bool(head_attention)), int(head_pool)
            if len(conv_ops) > MWW_MAX_GRAPH_OPS:
                raise ValueError("too many ops")
            for i, op in enumerate(conv_ops):
                o = d.ops[i]
                src, drop = list(op["src"]), list(op.get("drop", [0] * len(op["src"])))
                if len(src) > MWW_MAX_OP_SOURCES or len(src) != len(drop):
                    raise ValueError("bad source lists")
                o.n_src = len(src)
                sl = list(op.get("slice", [(0, 0)] * len(src)))   # (first channel, width); width 0 = whole tensor
                for j, (sj, dj) in enumerate(zip(src, drop)):
                    o.src[j], o.src_drop[j] = int(sj), int(dj)
                    o.src_c0[j], o.src_cn[j] = int(sl[j][0]), int(sl[j][1])
                o.kernel, o.dilation, o.filters = int(op["kernel"]), int(op.get("dilation", 1)), int(op["filters"])
                o.bn_groups = int(op.get("bn_groups", 1))
                o.kind = OP_KINDS[op.get("kind", "conv")]
                o.stride = int(op.get("stride", 1))
                o.norm = NORMS[op.get("norm", "bn")]
                o.act = ACTS[op.get("act", "relu")]
                o.residual = 0 if op.get("residual") is None else int(op["residual"]) + 1
                o.residual_drop = int(op.get("residual_drop", 0))
            self.desc = d
            h = C.c_void_p()
            self.nl.check(self.nl.lib.mww_create_convnet(C.byref(d), int(device), C.c_void_p(stream or 0), C.byref(h)))
            self.h = h
            self.n_params = int(self.nl.lib.mww_num_params(h))
            self.n_state = int(self.nl.lib.mww_num_bn_state(h))
            Engine.latest = weakref.ref(self)
            return
        d = MixedNetDesc()
        d.frames, d.conv1_filters, d.conv1_kernel, d.conv1_stride = frames, conv1_filters, conv1_kernel, conv1_stride
        d.n_blocks = len(block_filters)
        if len(block_filters) != len(block_kernel) or len(block_filters) > MWW_MAX_BLOCKS:
            raise ValueError("bad block lists")
        for i, (f, k) in enumerate(zip(block_filters, block_kernel)):
            d.block_filters[i], d.block_kernel[i] = int(f), int(k)
        d.max_batch = int(max_batch)
        self.desc = d
        self.max_batch = int(max_batch)
        self.frames = int(frames)
        h = C.c_void_p()
        self.nl.check(self.nl.lib.mww_create(C.byref(d), int(device), C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.n_params = int(self.nl.lib.mww_num_params(h))
        self.n_state = int(self.nl.lib.mww_num_bn_state(h))
        Engine.latest = weakref.ref(self)

    def close(self):
        if getattr(self, "h", None):
            self.nl.lib.mww_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- vectors
    def _vec_in(self, fn, a, n):
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        if a.size != n:
            raise ValueError("expected %d values, got %d" % (n, a.size))
        self.nl.check(fn(self.h, _fptr(a), n))

    def _vec_out(self, fn, n):
        a = np.empty(n, np.float32)
        self.nl.check(fn(self.h, _fptr(a), n))
        return a

    def set_params(self, a):
        self._vec_in(self.nl.lib.mww_set_params, a, self.n_params)

    def get_params(self):
        return self._vec_out(self.nl.lib.mww_get_params, self.n_params)

    def set_bn_state(self, a):
        self._vec_in(self.nl.lib.mww_set_bn_state, a, self.n_state)

    def get_bn_state(self):
        return self._vec_out(self.nl.lib.mww_get_bn_state, self.n_state)

    def set_grad_mask(self, a):
        self._vec_in(self.nl.lib.mww_set_grad_mask, a, self.n_params)

    def get_grads(self):
        return self._vec_out(self.nl.lib.mww_get_grads, self.n_params)

    def set_opt_state(self, m, v, step):
        m = np.ascontiguousarray(m, np.float32).reshape(-1)
        v = np.ascontiguousarray(v, np.float32).reshape(-1)
        self.nl.check(self.nl.lib.mww_set_opt_state(self.h, _fptr(m), _fptr(v), self.n_params, int(step)))

    def get_opt_state(self):
        m, v = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32)
        step = C.c_int64()
        self.nl.check(self.nl.lib.mww_get_opt_state(self.h, _fptr(m), _fptr(v), self.n_params, C.byref(step)))
        return m, v, int(step.value)

    # ---- data
    def upload_store(self, store_id, flat: np.ndarray):
        flat = np.ascontiguousarray(flat).reshape(-1)
        if flat.dtype == np.uint16:
            dt = DTYPE_U16
        elif flat.dtype == np.float32:
            dt = DTYPE_F32
        else:
            raise ValueError("feature stores must be uint16 or float32")
        self.nl.check(self.nl.lib.mww_upload_store(self.h, int(store_id), flat.ctypes.data_as(C.c_void_p), flat.size, dt))

    def assemble(self, windows: np.ndarray, masks: Optional[np.ndarray], n_time, n_freq):
        windows = np.ascontiguousarray(windows, WINDOW_DTYPE)
        B = windows.shape[0]
        mp = None
        if n_time + n_freq:
            masks = np.ascontiguousarray(masks, np.int32)
            if masks.size != B * (n_time + n_freq) * 2:
                raise ValueError("mask array has the wrong size")
            mp = masks.ctypes.data_as(C.c_void_p)
        self.nl.check(self.nl.lib.mww_assemble_batch(self.h, windows.ctypes.data_as(C.c_void_p), mp, B, n_time, n_freq))
        return B

    def set_batch(self, x):
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 3 or x.shape[1] != self.frames or x.shape[2] != 40:
            raise ValueError("batch must be [B,%d,40], got %s" % (self.frames, x.shape))
        self.nl.check(self.nl.lib.mww_set_batch(self.h, _fptr(x), x.shape[0]))
        return x.shape[0]

    def get_batch(self, B):
        x = np.empty((B, self.frames, 40), np.float32)
        self.nl.check(self.nl.lib.mww_get_batch(self.h, _fptr(x), B))
        return x

    def set_targets(self, y, w):
        y = np.ascontiguousarray(np.asarray(y, np.float32).reshape(-1))
        w = np.ascontiguousarray(np.asarray(w, np.float32).reshape(-1))
        if y.size != w.size:
            raise ValueError("labels and weights differ in length")
        self.nl.check(self.nl.lib.mww_set_targets(self.h, _fptr(y), _fptr(w), y.size))

    def assemble_prefetched(self, pf: "Prefetcher", want_targets=False):
        """The next prefetched batch becomes the engine's current batch (descriptors, labels, weights): one native call,
        no numpy work on the launching thread."""
        if want_targets:
            y, w = np.empty(pf.B, np.float32), np.empty(pf.B, np.float32)
            self.nl.check(self.nl.lib.mww_assemble_prefetched(self.h, pf.h, y.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)))
            return y, w
        self.nl.check(self.nl.lib.mww_assemble_prefetched(self.h, pf.h, None, None))
        return None

    def allreduce_init(self, rank, world, unique_id, sync_bn=False):
        """Join the library-owned RCCL communicator (collective over the ranks; include/mww.h mww_allreduce_init)."""
        uid = np.ascontiguousarray(unique_id, np.uint8)
        if uid.size != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % UNIQUE_ID_BYTES)
        self.nl.check(self.nl.lib.mww_allreduce_init(self.h, int(rank), int(world), uid.ctypes.data_as(C.c_void_p), int(bool(sync_bn))))

    def allreduce_destroy(self):
        self.nl.check(self.nl.lib.mww_allreduce_destroy(self.h))

    def allreduce_world(self) -> int:
        """ranks of the library-owned RCCL communicator as RCCL counts them (``ncclCommCount``); 0 without one"""
        return int(self.nl.check(self.nl.lib.mww_allreduce_world(self.h)))

    def set_dropout_mask(self, keep):
        """``keep`` [B, T_last*C_last] of 0/1 (None = built-in generator)."""
        if keep is None:
            self.nl.check(self.nl.lib.mww_set_dropout_mask(self.h, None, 0))
            return
        k = np.ascontiguousarray(np.asarray(keep) != 0, np.uint8)
        self.nl.check(self.nl.lib.mww_set_dropout_mask(self.h, k.ctypes.data_as(C.c_void_p), k.shape[0]))

    def set_allreduce_hook(self, fn, world_size=1, sync_bn=False, reduce_grads=False):
        """``fn(device_ptr: int, n: int, flags: int) -> None`` must enqueue an in-place sum all-reduce of ``n``
        floats at ``device_ptr`` ordered as ``flags`` says (``EXCHANGE_IN_ORDER`` / ``_DEFERRED`` / ``_FLUSH``,
        include/mww.h; see ``parallel.DataParallel``); ``None`` removes it."""
        if fn is None:
            self._hook = ALLREDUCE_FN(0)
            self.nl.check(self.nl.lib.mww_set_allreduce_hook(self.h, self._hook, None, 1, 0, 0))
            return

        def tramp(_user, ptr, n, flags):
            try:
                fn(int(ptr or 0), int(n), int(flags))
                return 0
            except Exception:   # a Python exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1

        self._hook = ALLREDUCE_FN(tramp)   # keep the trampoline alive as long as the engine uses it
        self.nl.check(self.nl.lib.mww_set_allreduce_hook(self.h, self._hook, None, int(world_size), int(bool(sync_bn)),
                                                         int(bool(reduce_grads))))

    # ---- compute
    def train_step(self, B, lr, flags=0):
        self.nl.check(self.nl.lib.mww_train_step(self.h, int(B), float(lr), int(flags)))

    def apply_gradients(self, lr, grad_scale=1.0):
        self.nl.check(self.nl.lib.mww_apply_gradients(self.h, float(lr), float(grad_scale)))

    def forward(self, B, training=False, update_metrics=False):
        self.nl.check(self.nl.lib.mww_forward(self.h, int(B), int(bool(training)), int(bool(update_metrics))))

    def evaluate_windows(self, windows: np.ndarray, labels: np.ndarray, batch: int):
        """Inference forward + metric counters over a whole window list in batches of ``batch`` (one native call)."""
        windows = np.ascontiguousarray(windows, WINDOW_DTYPE)
        labels = np.ascontiguousarray(labels, np.float32).reshape(-1)
        if labels.size != windows.shape[0]:
            raise ValueError("one label per window")
        self.nl.check(self.nl.lib.mww_evaluate_windows(self.h, windows.ctypes.data_as(C.c_void_p), _fptr(labels), windows.shape[0], int(batch)))

    def read_outputs(self, B, want_loss=True):
        p, z = np.empty(B, np.float32), np.empty(B, np.float32)
        loss = C.c_float()
        self.nl.check(self.nl.lib.mww_read_outputs(self.h, int(B), _fptr(p), _fptr(z), C.byref(loss) if want_loss else None))
        return p, z, float(loss.value)

    def synchronize(self):
        self.nl.check(self.nl.lib.mww_synchronize(self.h))

    def metrics_raw(self) -> MetricsRaw:
        m = MetricsRaw()
        self.nl.check(self.nl.lib.mww_metrics_read(self.h, C.byref(m)))
        return m

    def metrics_reset(self):
        self.nl.check(self.nl.lib.mww_metrics_reset(self.h))

    def device_ptr(self, which) -> int:
        return int(self.nl.lib.mww_device_ptr(self.h, which) or 0)

    def debug_read(self, name, B, capacity):
        a = np.empty(capacity, np.float32)
        n = self.nl.check(self.nl.lib.mww_debug_read(self.h, name.encode(), int(B), _fptr(a), capacity))
        return a[:n]

    def set_option(self, name, value):
        self.nl.check(self.nl.lib.mww_set_option(self.h, name.encode(), int(value)))

    def profile_read(self, capacity=4096):
        names = C.create_string_buffer(capacity * 24)
        ms = np.empty(capacity, np.float32)
        n = self.nl.check(self.nl.lib.mww_profile_read(self.h, names, len(names), _fptr(ms), capacity))
        nm = names.value.decode().split("\n")[:n]
        return list(zip(nm, ms[:n].tolist()))


_METRIC_SCALARS = ("n", "correct", "tp5", "fp5", "fn5", "pos", "neg", "bce_sum")


class MetricsSum:
    """The fields of ``MetricsRaw`` as float64 (a sum of several contexts' counters)."""


def metrics_to_vector(m) -> np.ndarray:
    """The raw cumulative counters as one float64 vector (610 entries; the integer counters are exact below 2**53), the
    unit of the data-parallel validation's one all-reduce."""
    h101 = np.array([[m.hist101[l][i] for i in range(101)] for l in range(2)], np.float64)
    h200 = np.array([[m.hist200[l][i] for i in range(200)] for l in range(2)], np.float64)
    v = np.concatenate([h101.reshape(-1), h200.reshape(-1), [float(getattr(m, k)) for k in _METRIC_SCALARS]])
    if v[:-1].max(initial=0.0) >= 2.0 ** 53:
        raise OverflowError("metric counters beyond 2**53")
    return v


def metrics_from_vector(v: np.ndarray) -> MetricsSum:
    v = np.asarray(v, np.float64)
    m = MetricsSum()
    m.hist101 = v[:202].reshape(2, 101)
    m.hist200 = v[202:602].reshape(2, 200)
    for k, x in zip(_METRIC_SCALARS, v[602:]):
        setattr(m, k, float(x))
    return m


def metrics_from_raw(m) -> dict:
    """Derives the reference's nine compiled metrics (train.py:209-221) from the raw counters."""
    h101 = np.array([[m.hist101[l][i] for i in range(101)] for l in range(2)], np.float64)
    h200 = np.array([[m.hist200[l][i] for i in range(200)] for l in range(2)], np.float64)

    def div(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.where(b != 0, a / np.where(b != 0, b, 1), 0.0)

    pos, neg = float(m.pos), float(m.neg)
    tp = np.cumsum(h101[1][::-1])[::-1]
    fp = np.cumsum(h101[0][::-1])[::-1]
    tp2 = np.cumsum(h200[1][::-1])[::-1]
    fp2 = np.cumsum(h200[0][::-1])[::-1]
    rec = div(tp2, tp2 + (pos - tp2))
    fpr = div(fp2, fp2 + (neg - fp2))
    auc = float(np.sum((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0))
    return dict(accuracy=float(div(m.correct, m.n)), recall=float(div(m.tp5, m.tp5 + m.fn5)),
                precision=float(div(m.tp5, m.tp5 + m.fp5)), tp=tp, fp=fp, tn=neg - fp, fn=pos - tp, auc=auc,
                loss=float(div(m.bce_sum, m.n)), count=int(m.n))
