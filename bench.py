#!/usr/bin/env python
"""bench.py — spectrogram-windows/sec of the MixedNet train step on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus 8                      # launches its own 8 ranks (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic ragged spectrograms already
resident in HBM: exact-RNG window/mask draw (host C++, continuing the reference's MT19937 streams)
-> HIP batch assembly (gather + pad/truncate + uint16->f32 + SpecAugment) -> forward (batch-stat BN)
-> weighted Keras BCE -> backward -> gradient assembly [-> two-bucket RCCL all-reduce overlapped with the backward
tail] -> Adam, + metric update.
Workload = BASELINE configs[1]: default mixednet (argparse defaults + residual_connection "0,0,0,0"),
T=194, batch 1024 per GPU, fp32; weak scaling (per-GPU batch fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed in this process)
and, at N=1, `cpu_baseline` (the oracle port timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_FRAMES = 194
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
BYTES_PER_WINDOW_STEP = 793216  # fp32 algorithmic bytes per window for the whole train step

# per-kernel share of the §8(d) element accounting (elements per window, fp32 => x4 bytes); DESIGN.md §5
P_ELEMS = {1: 188 * 48, 2: 180 * 48, 3: 168 * 48, 4: 148 * 48}
X_ELEMS = 194 * 40
# The first block's kernels gather their rows straight from the uint16 feature stores ("fused_input", the default):
# x costs 2 B/elem there and no assembly kernel runs; with MWW_BENCH_FUSED_INPUT=0 the assembly kernel writes fp32 x.
FUSED_INPUT = os.environ.get("MWW_BENCH_FUSED_INPUT", "1") != "0"
X_READ = X_ELEMS // 2 if FUSED_INPUT else X_ELEMS   # in 4-byte units
KERNEL_ELEMS = {
    "assemble": X_ELEMS // 2 + X_ELEMS,  # uint16 source read (2 B/elem) + fp32 write, in 4-byte units
    "fwd_block1": X_READ + P_ELEMS[1],
    "fwd_block2": P_ELEMS[1] + P_ELEMS[2],
    "fwd_block3": P_ELEMS[2] + P_ELEMS[3],
    "fwd_block4": P_ELEMS[3] + P_ELEMS[4],
    "head": P_ELEMS[4],
    "bwd_block4": P_ELEMS[3] + P_ELEMS[4] + P_ELEMS[3],              # R p3, R p4, W g3
    "bwd_block3": P_ELEMS[2] + P_ELEMS[3] + P_ELEMS[3] + P_ELEMS[2],  # R p2, R p3, R g3, W g2
    "bwd_block2": P_ELEMS[1] + P_ELEMS[2] + P_ELEMS[2] + P_ELEMS[1],
    "bwd_block1": X_READ + P_ELEMS[1] + P_ELEMS[1],                   # R x, R p1, R g1
}




def kernel_elems_stored_bf16():
    """The same accounting with p_k / g_k held as bf16 ("storage_bf16"): their elements cost 2 B, the input rows
    keep their cost.  Returned in the 4-byte units of KERNEL_ELEMS."""
    xpart = {"assemble": KERNEL_ELEMS["assemble"], "fwd_block1": X_READ, "bwd_block1": X_READ}
    return {k: xpart.get(k, 0) + (v - xpart.get(k, 0)) / 2 for k, v in KERNEL_ELEMS.items()}


BYTES_PER_WINDOW_STEP_BF16 = 427648   # bf16 p_k / g_k, fp32 input


# exact-fp32 MFMA flops per window of the default MixedNet's GEMM-shaped phases: first conv as
# im2col GEMM, the 1x1 convolutions; the backward kernels run the 1x1 twice (weight + data gradient) and
# bwd_block1 forms the first conv's weight gradient (its input relu(conv1(x)) is read back, not recomputed: every
# flop counted here is a useful one)
FP32_MFMA_PEAK = 157.3e12  # FLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
CONV1_FLOPS, PW_FLOPS = 1474560, {1: 577536, 2: 829440, 3: 774144, 4: 681984}
# (bwd_block1: since round 6 the conv1 weight gradient - CONV1_FLOPS of it - runs as six bf16 slice products per fp32 product
# on v_mfma_f32_16x16x32_bf16 (option conv1_x6, csrc/common.hip.h): 6 x CONV1_FLOPS bf16 flops at 1/16 of the f32 cost each
# = 0.375 of the exact-fp32 MFMA time, counted here as that many f32-equivalent flops)
KERNEL_MFMA_FLOPS = {
    "fwd_block1": CONV1_FLOPS + PW_FLOPS[1], "fwd_block2": PW_FLOPS[2], "fwd_block3": PW_FLOPS[3], "fwd_block4": PW_FLOPS[4],
    "bwd_block4": 2 * PW_FLOPS[4], "bwd_block3": 2 * PW_FLOPS[3], "bwd_block2": 2 * PW_FLOPS[2],
    "bwd_block1": int(0.375 * CONV1_FLOPS) + 2 * PW_FLOPS[1],
}


def inception_kernel_elems(layout):
    """Algorithmic fp32 elements per window each conv/BN graph kernel must move: every op
    reads its (aligned) sources and writes its pre-BN output; the weight gradient re-reads the sources and
    reads (g, p) of the output; the data gradient reads (g, p) of the output and, per source, reads p and
    writes g (+ reads g when it accumulates after an earlier consumer)."""
    ops = layout.ops
    elems = {"assemble": X_ELEMS // 2 + X_ELEMS}
    consumers = {}
    for i, op in enumerate(ops):
        for s in op["src"]:
            if s >= 0:
                consumers.setdefault(s, []).append(i)

    def width(op, j):
        s = op["src"][j]
        return op["slice"][j][1] or (40 if s < 0 else ops[s]["filters"])

    def src_elems(op, full):
        n = 0
        for j, (s, d) in enumerate(zip(op["src"], op["drop"])):
            t = layout.frames if s < 0 else ops[s]["tout"]
            e = (t if full else t - d) * width(op, j)
            # (the stem gathers the spectrogram from the uint16 stores: 2 bytes per element, as in the MixedNet first block)
            n += e // 2 if (s < 0 and FUSED_INPUT and layout.frames <= 200) else e
        return n

    for i, op in enumerate(ops):
        out = op["tout"] * op["filters"]
        elems["conv_fwd%d" % (i + 1)] = src_elems(op, False) + out
        elems["conv_wgrad%d" % (i + 1)] = src_elems(op, False) + 2 * out
        if any(s >= 0 for s in op["src"]):
            n = 2 * out
            for j, s in enumerate(op["src"]):
                if s >= 0:
                    e = ops[s]["tout"] * width(op, j)
                    same = [i2 for i2 in consumers[s] if any(ops[i2]["src"][j2] == s and ops[i2]["slice"][j2] == op["slice"][j]
                                                             for j2 in range(len(ops[i2]["src"])))]
                    n += 2 * e + (e if i != max(same) else 0)
            elems["conv_dgrad%d" % (i + 1)] = n
            elems["conv_bwd%d" % (i + 1)] = n + elems["conv_wgrad%d" % (i + 1)]   # both halves in one launch
    last = ops[-1]["tout"] * ops[-1]["filters"]
    elems["head"] = 2 * last + (last if layout.dropout > 0 else 0)          # read p, write g (+ keep mask)
    elems["dense_grad"] = last + (last if layout.dropout > 0 else 0)
    if layout.dropout > 0:
        elems["dropout_mask"] = last
    return elems


PMC_FILE = "round6_kernel_stats_and_pmc.txt"   # tools/gpu_session.sh pmc step (trace + four PMC passes) on the kernel binary of this round
LIBRARY = os.environ.get("MWW_HIP_LIB") or os.path.join(ROOT, "microwakeword_amd", "libmww_hip.so")   # the file native.NativeLib.get() loads


def library_sha16(path=LIBRARY):
    import hashlib
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def _source_sha_of_loaded_library():
    try:
        from microwakeword_amd import build_native
        return build_native.library_source_sha16(LIBRARY)
    except Exception:   # noqa: BLE001 - a bench line must not die on its provenance fields
        return None


def _tree_source_sha():
    try:
        from microwakeword_amd import build_native
        return build_native.source_sha16()
    except Exception:   # noqa: BLE001
        return None


def pmc_profile_sha16(path, field="library sha256_16"):
    """tools/pmc_summary.py stamps the library it profiled into the first line of its summary
    (`# library sha256_16=... source_sha16=...`: the file's sha256 and the sha256 of the source set it was built from)."""
    import re
    try:
        with open(path) as fh:
            m = re.search(re.escape(field) + r"=([0-9a-f]{16})", fh.readline())
        return m.group(1) if m else None
    except OSError:
        return None


def pmc_traffic(kernel, model, path=None, library=LIBRARY):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this round (profiles/round3_*:
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each in its own run of `bench.py --no-graphs`), corrected
    as MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE (KB) counts half the bytes of wide coalesced
    reads -> x2; WRITE_SIZE (KB) as reported.  Returns (bytes, source) or (None, None).  The counters cannot be
    read from inside this process; the figure belongs to the kernel binary profiled at the end of the round (it includes
    the 24.6 KB/window of a0 = relu(conv1(x)) that fwd_block1 stores and bwd_block1 reads back, which SURVEY 8(d)'s
    algorithmic bytes do not count).  The summary carries the sha256 of the library it was measured on; when that is not the
    library this process loaded, the figure is NOT reported (traffic null, the source says why)."""
    import re
    if model == "inception":   # the two stem launches have kernels of their own (every other symbol serves several ops)
        path = path or os.path.join(ROOT, "profiles", PMC_FILE.replace(".txt", "_inception.txt"))
        names = {"conv_fwd1": "gconv_xg_kernel<24,", "conv_wgrad1": "gconv_wgrad_xg_kernel<24,"} if FUSED_INPUT else {}
    elif model == "mixednet":
        path = path or os.path.join(ROOT, "profiles", PMC_FILE)
        names = {"bwd_block1": "bwd_first_kernel<", "fwd_block1": r"fwd_first_kernel<", "fwd_block2": r"fwd_block_kernel<48, 48, 9,",
                 "fwd_block3": r"fwd_block_kernel<48, 48, 13,", "fwd_block4": r"fwd_block_kernel<48, 48, 21,",
                 # (the block backward is bwd_blockw_kernel by default, bwd_block_kernel with the option "bwd_wide" 0: one prefix serves both)
                 "bwd_block2": r"bwd_block(w?)_kernel<48, 48, 9,", "bwd_block3": r"bwd_block(w?)_kernel<48, 48, 13,",
                 "bwd_block4": r"bwd_block(w?)_kernel<48, 48, 21,", "assemble": r"assemble_kernel", "head": r"head_kernel<"}
    else:
        return None, None
    if not os.path.isfile(path):
        return None, "missing: no PMC summary of this round's library yet (%s)" % os.path.basename(path)
    prof_sha, lib_sha = pmc_profile_sha16(path), library_sha16(library)
    same_binary = prof_sha is not None and prof_sha == lib_sha
    if not same_binary:
        # a library rebuilt from the same source set holds the same kernels (hipcc builds are not bit-reproducible across
        # paths): accepted, and the source says so
        prof_src = pmc_profile_sha16(path, "source_sha16")
        try:
            from microwakeword_amd import build_native
            lib_src = build_native.library_source_sha16(library)
        except Exception:   # noqa: BLE001
            lib_src = None
        if prof_src is None or lib_src is None or prof_src != lib_src:
            return None, "stale: profile sha %s != library sha %s (%s)" % (prof_sha, lib_sha, os.path.basename(path))
    want = names.get(kernel)
    if not want:
        return None, None
    fetch = write = None
    for line in open(path):
        if re.match(want, line):
            m = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            fetch = float(m.group(1)) if m else fetch
            m = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            write = float(m.group(1)) if m else write
    if fetch is None or write is None:
        return None, None
    which = "library sha %s" % lib_sha if same_binary else "a library built from the same source set %s (binary sha %s, profiled %s)" % (prof_src, lib_sha, prof_sha)
    return int(2 * fetch * 1024 + write * 1024), "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, KB; %s)" % (os.path.basename(path), which)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU per step (weak scaling: fixed as N grows)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling (BASELINE configs[4]'s sweep: 4096 windows per step whatever N): windows per step over ALL GPUs, "
                         "--batch becomes global / N; refused when N does not divide it")
    ap.add_argument("--range-repeats", type=int, default=5,
                    help="N = 1: this many further K-step regions are timed AFTER the K steps `value` comes from (`value_range`, not part of `value`)")
    ap.add_argument("--model", choices=("mixednet", "inception", "notebook"), default="mixednet",
                    help="mixednet = BASELINE configs[1] (the headline workload); inception = configs[3] topology; notebook = the "
                         "MixedNet flags of the reference's training notebook (5x1 stride-3 first conv, 64 filters, MixConv groups, T=204)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="multi-GPU parity mode: BatchNorm statistics exchanged over RCCL (default: local-BN throughput mode)")
    ap.add_argument("--storage-bf16", action="store_true",
                    help="BASELINE configs[4], full form: p_k / g_k stored as bf16 in HBM on top of --pointwise-bf16 (fp32 accumulation and BN sums)")
    ap.add_argument("--pointwise-bf16", action="store_true",
                    help="BASELINE configs[4]: bf16-operand MFMA for the 1x1 contractions (not the headline configuration)")
    ap.add_argument("--force-generic", action="store_true",
                    help="run the default mixednet on the generic conv/BN graph kernels (what unusual MixedNet shapes fall back to)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the step from a hipGraph instead of launching its 20 kernels eagerly (measured 2 % slower: the host "
                         "needs 0.2 ms per step and stays ahead of the 0.43 ms the GPU needs)")
    ap.add_argument("--no-graphs", action="store_true", help="accepted for older scripts: eager launches are the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validation", action="store_true", help="skip the (untimed-for-value) validation-throughput leg")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the extra batch-4096 point (`batch_sweep`, after the timed region)")
    ap.add_argument("--store-samples", type=int, default=4096)
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--grid-fwd", type=int, default=0)
    ap.add_argument("--grid-bwd", type=int, default=0)
    ap.add_argument("--grid-head", type=int, default=0)
    ap.add_argument("--ablate", type=int, default=0, help="profiling only: skip kernel phases (results invalid)")
    ap.add_argument("--grad-buckets", type=int, default=1, choices=(1, 2),
                    help="N > 1: 1 = one gradient all-reduce after the backward pass (default: the faster schedule at W = 1, the only one measured "
                         "so far); 2 = two buckets, the first one overlapped with the backward tail")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="N > 1: issue the exchange through the mww_set_allreduce_hook callback into torch.distributed instead of "
                         "RCCL called from the library (mww_allreduce_init, the default)")
    ap.add_argument("--no-prefetch", action="store_true", help="draw every batch on the launching thread (default: a worker thread draws four batches ahead)")
    return ap.parse_args()


def per_gpu_batch(batch, global_batch, world):
    """Windows per GPU and step: --batch (weak scaling), or --global-batch / N (strong scaling: BASELINE configs[4]'s sweep keeps 4096
    windows per step whatever N) - refused when N does not divide it."""
    if global_batch:
        if global_batch % world:
            raise SystemExit("--global-batch %d is not divisible by the %d GPUs of the job" % (global_batch, world))
        return global_batch // world
    return batch


def _reference_loader(batch, n_samples, policy):
    """the reference's UNMODIFIED FeatureHandler.get_data (microwakeword/data.py:497-597), imported
    through oracle/ref_data_shim over the same synthetic stores - only where /root/reference exists (the build
    container; never on the GPU box).  Returns a zero-argument callable or None."""
    try:
        from oracle import ref_data_shim as shim
        if not shim.available():
            return None
        import tempfile

        from microwakeword_amd.ragged import write_ragged_store
        from oracle import data_oracle as do
        ref = shim.load_reference_data_module()
        pos, neg = do.synthetic_stores(n_samples, 1234)
        tmp = tempfile.mkdtemp(prefix="mww_cpu_baseline_")
        write_ragged_store(os.path.join(tmp, "pos", "training", "a_mmap"), pos)
        write_ragged_store(os.path.join(tmp, "neg", "training", "a_mmap"), neg)
        config = {"stride": 1, "window_step_ms": 10, "features": [
            dict(type="mmap", features_dir=os.path.join(tmp, "pos"), truth=True, sampling_weight=2.0, penalty_weight=1.0, truncation_strategy="truncate_start"),
            dict(type="mmap", features_dir=os.path.join(tmp, "neg"), truth=False, sampling_weight=10.0, penalty_weight=1.0, truncation_strategy="random")]}
        fh = ref.FeatureHandler(config)
        return lambda B: fh.get_data("training", B, T_FRAMES, "default", policy)
    except Exception as e:   # the baseline must never take the benchmark down
        sys.stderr.write("[cpu_baseline] reference loader unavailable: %r\n" % (e,))
        return None


def cpu_baseline(batch, budget_s=24.0, model="mixednet"):
    """The CPU train.py loop of SURVEY 8(d) on this box's host cores, on a bounded sample, rank 0 / N=1 only:
    loader (the reference's own data.py when /root/reference is present, else the oracle's restatement of it) +
    the torch-CPU fp32 restatement of train_on_batch, at the reference's plumbing batch (32) and at the headline
    batch.  kind = "port": TensorFlow is not installable here, so the model half can never be the reference itself."""
    import torch

    from oracle import data_oracle as do
    from oracle import model_oracle as mo

    # torch's intra-op pool stops scaling (and then collapses) far below the 256 hardware threads of the
    # GPU box's host for these small convolutions; 32 threads is the fastest setting measured
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    flags = dict(mo.MIXEDNET_DEFAULTS, residual_connection="0,0,0,0")
    random.seed(0)
    np.random.seed(0)
    n_samples = 512
    if model == "inception":
        flags = dict(mo.INCEPTION_DEFAULTS)
    elif model == "notebook":
        from microwakeword_amd import synthetic
        flags = dict(synthetic.NOTEBOOK_MIXEDNET_FLAGS)
    om = mo.OracleModel("inception" if model == "inception" else "mixednet", flags, T_FRAMES, seed=42, dtype=torch.float32)
    n_keep = (T_FRAMES - mo.inception_slices_dropped(flags)) * 16 if model == "inception" else 0
    pol = dict(time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)
    ref_get = _reference_loader(batch, n_samples, pol)
    provs = do.synthetic_providers(n_samples, 1234)

    def load(B):
        if ref_get is not None:
            return ref_get(B)
        return do.get_data(provs, "training", B, T_FRAMES, "default", pol)[:3]

    def keep_mask(B):
        return (np.random.default_rng(0).random((B, n_keep)) >= 0.2).astype(np.float32) if n_keep else None

    def run(B, share, max_steps):
        x, y, w = load(B)     # one untimed step (allocator / thread-pool warm-up)
        om.train_step(x, y, w, 1e-3, dropout_mask=keep_mask(B))
        t_load = t_model = 0.0
        n = 0
        t_start = time.perf_counter()
        while n < 1 or ((time.perf_counter() - t_start) < share and n < max_steps):
            t0 = time.perf_counter()
            x, y, w = load(B)
            t1 = time.perf_counter()
            om.train_step(x, y, w, 1e-3, dropout_mask=keep_mask(B))
            t2 = time.perf_counter()
            t_load += t1 - t0
            t_model += t2 - t1
            n += 1
        return {"steps": n, "windows_per_s": round(n * B / (t_load + t_model), 1), "loader_windows_per_s": round(n * B / t_load, 1),
                "model_windows_per_s": round(n * B / t_model, 1), "loader_s": round(t_load, 3), "model_s": round(t_model, 3)}

    small = run(32, 0.15 * budget_s, 40)
    big = small if batch == 32 else run(batch, 0.45 * budget_s, 8)
    return {"value": big["windows_per_s"], "unit": "windows/s", "cores": cores, "kind": "port",
            "loader_kind": "reference" if ref_get is not None else "port",
            "loader": "reference data.py (unmodified, via oracle/ref_data_shim)" if ref_get is not None else "oracle/data_oracle.py (restatement; /root/reference is not on this box)",
            "model": "oracle/model_oracle.py torch-CPU fp32 fwd/bwd/Keras-Adam",
            "sample": "%d train steps of batch %d (loader %.2fs + model %.2fs) and %d of batch 32, %d torch threads of %d host threads"
                      % (big["steps"], batch, big["loader_s"], big["model_s"], small["steps"], cores, os.cpu_count() or 1),
            "loader_windows_per_s": big["loader_windows_per_s"], "model_windows_per_s": big["model_windows_per_s"],
            "loader_note": None if ref_get is not None else "loader_windows_per_s is the ORACLE'S restatement of get_data (vectorised numpy), not the "
                           "reference's loader: the reference's own data.py measured 7.2 k windows/s at batch 1024 in the build container (BASELINE.md section 2)",
            "by_batch": {"32": small, str(batch): big}}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside a launcher: become the launcher (one rank per GPU over RCCL)."""
    import socket
    import subprocess

    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible; refusing to report a smaller job under that label" % (args.gpus, ndev))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    global T_FRAMES
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist

    from microwakeword_amd import native, synthetic
    from microwakeword_amd.data import FeatureHandler
    from microwakeword_amd.model import Model
    from microwakeword_amd.parallel import DataParallel, shard_feature_handler

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dp = os.environ.get("MWW_BENCH_FORCE_DP") == "1"   # exercise the collective path on a 1-GPU box
    real_stdout = None
    if world > 1 or force_dp:
        # RCCL prints a version banner on the C stdout; keep the process' fd 1 for the ONE JSON line
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    if world > 1 or force_dp:
        # 88 KB of gradient per step: one channel moves it as fast as many and takes fewer CUs from the backward kernels
        # the first bucket overlaps (override with NCCL_MAX_NCHANNELS)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    stream = torch.cuda.Stream(device=device)
    B = args.batch = per_gpu_batch(args.batch, args.global_batch, world)

    with torch.cuda.stream(stream):
        if args.model == "inception":
            from microwakeword_amd import inception
            model = inception.model(dict(synthetic.DEFAULT_INCEPTION_FLAGS), (T_FRAMES, 40), B, device=local_rank, stream=stream.cuda_stream,
                                    seed=42, max_batch=B)
            kernel_elems = inception_kernel_elems(model.layout)
            step_bytes = 1222208   # algorithmic bytes per window of the default Inception train step (fp32, T = 194)
        elif args.model == "notebook":
            from microwakeword_amd import mixednet
            T_FRAMES = int(os.environ.get("MWW_BENCH_T", "204"))   # (204 = the notebook's clip length; the override is for tile-shape experiments)
            model = mixednet.model(synthetic.NOTEBOOK_MIXEDNET_FLAGS, (T_FRAMES, 40), B, device=local_rank, stream=stream.cuda_stream,
                                   seed=42, max_batch=B)
            lay = model.layout
            # same accounting as SURVEY 8(d): x + 2 p_k forward; 2 p_k + x + 2 (g_1..g_{L-1}) backward
            pk = [b.tout * b.cout for b in lay.blocks]
            step_bytes = 4 * (2 * T_FRAMES * 40 + 4 * sum(pk) + 2 * sum(pk[:-1]))
            kernel_elems = {"assemble": T_FRAMES * 40 * 3 // 2}
        elif args.force_generic:
            from microwakeword_amd.layout import GraphMixedNetLayout
            model = Model(synthetic.DEFAULT_MIXEDNET_FLAGS, (T_FRAMES, 40), B, device=local_rank, stream=stream.cuda_stream, seed=42,
                          max_batch=B, layout=GraphMixedNetLayout(synthetic.DEFAULT_MIXEDNET_FLAGS, T_FRAMES), name="mixednet (generic)")
            model.engine.set_grad_mask(model.layout.grad_mask())
            kernel_elems, step_bytes = {"assemble": KERNEL_ELEMS["assemble"]}, BYTES_PER_WINDOW_STEP
        else:
            model = Model(synthetic.DEFAULT_MIXEDNET_FLAGS, (T_FRAMES, 40), B, device=local_rank, stream=stream.cuda_stream,
                          seed=42, max_batch=B)
            kernel_elems, step_bytes = KERNEL_ELEMS, BYTES_PER_WINDOW_STEP
            if args.storage_bf16:
                kernel_elems, step_bytes = kernel_elems_stored_bf16(), BYTES_PER_WINDOW_STEP_BF16
        eng = model.engine
        n_val = 4096 if (world == 1 and not force_dp and not args.no_validation) else 0
        cfg, _ = synthetic.benchmark_config(args.store_samples, 1234, n_val=n_val, n_ambient=n_val // 8)
        random.seed(0)
        np.random.seed(0)
        fh = FeatureHandler(cfg, engine=eng)
        if world > 1 or force_dp:
            shard_feature_handler(fh, rank, world, seed=0)
        else:
            fh.use_private_rng()
        if args.no_prefetch:
            fh.use_private_rng(prefetch=0)
        dp = None
        if world > 1 or force_dp:
            dp = DataParallel.for_engine(eng, device, sync_bn=args.sync_bn, grad_buckets=args.grad_buckets,
                                         library_comm=not args.torch_collectives)
            dp.broadcast_parameters(0)
        for opt, v in (("grid_fwd", args.grid_fwd), ("grid_bwd", args.grid_bwd), ("grid_head", args.grid_head)):
            if v:
                eng.set_option(opt, v)
        if args.ablate:
            eng.set_option("ablate", args.ablate)
        if args.pointwise_bf16:
            eng.set_option("pointwise_bf16", 1)
        if args.storage_bf16:
            eng.set_option("storage_bf16", 1)
        # engine options for A/B sweeps: MWW_BENCH_OPTIONS="name=value,..." (tools/: e.g. bwd_wide=0, bn_inline=0, grid_graph=768) - the one
        # knob besides MWW_BENCH_FUSED_INPUT, which also changes the byte accounting above
        for kv in filter(None, os.environ.get("MWW_BENCH_OPTIONS", "").split(",")):
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        if os.environ.get("MWW_BENCH_FUSED_INPUT") is not None:
            eng.set_option("fused_input", int(os.environ["MWW_BENCH_FUSED_INPUT"]))
        if args.graphs and not args.no_graphs:
            eng.set_option("graphs", 1)
        policy = synthetic.SPEC_AUGMENT_POLICY
        lr = 1e-3

        def next_batch():
            fh.next_training_batch_on_device(B, T_FRAMES, "default", policy)  # class weights 1/1 (train.py:176-187 defaults)

        if (world > 1 or force_dp) and os.environ.get("MWW_BENCH_DP_PREFETCH", "0") == "1":
            # opt-in: software-pipelined by one batch — every step still draws + assembles exactly one batch, but
            # that batch is the NEXT step's, gathered while RCCL reduces this step's gradient.  Measured at W=1
            # (all-reduce ~free) it costs 10 us/step, so it is off until it can be measured at W>1.
            next_batch()

            def one_step():
                dp.train_step(B, lr, prefetch=next_batch)
        elif world > 1 or force_dp:
            def one_step():
                next_batch()
                dp.train_step(B, lr)
        else:
            host_parts = [0.0, 0.0]   # launching thread, seconds inside the two native calls of a step

            def one_step():
                h0 = time.perf_counter()
                next_batch()
                h1 = time.perf_counter()
                eng.train_step(B, lr)
                host_parts[0] += h1 - h0
                host_parts[1] += time.perf_counter() - h1

        eng.synchronize()
        t_pre_roll = time.perf_counter()
        # ---- the train kernels' first launches happen here, FIRST in the pre-roll: the library is several code objects (one per
        # translation unit, ~10 MB together) that the runtime loads at the first launch out of each - tens of milliseconds of an
        # idle GPU.  Behind the validation leg / settle below (where the round-4 single-object library loaded everything at its
        # very first launch) that idle time sat right in front of the W warm-up steps and a 5 + 20-step run read 0.33 ms
        # against 0.303 over 200 steps; in front of them the device is back at its steady clocks when the warm-up starts.
        def untimed_step(profiled_rank0_only=False):
            fh.next_training_batch_on_device(B, T_FRAMES, "default", policy)
            if dp is not None and world == 1:
                dp.train_step(B, lr)          # forced-DP on one GPU: the exchanges are degenerate but present
            elif dp is not None:
                # no collective may be issued outside the timed loop's lockstep: forward + backward without the exchange /
                # Adam (every rank still starts the timed loop from the broadcast weights)
                eng.train_step(B, lr, flags=native.STEP_NO_APPLY)
            else:
                eng.train_step(B, lr)

        if args.profile_steps > 0:
            eng.set_option("graphs", 0)
            for _ in range(2):
                untimed_step()
            if args.graphs and not args.no_graphs:
                eng.set_option("graphs", 1)

        # ---- validation leg (N=1 only, not part of `value`): validate_nonstreaming's two
        # passes (validation set, truncate_start; ambient set, 100 ms-stride split) with the windows gathered and
        # scored in HBM, threshold counters accumulated on the device.  It runs BEFORE the timed train steps (as does the
        # per-kernel event pass below): together they are ~0.1 s of GPU work, after which the device is at its steady clocks -
        # the W warm-up steps the driver asks for (5 = 2 ms) are not, and the same binary measured 0.362 ms/step in a
        # 5 + 20-step run against 0.342 in a 20 + 200-step run (profiles/round3_*: the per-kernel event times are 2-3 %
        # longer, the rest is the pipeline fill of the first step).  K and W themselves are exactly what was asked for.
        validation = None
        if n_val and rank == 0:
            for mode, strat in (("validation", "truncate_start"), ("validation_ambient", "split")):   # warm-up: index build, caches
                fh.evaluate_on_device(model, mode, T_FRAMES, strat, 1024)
            eng.synchronize()
            reps, best = 3, None
            for _ in range(reps):
                tv0 = time.perf_counter()
                nv1, _, _ = fh.evaluate_on_device(model, "validation", T_FRAMES, "truncate_start", 1024)
                eng.synchronize()
                tv1 = time.perf_counter()
                nv2, _, res = fh.evaluate_on_device(model, "validation_ambient", T_FRAMES, "split", 1024)
                eng.synchronize()
                tv2 = time.perf_counter()
                if best is None or tv2 - tv0 < best[2] - best[0]:
                    best = (tv0, tv1, tv2)
            tv0, tv1, tv2 = best
            validation = {"windows": int(nv1 + nv2), "windows_per_s": round((nv1 + nv2) / (tv2 - tv0), 1),
                          "validation_set": {"windows": int(nv1), "s": round(tv1 - tv0, 4)},
                          "ambient_split": {"windows": int(nv2), "s": round(tv2 - tv1, 4)},
                          "note": "cached window index -> one mww_evaluate_windows call per set: descriptor upload + HBM gather + inference "
                                  "forward + threshold metrics in batches of 1024, metric read-back included; best of %d" % reps}

        # device settle: inference forwards on the last batch (no weights, statistics or RNG touched) until ~80 ms of GPU work
        # have gone by in total, so that a run with a 2 ms warm-up starts its timed region at the same clocks as a long one
        # (skipped with --profile-steps 0, the form the rocprofv3 passes use: their per-kernel averages then hold train steps only)
        if args.profile_steps > 0:
            next_batch()
        eng.synchronize()
        t_settle = time.perf_counter()
        while args.profile_steps > 0 and time.perf_counter() - t_settle < float(os.environ.get("MWW_BENCH_SETTLE_S", "0.08")):
            for _ in range(16):
                eng.forward(B, training=False)
            eng.synchronize()

        def fence():
            eng.synchronize()
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize(device)

        eng.synchronize()
        pre_roll_s = time.perf_counter() - t_pre_roll
        for _ in range(args.warmup):
            one_step()
        fence()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        if dp is None:
            host_parts[0] = host_parts[1] = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        ev1.record(stream)
        host_split = None if dp is not None else {"assemble_prefetched_ms": round(1e3 * host_parts[0] / args.steps, 4),
                                                   "train_step_ms": round(1e3 * host_parts[1] / args.steps, 4)}
        host_enqueue = time.perf_counter() - t0   # the host's share: sampler + launches (+ exchange hooks); it must stay below the GPU's
        fence()
        elapsed = time.perf_counter() - t0
        gpu_ms = ev0.elapsed_time(ev1)
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        _, _, last_loss = eng.read_outputs(B)

        # ---- the launching thread's own cost per step, free of back-pressure: the mailbox ring holds eight steps, so in any run
        # longer than that the enqueue loop above is throttled to the GPU's pace (mail_begin waits for the slot's previous step)
        # and host_enqueue_ms_per_step tends to ms_per_step.  Six steps from an idle device and an empty ring measure the host alone.
        host_unblocked = None
        if world == 1 and dp is None:
            fence()
            host_parts[0] = host_parts[1] = 0.0
            tu0 = time.perf_counter()
            for _ in range(6):
                one_step()
            host_unblocked = {"ms_per_step": round(1e3 * (time.perf_counter() - tu0) / 6, 4), "steps": 6,
                              "assemble_prefetched_ms": round(1e3 * host_parts[0] / 6, 4), "train_step_ms": round(1e3 * host_parts[1] / 6, 4)}
            fence()

        # ---- the spread of the figure: further K-step regions of the same loop, timed the same way AFTER the region `value` comes
        # from (a 5 + 20-step run times 6 ms; fifteen of them on five boxes spanned 0.298-0.306 ms in round 5).  Not part of `value`.
        value_range = None
        if world == 1 and dp is None and args.range_repeats > 0:
            reps = []
            for _ in range(args.range_repeats):
                fence()
                tr0 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                fence()
                reps.append((time.perf_counter() - tr0) / args.steps)
            value_range = {"repeats": args.range_repeats, "steps_each": args.steps,
                           "ms_per_step": {"min": round(1e3 * min(reps), 4), "median": round(1e3 * float(np.median(reps)), 4), "max": round(1e3 * max(reps), 4)},
                           "value": {"min": round(B / max(reps), 1), "median": round(B / float(np.median(reps)), 1), "max": round(B / min(reps), 1)},
                           "note": "further K-step regions timed like the one `value` comes from, after it; `value` itself is the first region only"}

        # ---- per-kernel durations with HIP events on the engine's stream (eager launches, a separate pass AFTER the timed region:
        # the device is at the clocks of the timed steps and no kernel is on its first launch; until round 4 this pass ran in
        # front of the timed region, on a colder device, and read 2-8 % above the rocprofv3 averages of profiles/)
        prof = {}
        if args.profile_steps > 0 and rank == 0:
            eng.set_option("graphs", 0)
            eng.set_option("profile", 1)
            for _ in range(args.profile_steps):
                untimed_step()
            for name, ms in eng.profile_read():
                prof.setdefault(name, []).append(ms)
            eng.set_option("profile", 0)
            if args.graphs and not args.no_graphs:
                eng.set_option("graphs", 1)

        # ---- one more point of the batch sweep in the driver's own line (after the timed region, not part of `value`): the
        # same step at batch 4096 on a second context.  ~118 us of the step do not scale with the batch, so the
        # roofline fraction of the step is a function of the batch; this is the point the 40 % claim refers to.
        batch_sweep = None
        if (world == 1 and not force_dp and args.model == "mixednet" and B == 1024 and not args.no_batch_sweep and not args.force_generic
                and not (args.pointwise_bf16 or args.storage_bf16) and args.profile_steps > 0):
            # (the context is new and the device idled while it was built: ~80 steps = 80 ms of warm-up bring the clocks back)
            Bs, Ks, Ws = 4096, 40, 80
            m2 = Model(synthetic.DEFAULT_MIXEDNET_FLAGS, (T_FRAMES, 40), Bs, device=local_rank, stream=stream.cuda_stream, seed=42, max_batch=Bs)
            fh._drop_prefetcher()
            fh2 = FeatureHandler(cfg, engine=m2.engine)
            fh2.use_private_rng()
            for k in range(Ws + Ks):
                if k == Ws:
                    m2.engine.synchronize()
                    ts0 = time.perf_counter()
                fh2.next_training_batch_on_device(Bs, T_FRAMES, "default", policy)
                m2.engine.train_step(Bs, lr)
            m2.engine.synchronize()
            dts = (time.perf_counter() - ts0) / Ks
            batch_sweep = {str(Bs): {"value": round(Bs / dts, 1), "ms_per_step": round(1e3 * dts, 4), "steps": Ks, "warmup": Ws,
                                     "step_frac": round(Bs / dts * step_bytes / HBM_PEAK, 4)}}
            fh2._drop_prefetcher()
            m2.engine.close()

        comm_ranks = None
        if dp is not None:
            try:
                comm_ranks = eng.allreduce_world() if getattr(dp, "library_comm", False) else (dist.get_world_size() if dist.is_initialized() else 1)
            except Exception as e:   # noqa: BLE001 - provenance only
                comm_ranks = "unknown (%s)" % e
            if isinstance(comm_ranks, int) and comm_ranks != world:
                raise SystemExit("the step's communicator has %d ranks, the job %d" % (comm_ranks, world))
        if world > 1:
            dist.barrier()

    if rank != 0:
        dist.destroy_process_group()
        return
    windows = B * world * args.steps
    value = windows / elapsed
    kern = {k: float(np.mean(v)) for k, v in prof.items()}
    ksum = sum(kern.values())
    cands = [k for k in kern if k in kernel_elems]
    if cands:
        dominant = max(cands, key=lambda k: kern[k])
        dom_bytes = kernel_elems[dominant] * 4 * B
        achieved = dom_bytes / (kern[dominant] * 1e-3)
    else:  # --profile-steps 0 (e.g. under rocprofv3): whole-step figure only
        dominant, dom_bytes, achieved = "train_step(all kernels)", step_bytes * B, value / world * step_bytes
        kern[dominant] = 1e3 * elapsed / args.steps
    traffic, traffic_src = pmc_traffic(dominant, args.model) if B == 1024 and not (args.pointwise_bf16 or args.storage_bf16) else (None, None)
    # roofline of the dominant kernel.  SURVEY 8(d) names HBM as the governing roofline of the train step; a kernel whose
    # exact-fp32 MFMA time at the spec peak exceeds its HBM time at the spec peak is priced against the MFMA peak
    # instead - and BOTH fractions are always reported, with the flops counted (useful ones only: nothing is recomputed
    # on MFMA since bwd_block1 reads relu(conv1(x)) back)
    hbm_frac = achieved / HBM_PEAK
    roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(hbm_frac, 4), "hbm_achieved_GBps": round(achieved / 1e9, 1), "hbm_frac": round(hbm_frac, 4)}
    mfma_flops = KERNEL_MFMA_FLOPS.get(dominant, 0) * B if (args.model == "mixednet" and not args.force_generic and not (args.pointwise_bf16 or args.storage_bf16)) else 0
    if mfma_flops:
        tf = mfma_flops / (kern[dominant] * 1e-3)
        roof.update({"mfma_achieved_TFLOPs": round(tf / 1e12, 2), "mfma_peak_TFLOPs": FP32_MFMA_PEAK / 1e12, "mfma_frac": round(tf / FP32_MFMA_PEAK, 4),
                     "useful_mfma_flops_per_launch": mfma_flops})
        if mfma_flops / FP32_MFMA_PEAK > dom_bytes / HBM_PEAK:
            roof.update({"bound": "mfma", "achieved": round(tf / 1e12, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(tf / FP32_MFMA_PEAK, 4)})
    if args.pointwise_bf16 or args.storage_bf16:
        # the bf16 modes are governed by neither peak (halving the bytes with bf16 storage does not move the step, and
        # the bf16 matrix time is nearly free): the per-tile chain of VALU depthwise phases, LDS windows and barriers at two
        # workgroups per CU.  The HBM fraction stays in the line (hbm_frac); `bound` says what it is not.
        roof["bound"] = "issue"
        roof["bound_note"] = ("not HBM- and not MFMA-bound: the block backward launches take the same time with bf16 operands and with bf16 "
                              "operands + bf16 storage (half the bytes); the depthwise VALU phases / LDS windows / barriers of a tile govern "
                              "(DESIGN.md 4c, profiles/round5_bf16_modes_ab.txt)")
    out = {
        "metric": "spectrogram-windows/sec (train step) on default %s" % args.model,
        "value": round(value, 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None, "dtype": ("bf16 storage of p_k/g_k, bf16-operand MFMA in the 1x1 contractions, f32 accumulate / BN sums / parameters" if args.storage_bf16 else
                                      "f32 storage/accumulate, bf16-operand MFMA in the 1x1 contractions" if args.pointwise_bf16 else "f32"),
        "data": "synthetic",
        # every tensor, sum and update is fp32; ONE contraction - the conv1 weight gradient of stride-1 first blocks - is formed as six
        # bf16 slice products per fp32 product (the three exact 8-bit slices of each operand's significand, fp32 accumulation:
        # max error 1.08e-7 of sum|x w| against 1.19e-7 for the exact-fp32 MFMA's fma chain, tools/ubench/mfma_bf16x9).
        # engine option conv1_x6 0 (MWW_BENCH_OPTIONS=conv1_x6=0) runs it on the exact-fp32 MFMA: +5.7 us per step
        "precision_note": None if (args.model != "mixednet" or args.force_generic) else
                          "fp32 tensors / sums / updates; the conv1 weight gradient as six exact bf16 slice products per fp32 product with fp32 accumulation "
                          "(fp32-grade: 1.08e-7 vs 1.19e-7 of sum|x w| for the exact-fp32 MFMA; option conv1_x6 0 = exact-fp32 MFMA, +5.7 us/step)",
        "config": {"workload": "default %s (argparse defaults%s), T=194, batch %d/GPU, %s, "
                               "SpecAugment 5/2/5/2, 2 providers x %d ragged uint16 samples resident in HBM"
                               % (args.model, " + residual_connection 0,0,0,0" if args.model == "mixednet" else ", dropout 0.2 from the built-in generator",
                                  B, "bf16 p_k/g_k + bf16 MFMA operands" if args.storage_bf16 else "bf16 MFMA operands" if args.pointwise_bf16 else "fp32",
                                  args.store_samples),
                   "global_batch": B * world, "parallelism": "dp%d" % world, "hip_graph": bool(args.graphs and not args.no_graphs),
                   "bn": ("sync" if args.sync_bn else "local") if (world > 1 or force_dp) else "batch",
                   "sampler": "synchronous" if args.no_prefetch else "worker thread, 4 batches ahead",
                   "collectives": (("torch.distributed via callback" if args.torch_collectives else "RCCL called from the library")
                                   + ", %d gradient bucket(s)" % args.grad_buckets) if (world > 1 or force_dp) else None,
                   # ranks that took part in the step's collectives, as the communicator itself counts them (ncclCommCount of the
                   # library-owned communicator; the process group's size for the callback form): must equal n_gpus
                   "collective_ranks": comm_ranks},
        "roofline": {**roof, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(kern[dominant], 5),
                     "step_frac": round(value / world * step_bytes / HBM_PEAK, 4), "step_bytes_per_window": step_bytes,
                     "step_frac_note": "whole step, per GPU: windows/s x SURVEY 8(d) algorithmic bytes per window / 8.0 TB/s",
                     "kernel_ms": {k: round(v, 5) for k, v in sorted(kern.items())}, "kernel_ms_sum": round(ksum, 4),
                     "kernel_ms_note": "HIP-event pass of %d eager steps AFTER the timed region: every launch sits between two event records on the stream, "
                                       "which adds what the back-to-back launches of the timed loop hide - the sum is %+.1f %% of ms_per_step; "
                                       "kernel_ms_scaled = the same figures scaled to sum to ms_per_step"
                                       % (args.profile_steps, 100.0 * (ksum / (1e3 * elapsed / args.steps) - 1.0)) if cands else None,
                     "kernel_ms_scaled": {k: round(v * (1e3 * elapsed / args.steps) / ksum, 5) for k, v in sorted(kern.items())} if cands and ksum > 0 else None},
        "gpu_stream_ms_per_step": round(gpu_ms / args.steps, 4), "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 4),
        # the launching thread's two native calls per step: mww_assemble_prefetched (wait for the worker's batch + descriptor / target
        # upload) and mww_train_step (ten launches); in runs much longer than the mailbox ring both include back-pressure from the GPU
        "host_enqueue_split": host_split,
        # ... and without it: six steps enqueued from an idle device with an empty ring, after the timed region (not part of `value`)
        "host_enqueue_unblocked": host_unblocked, "final_loss": round(float(last_loss), 5),
        "pre_roll_s": round(pre_roll_s, 3),
        # which binary was timed, and which source set it was built from (mww_version() carries the sha256 of csrc/* +
        # include/mww.h; __graft_entry__.build() rebuilds when it differs from the tree's): library_sha16 ties the line to a
        # file, source_sha16 == tree_source_sha16 ties that file to this tree
        "library_sha16": library_sha16(), "source_sha16": _source_sha_of_loaded_library(), "tree_source_sha16": _tree_source_sha(),
        "pre_roll_note": "GPU work of this process BEFORE the W warm-up steps, outside the timed region: validation leg (3 x both sets), ~80 ms of "
                         "inference forwards (device settle) and two plain train steps (first launches of the train kernels).  It brings the device to its steady "
                         "clocks; without it (--no-validation --profile-steps 0) a 5 + 20-step run reads ~0.36 ms/step instead; the per-kernel HIP-event pass "
                         "(%d train steps) runs AFTER the timed region" % args.profile_steps,
    }
    if value_range is not None:
        out["value_range"] = value_range
    if batch_sweep is not None:
        out["batch_sweep"] = batch_sweep
    if validation is not None:
        out["validation"] = validation
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, model=args.model)
    line = json.dumps(out) + "\n"
    if real_stdout is not None:
        os.write(real_stdout, line.encode())
    else:
        sys.stdout.write(line)
        sys.stdout.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
