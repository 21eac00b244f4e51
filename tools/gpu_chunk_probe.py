"""First GPU run of the frame-chunk kernels ("graph_frame_chunks", added after the GPU budget of round 2 was spent: they
have passed the parity checks on the CPU emulator only).  Run from the repo root on a GPU box:
    python tools/gpu_chunk_probe.py
then time them:  MWW_BENCH_OPTIONS=graph_frame_chunks=1 python bench.py --model inception   (and --force-generic)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import engine_checks as ec  # noqa: E402
from microwakeword_amd import native  # noqa: E402

lib = native.NativeLib.get()
ec.check_graph_mixednet(lib, ec.GRAPH_MIXEDNET, B=8, T=100, steps=2, grid=0, options={"graph_frame_chunks": 3})
print("chunks, MixedNet graph: ok", flush=True)
ec.check_inception_train_steps(lib, B=6, T=150, steps=2, grid=0, options={"graph_frame_chunks": 2})
ec.check_inception_train_steps(lib, B=6, T=194, steps=1, grid=0, options={"graph_frame_chunks": 1})
print("chunks, Inception: ok", flush=True)
ec.check_graph_mixednet(lib, ec.DEF, B=4, T=194, steps=1, grid=0, options={"graph_frame_chunks": 1})
print("chunks, default MixedNet on the generic engine: ok", flush=True)
