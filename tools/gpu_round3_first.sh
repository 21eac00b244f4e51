#!/bin/bash
# First GPU call of the next round: everything that was finished after the GPU budget of round 2 ran out.
#   1. parity of the frame-chunk kernels on the GPU (tools/gpu_chunk_probe.py)
#   2. same-session timing of graph_frame_chunks = 0 / 1 / 2 / 3 on the Inception step and on the default MixedNet forced
#      onto the generic engine
# usage (repo root): bash tools/gpu_round3_first.sh [out dir under gpurun_out]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r3first}
mkdir -p $OUT; cd $R
timeout 300 python tools/gpu_chunk_probe.py > $OUT/chunk_probe.log 2>&1; tail -4 $OUT/chunk_probe.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
A="--no-cpu-baseline --no-validation --profile-steps 0 --steps 200 --warmup 20"
for rep in 1 2; do
  for ch in 0 1 2 3; do
    MWW_BENCH_OPTIONS=graph_frame_chunks=$ch timeout 300 python bench.py --model inception $A 2>/dev/null | line inception_chunks=$ch
    MWW_BENCH_OPTIONS=graph_frame_chunks=$ch timeout 300 python bench.py --force-generic $A 2>/dev/null | line generic_chunks=$ch
  done
done | tee $OUT/chunks_ab.txt
