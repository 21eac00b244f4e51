#!/bin/bash
# host exposure of the step: eager launches vs hipGraph replay, at the headline batch and at a batch so small that the host is the limit
TAG=${1:-hostab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
Q="--no-cpu-baseline --no-validation"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'gpu', d['gpu_stream_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'ksum', d['roofline']['kernel_ms_sum'])"; }
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line eager
  timeout 300 python bench.py --steps 200 --warmup 20 $Q --graphs 2>/dev/null | line graphs
  MWW_BENCH_OPTIONS=fused_stages=1 timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line eager_fused_bwd
  MWW_BENCH_OPTIONS=fused_stages=1 timeout 300 python bench.py --steps 200 --warmup 20 $Q --graphs 2>/dev/null | line graphs_fused_bwd
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line eager_driver_form
  timeout 300 python bench.py --steps 20 --warmup 5 $Q --graphs 2>/dev/null | line graphs_driver_form
done 2>&1 | tee $OUT/host_ab.txt
for rep in 1 2; do
  timeout 300 python bench.py --batch 32 --steps 400 --warmup 40 $Q 2>/dev/null | line eager_b32
  timeout 300 python bench.py --batch 32 --steps 400 --warmup 40 $Q --graphs 2>/dev/null | line graphs_b32
  timeout 300 python bench.py --batch 32 --steps 400 --warmup 40 $Q --no-prefetch 2>/dev/null | line eager_b32_sync_sampler
done 2>&1 | tee -a $OUT/host_ab.txt
nproc; lscpu | grep -E "Model name|MHz" | head -3
