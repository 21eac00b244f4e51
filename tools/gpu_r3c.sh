#!/bin/bash
# Round 3, third GPU session: the fused launches (kernels_fused.hip.h) - parity, then timing against one launch per layer
TAG=${1:-r3c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "fused_stages or baseline_size or train_steps" > $OUT/pytest_fused.log 2>&1; grep -E "passed|failed|error|Error|assert" $OUT/pytest_fused.log | tail -8
Q="--no-cpu-baseline --no-validation"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$1', 'ms/step', d['ms_per_step'], 'gpu', d['gpu_stream_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'loss', d['final_loss'], {n:round(v*1e3,1) for n,v in k.items()})"; }
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>$OUT/fused.err | tee $OUT/bench_fused_$rep.json | line fused
  MWW_BENCH_OPTIONS=fused_stages=0 timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line layers
done 2>&1 | tee $OUT/fused_ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench20.json | line driver_form
for b in 256 512 2048 4096; do timeout 300 python bench.py --batch $b --steps 100 --warmup 10 $Q 2>/dev/null | line "batch=$b"; done 2>&1 | tee $OUT/batch_sweep.txt
timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q 2>/dev/null | line notebook_fused
MWW_BENCH_OPTIONS=fused_stages=0 timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q 2>/dev/null | line notebook_layers
echo "== done"
