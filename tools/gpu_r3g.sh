#!/bin/bash
# Round 3: full GPU suite on the noslp / fused-backward build, notebook topology grid sweep, default bench in both forms
TAG=${1:-r3g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
Q="--no-cpu-baseline --no-validation"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$1', 'ms/step', d['ms_per_step'], 'gpu', d['gpu_stream_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'loss', d['final_loss'], {n:round(v*1e3,1) for n,v in k.items()})"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line default
  MWW_BENCH_OPTIONS=fused_stages=0 timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line default_layers
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line default_driver_form
done 2>&1 | tee $OUT/default.txt
for g in "1024 512" "512 512" "512 256" "768 256" "1024 256"; do
  set -- $g
  MWW_BENCH_OPTIONS=fused_stages=0 timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q --grid-fwd $1 --grid-bwd $2 2>/dev/null | line "notebook_layers_grids_$1_$2"
done 2>&1 | tee $OUT/notebook_grids.txt
timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q 2>/dev/null | tee $OUT/bench_notebook.json | line notebook_default
timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q 2>/dev/null | line notebook_default
echo "== done"
