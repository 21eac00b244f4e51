#!/bin/bash
# Round 3, second GPU session: full GPU suite + the bench line as the driver runs it (5 + 20 steps) and at 20 + 200 steps
TAG=${1:-r3b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$1', 'ms/step', d['ms_per_step'], 'gpu', d['gpu_stream_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'val', (d.get('validation') or {}).get('windows_per_s'), d.get('validation'), {n:round(v*1e3,1) for n,v in k.items()})"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/b20.err | tee $OUT/bench20_$rep.json | line driver_form
  MWW_BENCH_SETTLE_S=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-validation 2>/dev/null | line driver_form_no_settle
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tee $OUT/bench200_$rep.json | line long_form
done 2>&1 | tee $OUT/bench_forms.txt
echo "== done"
