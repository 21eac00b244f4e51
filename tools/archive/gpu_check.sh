#!/bin/bash
# quick end-of-work check: full GPU suite + smoke + the bench line in the driver's form and for the other two models
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'value', d['value'], 'step_frac', d['roofline']['step_frac'], 'dominant', d['roofline']['kernel'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])"; }
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench_driver_form.json | line driver_form
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line default
timeout 600 python bench.py --model inception --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | line inception
timeout 600 python bench.py --model notebook --no-cpu-baseline 2>/dev/null | line notebook
