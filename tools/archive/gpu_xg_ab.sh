#!/bin/bash
# Inception stem gathering from the feature stores (XG instantiations) against the materialised batch:
# the GPU parity check, then alternating bench runs (this library with the option on / off, and the base build), then a kernel trace
TAG=${1:-xg}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -p no:cacheprovider -k "stem_gathers or fused_input or inception_static" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline'].get('kernel_ms_sum'))"; }
ARGS="--model inception --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0 --steps 100 --warmup 20"
for rep in 1 2 3; do
  timeout 300 python bench.py $ARGS 2>/dev/null | line gather
  MWW_BENCH_FUSED_INPUT=0 timeout 300 python bench.py $ARGS 2>/dev/null | line materialised
  [ -f $R/microwakeword_amd/libmww_base.so ] && MWW_HIP_LIB=$R/microwakeword_amd/libmww_base.so timeout 300 python bench.py $ARGS 2>/dev/null | line base
done
timeout 300 python bench.py --model inception --no-cpu-baseline --no-validation --no-batch-sweep --steps 100 --warmup 20 2>/dev/null | tee $OUT/bench_inception.json | line gather_profiled
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --model inception --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0 --steps 60 --warmup 10 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 $f | cut -c1-160
