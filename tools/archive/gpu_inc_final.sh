#!/bin/bash
# Inception: GPU parity tests of the conv/BN graph kernels, bench line, kernel trace
TAG=${1:-incfinal}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "inception or graph or residual or attention or train_loop" 2>&1 | tail -8 | tee $OUT/pytest.log
timeout 600 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_inception.json 2> $OUT/bench_inception.err; tail -c 600 $OUT/bench_inception.json
timeout 600 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation --no-graphs > $OUT/bench_inception_eager.json 2> $OUT/bench_inception_eager.err
MWW_BENCH_OPTIONS=bn_inline=0 timeout 600 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation > $OUT/bench_inception_finalize_launches.json 2> $OUT/bench_inception_fl.err
for b in 256 4096; do timeout 300 python bench.py --model inception --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', d['value'], d['ms_per_step'], d['roofline']['step_frac'])" | tee -a $OUT/batch_sweep_inception.txt; done
python - $OUT <<'PY'
import json,sys,os
for f in ("bench_inception.json","bench_inception_eager.json","bench_inception_finalize_launches.json"):
    d=json.loads(open(os.path.join(sys.argv[1],f)).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["step_frac"])
PY
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inc -o t -- python $R/bench.py --model inception --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --no-validation --profile-steps 0 > /dev/null 2> $OUT/trace_inc.err
cd $R
python tools/pmc_summary.py $OUT/trace_inc > $OUT/kernel_stats_inception.txt 2>&1
head -30 $OUT/kernel_stats_inception.txt | cut -c1-130
find $OUT -name "*kernel_trace.csv" -size +8M -delete
