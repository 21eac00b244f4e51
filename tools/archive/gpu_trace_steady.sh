#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command over 10 + 50 steps (steady device, first-call cost 1/60 of the averages)
TAG=${1:-trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0 > $OUT/bench_under_tracer.json 2> $OUT/trace.err
cd $R
python tools/pmc_summary.py $OUT/trace > $OUT/kernel_stats_steady.txt 2>&1
head -16 $OUT/kernel_stats_steady.txt | cut -c1-160
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('events', d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" | tee $OUT/events_same_session.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
