#!/bin/bash
# same-session A/B of runtime environment settings on the default bench line: bash tools/archive/gpu_env_ab.sh tag "VAR=a" "VAR=b" ...  ("-" = unchanged environment)
TAG=${1:-env}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
for rep in 1 2 3; do
  for e in "$@"; do
    ( [ "$e" != "-" ] && export $e; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$e', 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], {n:round(v*1e3,1) for n,v in k.items()})" )
  done
done 2>&1 | tee $OUT/env_ab.txt
