#!/bin/bash
# A/B of two library builds in one session on one box: alternating bench.py runs (per-kernel HIP-event times)
# usage: gpu_ab.sh <libA.so> <libB.so> [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
cd $R
for rep in 1 2 3; do
  for lib in $A $B; do
    MWW_HIP_LIB=$R/microwakeword_amd/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-validation "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lib', d['ms_per_step'], {n:round(v*1e3,1) for n,v in k.items()})"
  done
done
