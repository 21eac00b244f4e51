#!/bin/bash
# A/B/C... of several library builds in one session on one box: alternating bench.py runs
# usage: gpu_abn.sh "<libA.so> <libB.so> ..." [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=$1; shift
cd $R
for rep in 1 2 3; do
  for lib in $LIBS; do
    MWW_HIP_LIB=$R/microwakeword_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-validation --profile-steps 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'])"
  done
done
