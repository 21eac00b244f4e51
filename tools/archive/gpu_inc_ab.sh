#!/bin/bash
# Inception: base build (microwakeword_amd/libmww_base.so) against this one, alternating bench runs + per-kernel times of both
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
ARGS="--model inception --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0 --steps 100 --warmup 20"
for rep in 1 2 3; do
  for lib in libmww_base.so libmww_hip.so; do
    MWW_HIP_LIB=$R/microwakeword_amd/$lib timeout 300 python bench.py $ARGS 2>/dev/null | line $lib
  done
done
for lib in libmww_base.so libmww_hip.so; do
  MWW_HIP_LIB=$R/microwakeword_amd/$lib timeout 300 python bench.py --model inception --no-cpu-baseline --no-validation --no-batch-sweep --steps 60 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernel_ms',{})
print('$lib', ' '.join('%s=%.1f' % (n, 1000*v) for n, v in k.items()))" | cut -c1-1500
done
