#!/bin/bash
# Same-session sweep of bench.py environment settings (and libraries) on the Inception step.
# usage: gpu_knobs.sh "<lib.so>:<VAR=val,...> ..." [bench args]     ("-" = no variables)
R=${GRAFT_REPO_ROOT:-$(pwd)}
CASES=$1; shift
cd $R
for rep in 1 2; do
  for cs in $CASES; do
    lib=${cs%%:*}; vars=${cs#*:}
    envs=""; [ "$vars" != "-" ] && envs=$(echo $vars | tr ',' ' ')
    env $envs MWW_HIP_LIB=$R/microwakeword_amd/$lib timeout 300 python bench.py --model inception --no-cpu-baseline --no-validation --profile-steps 0 --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cs', d['ms_per_step'])"
  done
done
