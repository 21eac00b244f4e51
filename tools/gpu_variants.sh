#!/bin/bash
# alternating bench runs of library variants (tools/build_variant.sh): VARIANTS="a b c" [OPTS="fused_stages=0"] bash tools/gpu_variants.sh tag [reps] [bench args]
TAG=${1:-var}; REPS=${2:-3}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$1', 'ms/step', d['ms_per_step'], 'loss', d['final_loss'], {n:round(v*1e3,1) for n,v in k.items()})"; }
for rep in $(seq $REPS); do
  for v in $VARIANTS; do
    lib=${v%%:*}; opt=""; [[ "$v" == *:* ]] && opt=${v#*:}
    MWW_BENCH_OPTIONS=$opt MWW_HIP_LIB=$R/microwakeword_amd/libmww_$lib.so timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation "$@" 2>$OUT/err_$lib.txt | line "$v"
  done
done 2>&1 | tee $OUT/variants.txt
