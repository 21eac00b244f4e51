#!/bin/bash
# Same-session comparison of grid policies of the conv/BN graph kernels on the Inception step:
#   base library (one grid of 3 workgroups per CU)  |  this library, per-launch grids  |  this library, grid_graph = 768
# usage: gpu_grid_ab.sh <base.so> <new.so> <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}
BASE=$1; NEW=$2; OUT=$R/gpurun_out/${3:-grid1}
mkdir -p $OUT; cd $R
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
A="--no-cpu-baseline --no-validation"
for rep in 1 2 3; do
  MWW_HIP_LIB=$R/microwakeword_amd/$BASE timeout 300 python bench.py --model inception $A --profile-steps 0 --steps 200 --warmup 20 2>/dev/null | line base
  MWW_HIP_LIB=$R/microwakeword_amd/$NEW timeout 300 python bench.py --model inception $A --profile-steps 0 --steps 200 --warmup 20 2>/dev/null | line new_auto
  MWW_BENCH_GRID_GRAPH=768 MWW_HIP_LIB=$R/microwakeword_amd/$NEW timeout 300 python bench.py --model inception $A --profile-steps 0 --steps 200 --warmup 20 2>/dev/null | line new_768
done
MWW_HIP_LIB=$R/microwakeword_amd/$BASE timeout 300 python bench.py --model inception $A --steps 100 --warmup 10 > $OUT/base_prof.json 2> $OUT/base_prof.err
MWW_HIP_LIB=$R/microwakeword_amd/$NEW timeout 300 python bench.py --model inception $A --steps 100 --warmup 10 > $OUT/auto_prof.json 2> $OUT/auto_prof.err
MWW_BENCH_GRID_GRAPH=768 MWW_HIP_LIB=$R/microwakeword_amd/$NEW timeout 300 python bench.py --model inception $A --steps 100 --warmup 10 > $OUT/g768_prof.json 2> $OUT/g768_prof.err
MWW_HIP_LIB=$R/microwakeword_amd/$BASE timeout 300 python bench.py --model notebook $A --profile-steps 0 --steps 100 --warmup 10 2>/dev/null | line notebook_base
MWW_HIP_LIB=$R/microwakeword_amd/$NEW timeout 300 python bench.py --model notebook $A --profile-steps 0 --steps 100 --warmup 10 2>/dev/null | line notebook_new
