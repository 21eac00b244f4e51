#!/bin/bash
# End-of-round random sweeps beyond the committed case counts (same checks as the -m gpu tests)
TAG=${1:-fuzz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python tools/gpu_shape_fuzz.py ${F_SHAPE:-3000} ${N_SHAPE:-250} 2>&1 | tail -6 ) | tee $OUT/shape_fuzz.txt
( timeout 900 python tools/gpu_inc_fuzz.py ${F_INC:-2000} ${N_INC:-250} 2>&1 | tail -6 ) | tee $OUT/inc_fuzz.txt
( timeout 900 python tools/gpu_topo_fuzz.py ${F_TOPO:-2000} ${N_TOPO:-250} 2>&1 | tail -6 ) | tee $OUT/topo_fuzz.txt
( timeout 900 python tools/gpu_big_fuzz.py 200 60 2>&1 | tail -4 ) | tee $OUT/big_fuzz.txt
