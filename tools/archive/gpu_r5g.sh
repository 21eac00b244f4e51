#!/bin/bash
# Round 5, session G: staggered start in the conv/BN graph kernels (Inception): variants libmww_<name>.so vs the shipped library.
# usage (repo root): bash tools/archive/gpu_r5g.sh <tag> [variant ...]
TAG=${1:-r5g}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "
from microwakeword_amd import build_native as bn
print('library sha256_16 =', bn.library_sha16(), 'source sha16 =', bn.library_source_sha16(), 'tree', bn.source_sha16())" | tee $OUT/summary.txt
line() {
  local lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), 'kernel_ms_sum', d['roofline']['kernel_ms_sum'])" | tee -a $OUT/summary.txt
}
for rep in 1 2 3; do
  line "inception shipped" --steps 100 --warmup 10 --model inception
  for v in "$@"; do
    MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so line "inception variant=$v" --steps 100 --warmup 10 --model inception
  done
done
line "inception shipped, 20 steps" --steps 20 --warmup 5 --model inception
line "mixednet on the graph kernels, shipped" --steps 100 --warmup 10 --force-generic
for v in "$@"; do MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so line "mixednet on the graph kernels, variant=$v" --steps 100 --warmup 10 --force-generic; done
tail -20 $OUT/summary.txt | cut -c1-250
