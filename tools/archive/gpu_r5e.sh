#!/bin/bash
# Round 5, session E: staggered start of the workgroups that share a CU (MWW_STAGGER_*: slim variant builds) against the
# shipped library, and the driver's exact command three times.  usage (repo root): bash tools/archive/gpu_r5e.sh <tag> [variant ...]
TAG=${1:-r5e}; shift
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
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" | tee -a $OUT/summary.txt
}
for rep in 1 2 3; do
  line "shipped" --steps 200 --warmup 20
  for v in "$@"; do
    MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so line "variant=$v" --steps 200 --warmup 20
  done
done
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('driver command (python bench.py --steps 20 --warmup 5):', d['ms_per_step'], 'step_frac', d['roofline']['step_frac'], 'batch4096', d.get('batch_sweep'), 'kernel_ms_sum', d['roofline']['kernel_ms_sum'], 'lib', d.get('library_sha16'), d.get('source_sha16'), d.get('tree_source_sha16'))" | tee -a $OUT/summary.txt
done
tail -30 $OUT/summary.txt | cut -c1-330
