#!/bin/bash
# Round 5, session D: grad_final with one batch of partial-row loads per slice (shipped) vs four batches of 16 (variant final16),
# and grid sweeps of the head / forward kernels.  usage (repo root): bash tools/archive/gpu_r5d.sh <tag>
TAG=${1:-r5d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "
from microwakeword_amd import build_native as bn
print('library sha256_16 =', bn.library_sha16(), 'source sha16 =', bn.library_source_sha16(), 'tree', bn.source_sha16())" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "determinism or train_steps_small or tail_roles or dense" > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/summary.txt
line() {
  local lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" | tee -a $OUT/summary.txt
}
for rep in 1 2 3; do
  line "shipped" --steps 200 --warmup 20
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_final16.so line "final16" --steps 200 --warmup 20
done
for g in 256 768 1024; do line "grid-head $g" --steps 200 --warmup 20 --grid-head $g; done
for g in 512 768 1024; do line "grid-fwd $g" --steps 200 --warmup 20 --grid-fwd $g; done
line "driver-form shipped" --steps 20 --warmup 5
tail -20 $OUT/summary.txt | cut -c1-330
