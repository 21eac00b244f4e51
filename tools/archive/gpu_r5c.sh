#!/bin/bash
# Round 5, session C: the bf16 modes (BASELINE configs[4]) on the wide block backward kernels next to the 256-thread ones
# ("bwd_wide" 1 / 0), and kernel-tuning variants of the library given as extra arguments (libmww_<name>.so, slim builds).
# usage (repo root): bash tools/archive/gpu_r5c.sh <tag> [variant ...]
TAG=${1:-r5c}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "
from microwakeword_amd import build_native as bn
print('library sha256_16 =', bn.library_sha16(), 'source sha16 =', bn.library_source_sha16(), 'tree', bn.source_sha16())" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "bf16" > $OUT/pytest_bf16.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_bf16.log | tail -3 | tee -a $OUT/summary.txt
line() {  # <label> <bench args...>
  local lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" | tee -a $OUT/summary.txt
}
for rep in 1 2; do
  for w in 0 1; do
    MWW_BENCH_OPTIONS=bwd_wide=$w line "pointwise-bf16 B1024 wide=$w" --steps 200 --warmup 20 --pointwise-bf16
    MWW_BENCH_OPTIONS=bwd_wide=$w line "storage-bf16 B1024 wide=$w" --steps 200 --warmup 20 --storage-bf16
  done
done
for w in 0 1; do
  MWW_BENCH_OPTIONS=bwd_wide=$w line "pointwise-bf16 B4096 wide=$w" --steps 100 --warmup 10 --batch 4096 --pointwise-bf16
  MWW_BENCH_OPTIONS=bwd_wide=$w line "storage-bf16 B4096 wide=$w" --steps 100 --warmup 10 --batch 4096 --storage-bf16
done
for rep in 1 2 3; do
  line "fp32 B1024 shipped" --steps 200 --warmup 20
  for v in "$@"; do
    MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so line "fp32 B1024 variant=$v" --steps 200 --warmup 20
  done
done
tail -40 $OUT/summary.txt | cut -c1-330
