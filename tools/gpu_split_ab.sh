#!/bin/bash
# assembly kernel: workgroups per window
set -u
out=gpurun_out/split_ab
mkdir -p $out
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
for rep in 1 2; do
  for v in 1 2 4; do
    echo "split=$v rep=$rep $(MWW_BENCH_ASM_SPLIT=$v timeout 300 python bench.py --steps 400 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/b_${v}_$rep.err | python -c "$P")"
  done
done
MWW_BENCH_ASM_SPLIT=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-validation --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("split1", d.get("kernels_us"))'
MWW_BENCH_ASM_SPLIT=2 timeout 300 python bench.py --steps 50 --warmup 10 --no-validation --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("split2", d.get("kernels_us"))'
MWW_BENCH_ASM_SPLIT=4 timeout 300 python bench.py --steps 50 --warmup 10 --no-validation --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("split4", d.get("kernels_us"))'
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "data or golden or sampler or fuzz or overlap or validation" 2>&1 | tail -2
