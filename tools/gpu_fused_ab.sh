#!/bin/bash
# A/B of the "fused_input" option (first block gathers its rows from the feature stores) + the GPU suite
set -u
out=gpurun_out/fused_ab
mkdir -p $out
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
for rep in 1 2; do
  for v in 0 1; do
    echo "fused=$v rep=$rep $(MWW_BENCH_FUSED_INPUT=$v timeout 300 python bench.py --steps 400 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/b_${v}_$rep.err | python -c "$P")"
  done
done
for v in 0 1; do
  echo "notebook fused=$v $(MWW_BENCH_FUSED_INPUT=$v timeout 300 python bench.py --model notebook --steps 300 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/n_$v.err | python -c "$P")"
  echo "bf16 fused=$v $(MWW_BENCH_FUSED_INPUT=$v timeout 300 python bench.py --pointwise-bf16 --steps 300 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/h_$v.err | python -c "$P")"
  echo "B=4096 fused=$v $(MWW_BENCH_FUSED_INPUT=$v timeout 300 python bench.py --batch 4096 --steps 100 --warmup 20 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/l_$v.err | python -c "$P")"
  echo "validation fused=$v $(MWW_BENCH_FUSED_INPUT=$v timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 2>$out/v_$v.err | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get("validation"))')"
done
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
