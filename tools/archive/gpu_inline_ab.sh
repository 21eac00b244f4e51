#!/bin/bash
# A/B of the "bn_inline" option (BN sums through fp64 accumulator rows, folded by the first consumer) + the GPU suite
set -u
out=gpurun_out/inline_ab
mkdir -p $out
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
for rep in 1 2; do
  for v in 0 1; do
    echo "inline=$v rep=$rep $(MWW_BENCH_OPTIONS=bn_inline=$v timeout 300 python bench.py --steps 400 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/b_${v}_$rep.err | python -c "$P")"
  done
done
for v in 0 1; do
  echo "notebook inline=$v $(MWW_BENCH_OPTIONS=bn_inline=$v timeout 300 python bench.py --model notebook --steps 300 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/n_$v.err | python -c "$P")"
  echo "bf16 inline=$v $(MWW_BENCH_OPTIONS=bn_inline=$v timeout 300 python bench.py --pointwise-bf16 --steps 300 --warmup 50 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/h_$v.err | python -c "$P")"
  echo "B=4096 inline=$v $(MWW_BENCH_OPTIONS=bn_inline=$v timeout 300 python bench.py --batch 4096 --steps 100 --warmup 20 --no-validation --no-cpu-baseline --profile-steps 0 2>$out/l_$v.err | python -c "$P")"
done
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q > $out/pytest.log 2>&1
tail -5 $out/pytest.log
