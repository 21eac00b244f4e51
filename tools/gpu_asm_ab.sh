#!/bin/bash
# A/B of the "assemble_overlap" option (batch assembly of step k+1 next to the gradient reduction / Adam of step k)
set -u
out=gpurun_out/asm_ab
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "overlap or frozen" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
for rep in 1 2; do
  for v in 0 1; do
    MWW_BENCH_ASM_OVERLAP=$v timeout 300 python bench.py --steps 400 --warmup 50 --no-validation > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    echo "overlap=$v rep=$rep $(python -c "import json,sys; d=json.loads(open('$out/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
for v in 0 1; do
  MWW_BENCH_FORCE_DP=1 MWW_BENCH_ASM_OVERLAP=$v timeout 300 python bench.py --steps 400 --warmup 50 --no-validation > $out/dp_${v}.json 2> $out/dp_${v}.err
  echo "dp overlap=$v $(python -c "import json,sys; d=json.loads(open('$out/dp_${v}.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
  MWW_BENCH_ASM_OVERLAP=$v timeout 300 python bench.py --model inception --steps 200 --warmup 30 --no-validation > $out/inc_${v}.json 2> $out/inc_${v}.err
  echo "inception overlap=$v $(python -c "import json,sys; d=json.loads(open('$out/inc_${v}.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
