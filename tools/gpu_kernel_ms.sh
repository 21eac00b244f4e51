#!/bin/bash
# per-kernel milliseconds (bench.py's event profile) for a list of env settings: tools/gpu_kernel_ms.sh "A=1" "A=0" ...
set -u
export TMPDIR=/tmp
for envs in "$@"; do
  echo "== $envs"
  env $envs timeout 300 python bench.py --steps 100 --warmup 20 --no-validation --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], json.dumps(d["roofline"]["kernel_ms"]))'
done
