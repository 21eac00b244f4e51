#!/bin/bash
TAG=${1:-e2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
kern() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms"]
print("ms/step=%.4f"%d["ms_per_step"], {n:round(v*1e3,1) for n,v in k.items()})
PY
}
for b in 8 64 256 512 2048; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --batch $b > $OUT/b$b.json 2>$OUT/b$b.err; echo "batch $b"; kern $OUT/b$b.json
done
