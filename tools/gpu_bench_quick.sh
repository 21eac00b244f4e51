#!/bin/bash
# quick: default bench (per-kernel ms) [+ extra bench args], optional B=8 floor
TAG=${1:-q}; shift
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
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation "$@" > $OUT/b0.json 2>$OUT/b0.err; kern $OUT/b0.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation "$@" > $OUT/b1.json 2>$OUT/b1.err; kern $OUT/b1.json
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --batch 8 "$@" > $OUT/b8.json 2>$OUT/b8.err; kern $OUT/b8.json
