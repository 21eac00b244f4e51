#!/bin/bash
TAG=${1:-abl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for m in 0 1 2 4 3 5 6 7; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --ablate $m > $OUT/abl_$m.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/abl_$m.json"))
k=d["roofline"]["kernel_ms"]
print("ablate=$m ms/step=%.4f"%d["ms_per_step"], {n:round(v*1e3,1) for n,v in k.items() if n in ("fwd_block2","fwd_block4","bwd_block2","bwd_block3","bwd_block4")})
PY
done
