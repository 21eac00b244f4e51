#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/abl2; mkdir -p $OUT; cd $R
for m in 0 8; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --ablate $m > $OUT/abl_$m.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/abl_$m.json"))
k=d["roofline"]["kernel_ms"]
print("ablate=$m ms/step=%.4f"%d["ms_per_step"], {n:round(v*1e3,1) for n,v in k.items() if n.startswith(("fwd_block","bwd_block"))})
PY
done
