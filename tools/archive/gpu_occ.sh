#!/bin/bash
# occupancy leverage of the backward kernels: 1 vs 2 workgroups per CU at the headline batch
TAG=${1:-occ}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for g in 256 384 512; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation --grid-bwd $g > $OUT/g$g.json 2> $OUT/g$g.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation --storage-bf16 > $OUT/st.json 2> $OUT/st.err
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["roofline"]["kernel_ms"]
        print(os.path.basename(f), "ms/step=%.4f"%d["ms_per_step"], {n:round(v*1e3,1) for n,v in k.items()})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
