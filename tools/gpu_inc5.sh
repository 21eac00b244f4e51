#!/bin/bash
# Inception: per-launch times against the batch size (latency floor vs throughput)
TAG=${1:-inc5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for b in 64 256 1024 4096; do
  MWW_BENCH_GRID_GRAPH=512 timeout 300 python bench.py --model inception --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-validation --no-graphs > $OUT/b$b.json 2> $OUT/b$b.err
done
python - $OUT <<'PY'
import json,sys,glob,os
rows={}
for b in (64,256,1024,4096):
    f=os.path.join(sys.argv[1],"b%d.json"%b)
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        rows[b]=d["roofline"]["kernel_ms"]; print(b, "ms/step=%.4f"%d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
names=sorted(rows[1024], key=lambda n:-rows[1024][n])
for n in names:
    print("%-22s"%n, "  ".join("%7.1f"%(rows[b].get(n,0)*1e3) for b in sorted(rows)))
PY
