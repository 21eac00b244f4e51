#!/bin/bash
# Inception A/B: library builds x workgroups per launch x role split (eager per-kernel sums + graph replay)
TAG=${1:-incab2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for lib in libmww_hip.so libmww_hip_v1.so; do
  for cfg in "512 1" "1024 1" "768 1" "512 0"; do
    set -- $cfg
    MWW_HIP_LIB=$R/microwakeword_amd/$lib MWW_BENCH_GRID_GRAPH=$1 MWW_BENCH_ROLE_SPLIT=$2 timeout 300 python bench.py --model inception --steps 60 --warmup 10 --no-cpu-baseline --no-validation --no-graphs > $OUT/${lib%.so}_g$1_s$2.json 2> $OUT/${lib%.so}_g$1_s$2.err
  done
done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["roofline"]["kernel_ms"]
        agg={}
        for n,v in k.items():
            b=n.rstrip("0123456789")
            agg.setdefault(b,[0,0.0]); agg[b][0]+=1; agg[b][1]+=v
        print(os.path.basename(f), "ms/step=%.4f"%d["ms_per_step"], {n:(c,round(v*1e3,1)) for n,(c,v) in sorted(agg.items(), key=lambda x:-x[1][1])[:6]})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
