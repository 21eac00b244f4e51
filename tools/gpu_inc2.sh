#!/bin/bash
# Inception bench variants: hand-over on/off, workgroups per launch
TAG=${1:-inc2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation --no-graphs > $OUT/$name.json 2> $OUT/$name.err; }
run inl1_g1024 MWW_BENCH_BN_INLINE=1
run inl0_g1024 MWW_BENCH_BN_INLINE=0
run inl1_g512 MWW_BENCH_BN_INLINE=1 MWW_BENCH_GRID_GRAPH=512
run inl0_g512 MWW_BENCH_BN_INLINE=0 MWW_BENCH_GRID_GRAPH=512
run inl1_g256 MWW_BENCH_BN_INLINE=1 MWW_BENCH_GRID_GRAPH=256
run inl1_g768 MWW_BENCH_BN_INLINE=1 MWW_BENCH_GRID_GRAPH=768
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
        print(os.path.basename(f), "ms/step=%.4f kernel_sum=%.4f"%(d["ms_per_step"], d["roofline"]["kernel_ms_sum"]), {n:(c,round(v*1e3,1)) for n,(c,v) in sorted(agg.items(), key=lambda x:-x[1][1])})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
