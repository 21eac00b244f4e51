#!/bin/bash
# Inception: GPU parity tests of the conv/BN graph kernels, then bench over workgroups per launch (eager per-kernel times) and graph replay
TAG=${1:-inc3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "inception or graph or residual or attention" 2>&1 | tail -8 | tee $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation $EXTRA > $OUT/$name.json 2> $OUT/$name.err; }
EXTRA=--no-graphs
run e_g1024 A=1
run e_g512 MWW_BENCH_GRID_GRAPH=512
run e_g256 MWW_BENCH_GRID_GRAPH=256
EXTRA=
run g_g1024 A=1
run g_g512 MWW_BENCH_GRID_GRAPH=512
run g_g768 MWW_BENCH_GRID_GRAPH=768
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
        if f.endswith("e_g512.json"): print("   ", {n:round(v*1e3,1) for n,v in k.items()})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
