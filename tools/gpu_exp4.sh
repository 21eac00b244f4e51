#!/bin/bash
TAG=${1:-e4}
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
echo "== default"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/b0.json 2>$OUT/b0.err; kern $OUT/b0.json; tail -2 $OUT/b0.err
echo "== forced DP, 2 buckets"
MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/dp2.json 2>$OUT/dp2.err; kern $OUT/dp2.json; tail -2 $OUT/dp2.err
echo "== forced DP, 1 bucket"
MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation --grad-buckets 1 > $OUT/dp1.json 2>$OUT/dp1.err; kern $OUT/dp1.json; tail -2 $OUT/dp1.err
echo "== grid-head sweep"
for g in 256 512; do timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-validation --grid-head $g > $OUT/gh$g.json 2>/dev/null; echo "grid-head $g"; kern $OUT/gh$g.json; done
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest.log
