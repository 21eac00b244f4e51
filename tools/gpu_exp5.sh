#!/bin/bash
TAG=${1:-e5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
kern() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms"]
print("ms/step=%.4f"%d["ms_per_step"], "host=%s" % d.get("host_enqueue_ms_per_step"), {n:round(v*1e3,1) for n,v in k.items()})
PY
}
echo "== default"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/b0.json 2>$OUT/b0.err; kern $OUT/b0.json
echo "== batch 8"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --batch 8 > $OUT/b8.json 2>$OUT/b8.err; kern $OUT/b8.json
echo "== forced DP, 2 buckets"
MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/dp2.json 2>$OUT/dp2.err; kern $OUT/dp2.json
echo "== forced DP, 1 bucket"
MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation --grad-buckets 1 > $OUT/dp1.json 2>$OUT/dp1.err; kern $OUT/dp1.json
echo "== gpu tests (failed one first)"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bn_inline or rccl or saturated or variable" 2>&1 | tail -4 | tee $OUT/pytest.log
