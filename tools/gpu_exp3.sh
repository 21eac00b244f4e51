#!/bin/bash
# compare library builds: default vs -fno-slp-vectorize; GPU parity tests on the default
TAG=${1:-e3}
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
echo "== default lib"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/b0.json 2>$OUT/b0.err; kern $OUT/b0.json
echo "== noslp lib"
MWW_HIP_LIB=$R/microwakeword_amd/libmww_hip_noslp.so timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/b1.json 2>$OUT/b1.err; kern $OUT/b1.json
echo "== default lib again"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/b2.json 2>$OUT/b2.err; kern $OUT/b2.json
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest.log
