#!/bin/bash
# the driver-form and the 20 + 200 bench lines (with cpu_baseline, validation, batch_sweep) plus the Inception / notebook
# lines, re-taken once the PMC summaries under profiles/ carry this library's stamp (roofline.traffic is then filled in)
TAG=${1:-lines}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err
timeout 900 python bench.py > $OUT/bench.json 2>> $OUT/bench.err
timeout 900 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2>> $OUT/bench.err
timeout 900 python bench.py --model notebook --no-cpu-baseline --no-validation > $OUT/bench_notebook.json 2>> $OUT/bench.err
python - <<PY
import json
for f in ["bench_driver_form", "bench", "bench_inception", "bench_notebook"]:
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], "step_frac", r.get("step_frac"), "kernel", r.get("kernel"), "frac", r.get("frac"), "traffic", r.get("traffic"), str(r.get("traffic_source"))[:70])
PY
