#!/bin/bash
# bf16 storage mode: parity tests and the configs[4] records
TAG=${1:-st}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "bf16" 2>&1 | tail -8 | tee $OUT/pytest.log
for b in 1024 4096; do
  timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-validation --storage-bf16 > $OUT/st_b$b.json 2> $OUT/st_b$b.err
  timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-validation --pointwise-bf16 > $OUT/pw_b$b.json 2> $OUT/pw_b$b.err
  timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-validation > $OUT/f32_b$b.json 2> $OUT/f32_b$b.err
done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*_b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["roofline"]["kernel_ms"]
        print(os.path.basename(f), "ms/step=%.4f"%d["ms_per_step"], d["value"], d["roofline"]["step_frac"], {n:round(v*1e3,1) for n,v in k.items()})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
