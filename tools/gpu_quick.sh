#!/bin/bash
# quick GPU check: selected tests (-k $2) + one bench run ($3 = extra bench args)
TAG=${1:-q}; SEL=${2:-inception}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -k "$SEL" --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest.log
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err ; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"])
k=d["roofline"]["kernel_ms"]
import collections,re
agg=collections.defaultdict(float)
for n,v in k.items(): agg[re.sub(r"\d+$","",n)]+=v
print({a:round(b,4) for a,b in sorted(agg.items(), key=lambda t:-t[1])})
PY
tail -2 $OUT/bench.err
