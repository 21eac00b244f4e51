#!/bin/bash
TAG=${1:-e6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
kern() { python - "$1" <<'PY'
import json,sys,re,collections
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms"]
agg=collections.defaultdict(lambda:[0,0.0])
for n,v in k.items():
    a=re.sub(r"\d+$","",n); agg[a][0]+=1; agg[a][1]+=v
print("ms/step=%.4f"%d["ms_per_step"], "kernel_sum=%.4f"%d["roofline"]["kernel_ms_sum"], {a:(c,round(v*1e3,1)) for a,(c,v) in sorted(agg.items(), key=lambda t:-t[1][1])})
PY
}
echo "== inception eager"
timeout 300 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation > $OUT/inc.json 2>$OUT/inc.err; kern $OUT/inc.json
echo "== inception graphs"
timeout 300 python bench.py --model inception --steps 100 --warmup 10 --no-cpu-baseline --no-validation --graphs > $OUT/incg.json 2>$OUT/incg.err; kern $OUT/incg.json
echo "== mixednet graphs"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-validation --graphs > $OUT/mg.json 2>$OUT/mg.err; kern $OUT/mg.json
echo "== inception B=1024 test"
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "inception_train_step_batch1024" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
