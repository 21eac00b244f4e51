#!/bin/bash
# experiment: per-kernel times, phase clocks and phase ablation with the MWW_PROFILE build; grid sweeps
TAG=${1:-e1}
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
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-validation > $OUT/b0.json 2>$OUT/b0.err; kern $OUT/b0.json
for g in 256 512 768; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --grid-fwd $g > $OUT/gf$g.json 2>/dev/null; echo "grid-fwd $g"; kern $OUT/gf$g.json
done
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --grid-bwd 256 > $OUT/gb256.json 2>/dev/null; echo "grid-bwd 256"; kern $OUT/gb256.json
echo "== profile lib"
export MWW_HIP_LIB=$R/microwakeword_amd/libmww_hip_prof.so
for m in 0 1 2 4 8 7 15; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-validation --ablate $m > $OUT/abl_$m.json 2>/dev/null; echo "ablate $m"; kern $OUT/abl_$m.json
done
timeout 300 python tools/phase_clocks.py 2>&1 | tail -20 | tee $OUT/phase.txt
