#!/bin/bash
# GPU session 2: tests, bench, grid sweep, PMC counters (separate passes, kernel-trace only).
TAG=${1:-r1b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== bench (hipGraph)"
timeout 600 python bench.py --steps 100 --warmup 20 > $OUT/bench_graph.json 2> $OUT/bench_graph.err ; tail -c 2600 $OUT/bench_graph.json ; tail -3 $OUT/bench_graph.err
echo "== grid sweep (eager, no cpu baseline)"
for cfg in "512 512" "768 512" "1024 256" "1024 512"; do
  set -- $cfg
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --grid-fwd $1 --grid-bwd $2 > $OUT/sweep_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/sweep_$1_$2.json"))
k=d["roofline"]["kernel_ms"]
print("grid_fwd=$1 grid_bwd=$2 ms/step=%.4f"%d["ms_per_step"], {n:round(v*1e3,1) for n,v in k.items() if n.startswith(("fwd","bwd","head","dense"))})
PY
done
echo "== rocprofv3 counters"
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
B="python $R/bench.py --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --profile-steps 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $B > /dev/null 2> $OUT/pmc4.err
for d in trace pmc1 pmc2 pmc3 pmc4; do echo "-- $d"; tail -2 $OUT/$d.err | cut -c1-200; find $OUT/$d -type f | head -5; done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +12M -delete
echo "== done"
