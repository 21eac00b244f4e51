#!/bin/bash
# One GPU session: build check, GPU parity tests, bench (graph + eager), rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
echo "== build" ; timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench (hipGraph)"
timeout 900 python bench.py --steps 100 --warmup 20 > $OUT/bench_graph.json 2> $OUT/bench_graph.err ; tail -c 3000 $OUT/bench_graph.json ; tail -5 $OUT/bench_graph.err
echo "== bench (eager)"
timeout 900 python bench.py --steps 100 --warmup 20 --no-graphs --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err ; tail -c 1500 $OUT/bench_eager.json ; tail -5 $OUT/bench_eager.err
echo "== rocprofv3 kernel trace"
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/bench.py --steps 30 --warmup 5 --no-graphs --no-cpu-baseline --profile-steps 0 > $OUT/prof_bench.json 2> $OUT/prof.err )
tail -3 $OUT/prof.err
find $OUT/prof -name "*stats*" | head ; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && head -30 "$f"
# keep the merged-back payload small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"
