#!/bin/bash
# GPU session: inception tests first, then the full GPU suite, then both benches + kernel trace of inception.
TAG=${1:-r1g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== pytest -m gpu -k inception"
timeout 900 python -m pytest tests -m gpu -q -k inception --timeout 600 -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_inception.log
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== bench inception"
timeout 600 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2> $OUT/bench_inception.err ; tail -c 4500 $OUT/bench_inception.json ; tail -3 $OUT/bench_inception.err
echo "== bench mixednet"
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_graph.json 2> $OUT/bench_graph.err ; tail -c 600 $OUT/bench_graph.json ; tail -3 $OUT/bench_graph.err
echo "== rocprofv3 inception"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inc -o t -- python $R/bench.py --model inception --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --profile-steps 0 > /dev/null 2> $OUT/trace_inc.err
cd $R
f=$(find $OUT/trace_inc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +8M -delete
echo "== done"
