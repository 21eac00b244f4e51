#!/bin/bash
# GPU session: tests + bench + light PMC on the main kernels.
TAG=${1:-r1c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== bench (hipGraph)"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench_graph.json 2> $OUT/bench_graph.err ; tail -c 2700 $OUT/bench_graph.json ; tail -3 $OUT/bench_graph.err
echo "== rocprofv3"
export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --profile-steps 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 2>&1 | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +8M -delete
echo "== done"
