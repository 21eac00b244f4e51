#!/bin/bash
# Inception: kernel trace + SQ counters of the conv/BN graph kernels
TAG=${1:-incpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp MWW_BENCH_OPTIONS=grid_graph=${GRID:-0}
B="python $R/bench.py --model inception --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --no-validation --profile-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES_EQ_64 -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/kernel_stats_and_pmc.txt 2>&1
head -70 $OUT/kernel_stats_and_pmc.txt | cut -c1-400
tail -3 $OUT/pmc3.err
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +12M -delete
