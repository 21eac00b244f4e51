#!/bin/bash
# counters of the Inception stem kernels, gathering (default) and dense (MWW_BENCH_FUSED_INPUT=0)
TAG=${1:-xgpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
B="python $R/bench.py --model inception --steps 8 --warmup 3 --no-graphs --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
for mode in 1 0; do
  export MWW_BENCH_FUSED_INPUT=$mode
  D=$OUT/f$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $D/pmc1 -o p -- $B > /dev/null 2> $D.pmc1.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $D/pmc2 -o p -- $B > /dev/null 2> $D.pmc2.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA -d $D/pmc3 -o p -- $B > /dev/null 2> $D.pmc3.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $D/pmc4 -o p -- $B > /dev/null 2> $D.pmc4.err
  (cd $R && python tools/pmc_summary.py $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 > $OUT/summary_f$mode.txt 2>&1)
  echo "== fused_input $mode"; grep "GShape<5, 1, 40, 40\|assemble" $OUT/summary_f$mode.txt | cut -c1-330
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
