#!/bin/bash
# Round 5, session H: which LDS array of the forward block kernels carries the remaining bank conflicts - the SQ LDS counters of
# variant builds with the old pitch (c + 4) on the input tile (fpa0), on the u tile (fpu0), on both (fp00) and the shipped pitches.
# usage (repo root): bash tools/archive/gpu_r5h.sh <tag> [variant ...]
TAG=${1:-r5h}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
: > $OUT/summary.txt
cd /tmp
for v in shipped "$@"; do
  if [ $v = shipped ]; then unset MWW_HIP_LIB; else export MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_MEM_VIOLATIONS -d $OUT/pmc_$v -o p -- $B > /dev/null 2> $OUT/pmc_$v.err
  (cd $R; python tools/pmc_summary.py $OUT/pmc_$v | grep "fwd_\|bwd_first" | sed "s/^/$v: /" | cut -c1-330 >> $OUT/summary.txt)
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
cat $OUT/summary.txt
