#!/bin/bash
# The kernel trace + the four PMC passes of the headline configuration on the library that travelled with the tree (no rebuild):
# usage (repo root): bash tools/archive/gpu_pmc_only.sh <tag>  -> gpurun_out/<tag>/kernel_stats_and_pmc.txt (first line = the library's sha256)
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])"
export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --profile-steps 0"
BS="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BS > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $B > /dev/null 2> $OUT/pmc4.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/kernel_stats_and_pmc.txt 2>&1
head -4 $OUT/kernel_stats_and_pmc.txt | cut -c1-160
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2>/dev/null
cp $OUT/kernel_stats_and_pmc.txt $R/profiles/round4_kernel_stats_and_pmc.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +12M -delete
