#!/bin/bash
# Round 5, session A: the block backward kernel forms (256 / 384 / 512 threads per 64-row tile, option "bwd_wide") on one box:
# parity tests of all forms, alternating bench lines with per-kernel HIP-event times, kernel trace + the two SQ counter
# passes per form.  usage (repo root): bash tools/archive/gpu_r5a.sh <tag>
TAG=${1:-r5a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "block_backward or mfma_layout_probe" > $OUT/pytest_forms.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_forms.log | tail -3 | tee -a $OUT/summary.txt
line() {  # <label> <bench args...>: one bench run -> "label ms_per_step {kernel: us}"
  local lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" | tee -a $OUT/summary.txt
}
for rep in 1 2; do
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_r5base.so MWW_BENCH_OPTIONS=bwd_wide=0 line "B1024 c+4-pitches(r5base) wide=0" --steps 200 --warmup 20
  for w in 0 384 512; do
    MWW_BENCH_OPTIONS=bwd_wide=$w line "B1024 wide=$w" --steps 200 --warmup 20
  done
done
for w in 0 384 512; do
  MWW_BENCH_OPTIONS=bwd_wide=$w line "driver-form wide=$w" --steps 20 --warmup 5
  MWW_BENCH_OPTIONS=bwd_wide=$w line "B4096 wide=$w" --steps 100 --warmup 10 --batch 4096
done
for w in 0 512; do
  MWW_BENCH_OPTIONS=bwd_wide=$w line "notebook wide=$w" --steps 200 --warmup 20 --model notebook
done
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
BS="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
cd /tmp
for w in 0 384 512; do
  export MWW_BENCH_OPTIONS=bwd_wide=$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o t -- $BS > /dev/null 2> $OUT/trace_$w.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_$w -o p -- $B > /dev/null 2> $OUT/pmc1_$w.err
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2_$w -o p -- $B > /dev/null 2> $OUT/pmc2_$w.err
  (cd $R; python tools/pmc_summary.py $OUT/trace_$w $OUT/pmc1_$w $OUT/pmc2_$w > $OUT/kernel_stats_and_pmc_wide$w.txt 2>&1)
done
export MWW_BENCH_OPTIONS=bwd_wide=512
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nb512 -o t -- $BS --model notebook > /dev/null 2> $OUT/trace_nb512.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_nb512 -o p -- $B --model notebook > /dev/null 2> $OUT/pmc1_nb512.err
(cd $R; python tools/pmc_summary.py $OUT/trace_nb512 $OUT/pmc1_nb512 > $OUT/kernel_stats_and_pmc_notebook_wide512.txt 2>&1)
unset MWW_BENCH_OPTIONS
cd $R
grep -h "bwd_block" $OUT/kernel_stats_and_pmc_wide*.txt | cut -c1-400 >> $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
tail -60 $OUT/summary.txt | cut -c1-330
