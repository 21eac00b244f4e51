#!/bin/bash
# Round 5, session B: the wide first-block backward kernel (bwd_firstw_kernel) next to the 256-thread one, same box, same session:
# parity tests, alternating bench lines ("bwd_wide" 0 / 1), kernel trace + SQ counter passes of the default form.
# usage (repo root): bash tools/archive/gpu_r5b.sh <tag>
TAG=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "
from microwakeword_amd import build_native as bn
print('library sha256_16 =', bn.library_sha16(), 'source sha16 =', bn.library_source_sha16(), 'tree', bn.source_sha16())" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "block_backward or mfma_layout_probe or notebook or tail_rows or crossed or determinism or train_steps_small" > $OUT/pytest_forms.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_forms.log | tail -3 | tee -a $OUT/summary.txt
line() {  # <label> <bench args...>: one bench run -> "label ms_per_step {kernel: us}"
  local lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$lab', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" | tee -a $OUT/summary.txt
}
for rep in 1 2 3; do
  for w in 0 1; do
    MWW_BENCH_OPTIONS=bwd_wide=$w line "B1024 wide=$w" --steps 200 --warmup 20
  done
done
for w in 0 1; do
  MWW_BENCH_OPTIONS=bwd_wide=$w line "driver-form wide=$w" --steps 20 --warmup 5
  MWW_BENCH_OPTIONS=bwd_wide=$w line "B4096 wide=$w" --steps 100 --warmup 10 --batch 4096
  MWW_BENCH_OPTIONS=bwd_wide=$w line "notebook wide=$w" --steps 200 --warmup 20 --model notebook
done
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
BS="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BS > /dev/null 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nb -o t -- $BS --model notebook > /dev/null 2> $OUT/trace_nb.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_nb -o p -- $B --model notebook > /dev/null 2> $OUT/pmc1_nb.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 > $OUT/kernel_stats_and_pmc.txt 2>&1
python tools/pmc_summary.py $OUT/trace_nb $OUT/pmc1_nb > $OUT/kernel_stats_and_pmc_notebook.txt 2>&1
grep -h "bwd_first" $OUT/kernel_stats_and_pmc.txt $OUT/kernel_stats_and_pmc_notebook.txt | cut -c1-400 >> $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
tail -40 $OUT/summary.txt | cut -c1-330
