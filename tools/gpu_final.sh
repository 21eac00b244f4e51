#!/bin/bash
# End-of-round GPU session: full GPU suite, smoke, the three bench configurations, kernel trace and PMC passes.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench default"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3200 $OUT/bench.json; tail -2 $OUT/bench.err
echo "== bench inception"
timeout 900 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2> $OUT/bench_inception.err; tail -c 400 $OUT/bench_inception.json | head -c 400; echo
echo "== bench bf16-operand"
timeout 900 python bench.py --pointwise-bf16 --no-cpu-baseline --no-validation > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; head -c 300 $OUT/bench_bf16.json; echo
echo "== bench bf16 storage (configs[4] full form), batch 1024 and 4096; fp32 and bf16-operand at 4096"
timeout 600 python bench.py --storage-bf16 --no-cpu-baseline --no-validation > $OUT/bench_bf16_storage.json 2> $OUT/bench_bf16_storage.err; head -c 200 $OUT/bench_bf16_storage.json; echo
for m in "--storage-bf16:bf16_storage" "--pointwise-bf16:bf16" ":f32"; do
  timeout 600 python bench.py ${m%%:*} --batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-validation > $OUT/bench_${m##*:}_b4096.json 2> $OUT/bench_${m##*:}_b4096.err; head -c 200 $OUT/bench_${m##*:}_b4096.json; echo
done
echo "== collective path forced on one GPU (RCCL world of one): local-BN with two buckets / one bucket, sync-BN"
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --no-validation 2> $OUT/bench_dp.err | tee $OUT/bench_dp.json | head -c 300; echo
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --no-validation --grad-buckets 1 2> $OUT/bench_dp1.err | tee $OUT/bench_dp1.json | head -c 300; echo
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --sync-bn --no-cpu-baseline --no-validation 2> $OUT/bench_dp_sync.err | tee $OUT/bench_dp_sync.json | head -c 300; echo
echo "== batch sweep"
for b in 256 512 2048 4096; do timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', d['value'], d['ms_per_step'], d['roofline']['step_frac'])" | tee -a $OUT/batch_sweep.txt; done
echo "== rocprofv3"
export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-graphs --no-cpu-baseline --no-validation --profile-steps 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $B > /dev/null 2> $OUT/pmc4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inc -o t -- $B --model inception > /dev/null 2> $OUT/trace_inc.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_st -o t -- $B --storage-bf16 > /dev/null 2> $OUT/trace_st.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/kernel_stats_and_pmc.txt 2>&1
python tools/pmc_summary.py $OUT/trace_inc > $OUT/kernel_stats_inception.txt 2>&1
python tools/pmc_summary.py $OUT/trace_st > $OUT/kernel_stats_bf16_storage.txt 2>&1
head -30 $OUT/kernel_stats_and_pmc.txt | cut -c1-160
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +12M -delete
echo "== done"
