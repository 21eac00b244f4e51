#!/bin/bash
# After a kernel change at the end of a round: the GPU suite + smoke on the library that ships with the tree, the kernel traces
# and counter passes of the default MixedNet and of Inception re-taken on it (profiles/round4_kernel_stats_and_pmc*.txt carry the
# library's sha256: bench.py reports roofline.traffic only from a summary of the library it loads), then the bench lines.
TAG=${1:-restamp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
BS="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BS > /dev/null 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $B > /dev/null 2> $OUT/pmc4.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inc -o t -- $BS --model inception > /dev/null 2> $OUT/trace_inc.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_inception -o p -- $B --model inception > /dev/null 2> $OUT/pmc1_inception.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3_inception -o p -- $B --model inception > /dev/null 2> $OUT/pmc3_inception.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4_inception -o p -- $B --model inception > /dev/null 2> $OUT/pmc4_inception.err
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/kernel_stats_and_pmc.txt 2>&1
python tools/pmc_summary.py $OUT/trace_inc $OUT/pmc1_inception $OUT/pmc3_inception $OUT/pmc4_inception > $OUT/kernel_stats_and_pmc_inception.txt 2>&1
cp $OUT/kernel_stats_and_pmc.txt profiles/round4_kernel_stats_and_pmc.txt
cp $OUT/kernel_stats_and_pmc_inception.txt profiles/round4_kernel_stats_and_pmc_inception.txt
cp $OUT/trace/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv; cp $OUT/trace_inc/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats_inception.csv
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err
timeout 600 python bench.py > $OUT/bench.json 2>> $OUT/bench.err
timeout 600 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2>> $OUT/bench.err
python - <<PY
import json
for f in ["bench_driver_form", "bench", "bench_inception"]:
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], "step_frac", r.get("step_frac"), "kernel", r.get("kernel"), "frac", r.get("frac"), "traffic", r.get("traffic"), str(r.get("traffic_source"))[:60])
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
