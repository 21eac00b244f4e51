#!/bin/bash
# After a kernel change at the end of a round: the GPU suite + smoke on the library that ships with the tree, the kernel traces
# and counter passes of the default MixedNet and of Inception re-taken on it (profiles/round5_kernel_stats_and_pmc*.txt carry the
# library's sha256: bench.py reports roofline.traffic only from a summary of the library it loads), then the bench lines.
TAG=${1:-restamp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
# fuzzers on the shipped library (per-case reports; each one ends with a failure count)
{ timeout 600 python tools/gpu_stem_fuzz.py 24 2>&1 | tail -2; timeout 600 python tools/gpu_inc_fuzz.py 0 40 2>&1 | tail -3; timeout 600 python tools/gpu_table_fuzz.py 300 60 2>&1 | tail -2; timeout 300 python tools/gpu_static_diag.py 2>&1 | grep "^T "; } > $OUT/fuzz.txt 2>&1; tail -4 $OUT/fuzz.txt
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
cp $OUT/kernel_stats_and_pmc.txt profiles/round5_kernel_stats_and_pmc.txt
cp $OUT/kernel_stats_and_pmc_inception.txt profiles/round5_kernel_stats_and_pmc_inception.txt
cp $OUT/trace/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv; cp $OUT/trace_inc/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats_inception.csv
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form_2.json 2>> $OUT/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form_3.json 2>> $OUT/bench.err
timeout 600 python bench.py > $OUT/bench.json 2>> $OUT/bench.err
timeout 600 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2>> $OUT/bench.err
timeout 600 python bench.py --model inception --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_inception_driver_form.json 2>> $OUT/bench.err
timeout 600 python bench.py --model notebook --no-cpu-baseline > $OUT/bench_notebook.json 2>> $OUT/bench.err
timeout 600 python bench.py --pointwise-bf16 --no-cpu-baseline > $OUT/bench_pointwise_bf16.json 2>> $OUT/bench.err
timeout 600 python bench.py --storage-bf16 --no-cpu-baseline > $OUT/bench_bf16_storage.json 2>> $OUT/bench.err
timeout 600 python bench.py --storage-bf16 --batch 4096 --steps 100 --no-cpu-baseline > $OUT/bench_bf16_storage_b4096.json 2>> $OUT/bench.err
timeout 600 python bench.py --pointwise-bf16 --batch 4096 --steps 100 --no-cpu-baseline > $OUT/bench_bf16_b4096.json 2>> $OUT/bench.err
timeout 600 python bench.py --batch 4096 --steps 100 --no-cpu-baseline > $OUT/bench_f32_b4096.json 2>> $OUT/bench.err
timeout 600 python bench.py --force-generic --steps 100 --no-cpu-baseline > $OUT/bench_mixednet_on_graph_kernels.json 2>> $OUT/bench.err
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_forced_dp_one_bucket.json 2>> $OUT/bench.err
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --grad-buckets 2 > $OUT/bench_forced_dp_two_buckets.json 2>> $OUT/bench.err
MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --sync-bn > $OUT/bench_forced_dp_sync_bn.json 2>> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nb -o t -- $BS --model notebook > /dev/null 2> $OUT/trace_nb.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_nb -o p -- $B --model notebook > /dev/null 2> $OUT/pmc1_nb.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2_nb -o p -- $B --model notebook > /dev/null 2> $OUT/pmc2_nb.err
cd $R
python tools/pmc_summary.py $OUT/trace_nb $OUT/pmc1_nb $OUT/pmc2_nb > $OUT/kernel_stats_and_pmc_notebook.txt 2>&1
cp $OUT/trace_nb/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats_notebook.csv
python - <<PY
import json
for f in ["bench_driver_form", "bench_driver_form_2", "bench_driver_form_3", "bench", "bench_inception", "bench_inception_driver_form", "bench_notebook", "bench_pointwise_bf16", "bench_bf16_storage", "bench_bf16_storage_b4096", "bench_bf16_b4096", "bench_f32_b4096", "bench_mixednet_on_graph_kernels", "bench_forced_dp_one_bucket", "bench_forced_dp_two_buckets", "bench_forced_dp_sync_bn"]:
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], "sweep", d.get("batch_sweep"), "host", d.get("host_enqueue_ms_per_step"), "step_frac", r.get("step_frac"), "kernel", r.get("kernel"), "frac", r.get("frac"), "traffic", r.get("traffic"), str(r.get("traffic_source"))[:60])
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
