#!/bin/bash
# End-of-round GPU session: full GPU suite, smoke, the bench configurations, kernel trace and PMC passes.
# usage (repo root): bash tools/archive/gpu_final.sh <tag>   -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/round<N>_*
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
# the library that travelled with the tree is the one measured (and the one whose sha256 the PMC summary is stamped with: a
# rebuild on the box - copied sources may look newer than the copied library - would produce another binary than the one the
# driver's own bench run loads)
python -c "import hashlib; print('library sha256_16 =', hashlib.sha256(open('microwakeword_amd/libmww_hip.so','rb').read()).hexdigest()[:16])"
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
Q="--no-cpu-baseline --no-validation"
echo "== bench default (as the driver runs it: 5 + 20 steps; then 20 + 200 steps)"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; tail -c 2600 $OUT/bench_driver_form.json; tail -2 $OUT/bench.err
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 300 $OUT/bench.json; echo
echo "== the product loop (train.train) at batch 1024"
timeout 600 python tools/train_loop_throughput.py 2000 2>/dev/null | grep "train.train" | tee $OUT/train_loop_throughput.txt
echo "== synchronous sampler"
timeout 600 python bench.py $Q --no-prefetch > $OUT/bench_sync_sampler.json 2>/dev/null; head -c 200 $OUT/bench_sync_sampler.json; echo
echo "== bench inception / notebook / generic"
timeout 900 python bench.py --model inception --steps 100 --warmup 10 > $OUT/bench_inception.json 2> $OUT/bench_inception.err; head -c 300 $OUT/bench_inception.json; echo
MWW_BENCH_OPTIONS=graph_static_shapes=0,graph_planar=0 timeout 900 python bench.py --model inception --steps 100 --warmup 10 $Q > $OUT/bench_inception_runtime_shapes.json 2>/dev/null; head -c 200 $OUT/bench_inception_runtime_shapes.json; echo
MWW_BENCH_OPTIONS=graph_planar=0 timeout 900 python bench.py --model inception --steps 100 --warmup 10 $Q > $OUT/bench_inception_interleaved.json 2>/dev/null; head -c 200 $OUT/bench_inception_interleaved.json; echo
MWW_BENCH_FUSED_INPUT=0 timeout 900 python bench.py --model inception --steps 100 --warmup 10 $Q > $OUT/bench_inception_materialised_input.json 2>/dev/null; head -c 200 $OUT/bench_inception_materialised_input.json; echo
timeout 900 python bench.py --model notebook $Q > $OUT/bench_notebook.json 2> $OUT/bench_notebook.err; head -c 300 $OUT/bench_notebook.json; echo
MWW_BENCH_T=194 timeout 900 python bench.py --model notebook $Q > $OUT/bench_notebook_T194_one_full_tile.json 2>/dev/null; head -c 200 $OUT/bench_notebook_T194_one_full_tile.json; echo
timeout 900 python bench.py --force-generic $Q --steps 100 --warmup 10 > $OUT/bench_mixednet_on_graph_kernels.json 2>/dev/null; head -c 200 $OUT/bench_mixednet_on_graph_kernels.json; echo
echo "== bf16 modes (configs[4]), batch 1024 and 4096; fp32 at 4096"
timeout 900 python bench.py --pointwise-bf16 $Q > $OUT/bench_pointwise_bf16.json 2>/dev/null; head -c 200 $OUT/bench_pointwise_bf16.json; echo
timeout 600 python bench.py --storage-bf16 $Q > $OUT/bench_bf16_storage.json 2>/dev/null; head -c 200 $OUT/bench_bf16_storage.json; echo
for m in "--storage-bf16:bf16_storage" "--pointwise-bf16:bf16" ":f32"; do
  timeout 600 python bench.py ${m%%:*} --batch 4096 --steps 100 --warmup 10 $Q > $OUT/bench_${m##*:}_b4096.json 2>/dev/null; head -c 200 $OUT/bench_${m##*:}_b4096.json; echo
done
echo "== collective path forced on one GPU (RCCL world of one): library RCCL one / two buckets, sync-BN; torch.distributed callback"
for a in ":one_bucket" "--grad-buckets 2:two_buckets" "--sync-bn:sync_bn" "--torch-collectives:torch_one_bucket" "--torch-collectives --grad-buckets 2:torch_two_buckets"; do
  MWW_BENCH_FORCE_DP=1 timeout 600 python bench.py $Q ${a%%:*} 2> $OUT/bench_dp.err > $OUT/bench_forced_dp_${a##*:}.json; head -c 160 $OUT/bench_forced_dp_${a##*:}.json; echo
done
echo "== batch sweep"
for b in 256 512 2048 4096; do timeout 300 python bench.py --batch $b --steps 100 --warmup 10 $Q --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', d['value'], d['ms_per_step'], d['roofline']['step_frac'])" | tee -a $OUT/batch_sweep.txt; done
for b in 256 4096; do timeout 300 python bench.py --model inception --batch $b --steps 60 --warmup 6 $Q --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inception batch $b', d['value'], d['ms_per_step'], d['roofline']['step_frac'])" | tee -a $OUT/batch_sweep_inception.txt; done
echo "== rocprofv3"
export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-validation --profile-steps 0"
# the kernel-time summary over 10 + 50 steps: the averages are those of a warm device, the first call's one-off cost is 1/60 of them
BS="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BS > /dev/null 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o p -- $B > /dev/null 2> $OUT/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $B > /dev/null 2> $OUT/pmc2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $B > /dev/null 2> $OUT/pmc3.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $B > /dev/null 2> $OUT/pmc4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inc -o t -- $BS --model inception > /dev/null 2> $OUT/trace_inc.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nb -o t -- $BS --model notebook > /dev/null 2> $OUT/trace_nb.err
# counter passes of the other two topologies (SQ set, FETCH_SIZE, WRITE_SIZE: each in its own run)
for m in inception notebook; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_$m -o p -- $B --model $m > /dev/null 2> $OUT/pmc1_$m.err
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3_$m -o p -- $B --model $m > /dev/null 2> $OUT/pmc3_$m.err
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4_$m -o p -- $B --model $m > /dev/null 2> $OUT/pmc4_$m.err
done
cd $R
python tools/pmc_summary.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/kernel_stats_and_pmc.txt 2>&1
python tools/pmc_summary.py $OUT/trace_inc $OUT/pmc1_inception $OUT/pmc3_inception $OUT/pmc4_inception > $OUT/kernel_stats_and_pmc_inception.txt 2>&1
python tools/pmc_summary.py $OUT/trace_nb $OUT/pmc1_notebook $OUT/pmc3_notebook $OUT/pmc4_notebook > $OUT/kernel_stats_and_pmc_notebook.txt 2>&1
head -14 $OUT/kernel_stats_and_pmc.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +12M -delete
echo "== done"
