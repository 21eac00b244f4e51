#!/bin/bash
# Round 5, session I: the Inception stem changes (kernels_graph.hip.h MWW_G_FWD_DIRECT / MWW_G_WGRAD_EVEN / MWW_G_WGRAD_XG_NARROW) against
# the library they replace, same box, same session: parity tests of each build, rocprofv3 per-kernel averages, alternating bench runs.
#   libmww_base.so    the shipped library before the change (source sha 360650b9ecfe83b3)
#   libmww_hip.so     stem forward from the accumulators + even deal of the weight gradient's task tiles (the defaults)
#   libmww_d2.so      ... every static forward convolution from the accumulators (-DMWW_G_FWD_DIRECT=2)
#   libmww_narrow.so  ... the stem weight gradient at three workgroups per CU (-DMWW_G_WGRAD_XG_NARROW=1)
# The variant libraries are built in the container before the call (they travel with the snapshot, *.so is git-ignored):
#   cp microwakeword_amd/libmww_hip.so microwakeword_amd/libmww_base.so            (before the change)
#   python -m microwakeword_amd.build_native --out microwakeword_amd/libmww_d2.so -- -DMWW_G_FWD_DIRECT=2
#   python -m microwakeword_amd.build_native --out microwakeword_amd/libmww_narrow.so -- -DMWW_G_WGRAD_XG_NARROW=1
#   python -m microwakeword_amd.build_native --out microwakeword_amd/libmww_breg.so -- -DMWW_G_STEM_BREG=1
# usage (repo root): [LIBS="a b" VARIANTS="b" PMCLIBS="a" ALLK=1] bash tools/archive/gpu_r5i.sh <tag>   (second sitting: LIBS="even hip breg", every gconv kernel listed)
TAG=${1:-r5i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
LIBS=${LIBS:-"base hip d2 narrow"}
S=$OUT/summary.txt; : > $S
for v in $LIBS; do python -c "
import sys; sys.path.insert(0, '$R')
from microwakeword_amd import build_native as b
p='$R/microwakeword_amd/libmww_$v.so'; print('$v', 'sha256_16', b.library_sha16(p), 'source', b.library_source_sha16(p))" >> $S; done
# 1. parity of the new default build (float64 oracle at B = 1024, gathers, static shapes, fuzz), then the cheap subset on the variants
K_ALL="inception or fused_input or graph"
K_FAST="inception_train_steps or stem_gathers or static_shapes or inception_forward"
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "$K_ALL" > $OUT/pytest_hip.log 2>&1; echo "tests hip: $(grep -E 'passed|failed|error' $OUT/pytest_hip.log | tail -1)" >> $S
# 2. per-kernel averages (rocprofv3 --kernel-trace --stats), Inception, 60 steps
BS="python $R/bench.py --model inception --steps 60 --warmup 10 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
cd /tmp
for v in $LIBS; do
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- $BS > $OUT/trace_$v.json 2> $OUT/trace_$v.err
  python - >> $S <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace_$v/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== $v: kernel sum per step %.1f us" % (tot / 70 / 1000))
for r in rows:
    n = r["Name"]
    if "GShape<5, 1, 40" in n or "$v" in ("d2", "base") or "$ALLK" == "1":
        if "gconv" in n and ("bwd" not in n or "GShape<5, 1, 40" in n or "$ALLK" == "1"):
            print("   %-70s calls %s avg %.2f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
# (the new stem forward held to two workgroups per CU: is it the third resident workgroup or the shorter epilogue?)
MWW_BENCH_OPTIONS="graph_fwd_wg_per_cu=2" MWW_HIP_LIB=$R/microwakeword_amd/libmww_hip.so timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hip_cap2 -o t -- $BS > /dev/null 2> $OUT/trace_hip_cap2.err
python - >> $S <<PY
import csv
for r in csv.DictReader(open("$OUT/trace_hip_cap2/t_kernel_stats.csv")):
    if "gconv_xg" in r["Name"]: print("== hip, graph_fwd_wg_per_cu=2:  %-50s avg %.2f us" % (r["Name"][:50], float(r["AverageNs"]) / 1000))
PY
cd $R
# 3. alternating bench runs
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $1 ms_per_step', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))"; }
ARGS="--model inception --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0 --steps 100 --warmup 20"
for rep in 1 2; do
  for v in $LIBS; do
    MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so timeout 200 python bench.py $ARGS 2>/dev/null | line $v >> $S
  done
done
# 4. the variants' parity (cheap subset)
for v in ${VARIANTS:-d2 narrow}; do
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "$K_FAST" > $OUT/pytest_$v.log 2>&1; echo "tests $v: $(grep -E 'passed|failed|error' $OUT/pytest_$v.log | tail -1)" >> $S
done
# 5. counters of the two stem kernels (new default and the narrow variant)
B="python $R/bench.py --model inception --steps 6 --warmup 2 --no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0"
cd /tmp
for v in ${PMCLIBS:-hip narrow}; do
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_$v -o p -- $B > /dev/null 2> $OUT/pmc1_$v.err
  (cd $R; python tools/pmc_summary.py $OUT/pmc1_$v | grep "xg_kernel" | sed "s/^/$v: /" | cut -c1-400 >> $S)
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
cat $S
