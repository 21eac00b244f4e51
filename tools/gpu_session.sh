#!/bin/bash
# ONE parameterised runner for everything a GPU lease is spent on (replaces the per-session gpu_r*.sh / gpu_ab*.sh scripts of
# rounds 2-5).  A session is a list of steps executed in order on the box; every step appends to gpurun_out/<tag>/summary.txt.
#
#   usage (repo root):  gpurun --timeout N -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
#   steps (arguments after ':' are comma-free words, separated by '+'):
#     libs:<a>+<b>...          print sha256 / source sha of microwakeword_amd/libmww_<a>.so ... (variant libraries are built in the
#                              container first: tools/build_variant.sh <name> -DFLAG..., FULL=1 for every shape)
#     tests[:<lib>[:<k-expr>]] pytest -m gpu on tests/test_engine_gpu.py with MWW_HIP_LIB=<lib> (default hip), optional -k (write '%' for a space)
#     ab:<a>+<b>[+..][:<reps>[:<bench args, '%' for a space>]]   alternating bench.py runs (200 steps), per-kernel event times
#     abopt:<lib>:<optsA>+<optsB>[+..][:<reps>[:<bench args>]]   the same with ONE library and engine options varied (MWW_BENCH_OPTIONS, e.g.
#                              abopt:hip:bwd_first_wide=0+bwd_first_wide=1:3); several options of one arm are joined by ','
#     driver[:<lib>[:<n>]]     the driver's command (python bench.py --steps 20 --warmup 5) n times (default 3)
#     trace[:<lib>[:<bench args>]]   rocprofv3 --kernel-trace --stats, 60 steps: per-kernel averages -> trace_<lib>_kernel_stats.csv
#     pmc[:<lib>[:<bench args>]]     trace + the four PMC passes (SQ x2, FETCH_SIZE, WRITE_SIZE) -> kernel_stats_and_pmc_<lib>.txt
#     bin:<path>[:<args>]      run a prebuilt micro-benchmark binary (tools/ubench/...), output into the summary
#     py:<script>[:<args>]     python <script> <args> (the fuzz sweeps under tools/), last lines of its output into the summary
#     benchline:<name>[:<lib>[:<bench args>]]  one bench.py JSON line saved as bench_<name>.json
TAG=${1:-session}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
lib() { echo $R/microwakeword_amd/libmww_${1:-hip}.so; }
sp() { echo "${1//%/ }"; }
NOX="--no-cpu-baseline --no-validation --no-batch-sweep --profile-steps 0 --range-repeats 0"
for step in "$@"; do
  IFS=':' read -r what a1 a2 a3 <<< "$step"
  echo "== $step" >> $S
  case $what in
    libs)
      for v in ${a1//+/ }; do python -c "
import sys; sys.path.insert(0, '$R')
from microwakeword_amd import build_native as b
p='$(lib $v)'; print('$v', 'sha256_16', b.library_sha16(p), 'source', b.library_source_sha16(p))" >> $S; done ;;
    tests)
      K=$(sp "$a2")
      MWW_HIP_LIB=$(lib $a1) timeout 2400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider ${K:+-k "$K"} > $OUT/pytest_${a1:-hip}.log 2>&1
      echo "tests ${a1:-hip} [$K]: $(grep -E 'passed|failed|error' $OUT/pytest_${a1:-hip}.log | tail -1)" >> $S
      grep -E "^(FAILED|ERROR)" $OUT/pytest_${a1:-hip}.log | head -5 >> $S ;;
    ab)
      ARGS=$(sp "$a3")
      for rep in $(seq 1 ${a2:-3}); do
        for v in ${a1//+/ }; do
          MWW_HIP_LIB=$(lib $v) timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-validation --no-batch-sweep --range-repeats 0 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernel_ms',{})
print('$v', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" >> $S
        done
      done ;;
    abopt)
      IFS=':' read -r what a1 a2 a3 a4 <<< "$step"
      ARGS=$(sp "$a4")
      for rep in $(seq 1 ${a3:-3}); do
        for o in ${a2//+/ }; do
          MWW_BENCH_OPTIONS="$o" MWW_HIP_LIB=$(lib $a1) timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-validation --no-batch-sweep --range-repeats 0 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernel_ms',{})
print('$a1 [$o]', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {n:round(v*1e3,1) for n,v in k.items()})" >> $S
        done
      done ;;
    driver)
      for rep in $(seq 1 ${a2:-3}); do
        MWW_HIP_LIB=$(lib $a1) timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null > $OUT/bench_driver_form_${a1:-hip}_$rep.json
        python -c "
import json
d=json.loads(open('$OUT/bench_driver_form_${a1:-hip}_$rep.json').read().strip().splitlines()[-1])
print('driver-form ${a1:-hip}', d['ms_per_step'], 'value', d['value'], 'step_frac', d['roofline'].get('step_frac'), 'b4096', (d.get('batch_sweep') or {}).get('4096'))" >> $S
      done ;;
    trace|pmc)
      ARGS=$(sp "$a2"); v=${a1:-hip}
      BS="python $R/bench.py --steps 60 --warmup 10 $NOX $ARGS"
      B="python $R/bench.py --steps 6 --warmup 2 $NOX $ARGS"
      cd /tmp
      MWW_HIP_LIB=$(lib $v) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- $BS > $OUT/trace_$v.json 2> $OUT/trace_$v.err
      cp $OUT/trace_$v/t_kernel_stats.csv $OUT/trace_${v}_kernel_stats.csv 2>/dev/null
      python - >> $S <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace_$v/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$v: kernel time per step %.1f us (70 steps traced)" % (tot / 70 / 1000))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("   %-86s calls %5s avg %7.2f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1000))
PY
      if [ $what = pmc ]; then
        MWW_HIP_LIB=$(lib $v) timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1_$v -o p -- $B > /dev/null 2> $OUT/pmc1_$v.err
        MWW_HIP_LIB=$(lib $v) timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2_$v -o p -- $B > /dev/null 2> $OUT/pmc2_$v.err
        MWW_HIP_LIB=$(lib $v) timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3_$v -o p -- $B > /dev/null 2> $OUT/pmc3_$v.err
        MWW_HIP_LIB=$(lib $v) timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4_$v -o p -- $B > /dev/null 2> $OUT/pmc4_$v.err
        cd $R
        python tools/pmc_summary.py $OUT/trace_$v $OUT/pmc1_$v $OUT/pmc2_$v $OUT/pmc3_$v $OUT/pmc4_$v > $OUT/kernel_stats_and_pmc_$v.txt 2>&1
        head -3 $OUT/kernel_stats_and_pmc_$v.txt | cut -c1-200 >> $S
      fi
      cd $R ;;
    py)
      timeout 2400 python $R/$a1 $(sp "$a2") > $OUT/py_$(basename $a1 .py).log 2>&1; tail -6 $OUT/py_$(basename $a1 .py).log >> $S ;;
    bin)
      timeout 300 $R/$a1 $(sp "$a2") >> $S 2>&1 ;;
    benchline)
      ARGS=$(sp "$a3")
      MWW_HIP_LIB=$(lib $a2) timeout 600 python bench.py $ARGS 2>/dev/null | tail -1 > $OUT/bench_$a1.json
      python -c "
import json
d=json.loads(open('$OUT/bench_$a1.json').read()); print('$a1', d['ms_per_step'], 'value', d['value'], 'host', d.get('host_enqueue_ms_per_step'))" >> $S ;;
    *) echo "unknown step $what" >> $S ;;
  esac
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $S
