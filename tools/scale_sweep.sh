#!/bin/bash
# The scaling sweeps of SURVEY 8(d)/(e) on one node, one JSON line per run (the driver's launch form: python -m
# torch.distributed.run, one rank per GPU over RCCL; every line carries config.collective_ranks = ncclCommCount of the library's
# communicator, which bench.py refuses to print if it differs from --gpus):
#   weak    BASELINE configs[2]: fp32, 1024 windows per GPU, N = 1, 2, 4, 8 x {one gradient bucket, two buckets (the overlapped
#           schedule: dense + last two blocks exchanged next to the remaining backward kernels)} x {rank-local BatchNorm, sync-BN}
#   strong  BASELINE configs[4]: bf16-operand MFMA pointwise (and: + bf16 storage of p_k / g_k), a FIXED global batch of 4096
#           windows per step (--global-batch 4096 => 4096 / N per GPU, "scaling": "strong" in the line), N = 1, 2, 4, 8
# The builder's GPU box has one GPU: this script is for the first multi-GPU lease (unmeasured on hardware until then).
# usage (repo root): bash tools/scale_sweep.sh [out.jsonl] [steps] [warmup]     (NGPUS="1 2 4 8", SWEEPS="weak strong" in the environment to restrict)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=${1:-$R/gpurun_out/scale_sweep.jsonl}
STEPS=${2:-200}; WARM=${3:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p $(dirname $OUT); : > $OUT
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
PORT=29531
run() {   # run <N> <label> <bench args...>
  local N=$1 LABEL=$2; shift 2
  PORT=$((PORT + 1))
  if [ "$N" = 1 ]; then
    LINE=$(cd $R && timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-validation --no-batch-sweep "$@" 2>/dev/null | tail -1)
  else
    LINE=$(cd $R && timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
           bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-validation "$@" 2>/dev/null | tail -1)
  fi
  echo "$LINE" >> $OUT
  echo "$LINE" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('N=%d $LABEL: %.4f ms/step, %.0f windows/s (%s scaling, global batch %d), collective_ranks %s' % (d['n_gpus'], d['ms_per_step'], d['value'], d['scaling'], d['config']['global_batch'], d['config'].get('collective_ranks')))
except Exception as e:
    print('N=$N $LABEL: no line (%s)' % e)"
}
for N in ${NGPUS:-1 2 4 8}; do
  [ "$N" -gt "$HAVE" ] && { echo "skipping N=$N ($HAVE GPU(s) visible)"; continue; }
  for SWEEP in ${SWEEPS:-weak strong}; do
    if [ $SWEEP = weak ]; then
      for BUCKETS in 1 2; do
        for BN in local sync; do
          [ "$N" = 1 ] && { [ "$BUCKETS" = 2 ] || [ "$BN" = sync ]; } && continue   # one GPU: no exchange to vary
          EXTRA="--grad-buckets $BUCKETS"; [ "$BN" = sync ] && EXTRA="$EXTRA --sync-bn"
          run $N "weak fp32 buckets=$BUCKETS bn=$BN" $EXTRA
        done
      done
    else
      run $N "strong bf16 operands" --pointwise-bf16 --global-batch 4096
      run $N "strong bf16 operands + storage" --storage-bf16 --global-batch 4096
    fi
  done
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$OUT") if l.strip().startswith("{")]
def key(r):
    return (r["scaling"], r["dtype"][:24])
for k in sorted({key(r) for r in rows}):
    grp = [r for r in rows if key(r) == k]
    base = [r for r in grp if r["n_gpus"] == 1]
    if not base:
        continue
    b = base[0]["value"]
    for r in grp:
        print("%-6s %-26s N=%d %-44s value %.0f  x%.2f of N=1  efficiency %.0f %%" % (k[0], k[1], r["n_gpus"],
              str(r["config"].get("collectives"))[:30] + " bn=" + str(r["config"].get("bn")), r["value"], r["value"] / b, 100 * r["value"] / b / r["n_gpus"]))
PY
