#!/bin/bash
# The scaling sweep of SURVEY 8(e) on one node: bench.py at N = 1, 2, 4, 8 GPUs x {one gradient bucket, two buckets (the
# overlapped schedule: dense + last two blocks exchanged next to the remaining backward kernels)} x {rank-local BatchNorm,
# sync-BN}, one JSON line each (the driver's launch form: python -m torch.distributed.run, one rank per GPU over RCCL).
# Every line carries config.collective_ranks = ncclCommCount of the library's communicator, which bench.py refuses to
# print if it differs from --gpus.  The builder's GPU box has one GPU: this script is for the first multi-GPU lease.
# usage (repo root): bash tools/scale_sweep.sh [out.jsonl] [steps] [warmup]     (NGPUS="1 2 4 8" in the environment to restrict)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=${1:-$R/gpurun_out/scale_sweep.jsonl}
STEPS=${2:-200}; WARM=${3:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p $(dirname $OUT); : > $OUT
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
PORT=29531
for N in ${NGPUS:-1 2 4 8}; do
  [ "$N" -gt "$HAVE" ] && { echo "skipping N=$N ($HAVE GPU(s) visible)"; continue; }
  for BUCKETS in 1 2; do
    for BN in local sync; do
      [ "$N" = 1 ] && { [ "$BUCKETS" = 2 ] || [ "$BN" = sync ]; } && continue   # one GPU: no exchange to vary
      EXTRA="--grad-buckets $BUCKETS"; [ "$BN" = sync ] && EXTRA="$EXTRA --sync-bn"
      PORT=$((PORT + 1))
      if [ "$N" = 1 ]; then
        LINE=$(cd $R && timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-validation --no-batch-sweep 2>/dev/null | tail -1)
      else
        LINE=$(cd $R && timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
               bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-validation $EXTRA 2>/dev/null | tail -1)
      fi
      echo "$LINE" >> $OUT
      echo "$LINE" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('N=%d buckets=$BUCKETS bn=$BN: %.4f ms/step, %.0f windows/s, collective_ranks %s' % (d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('collective_ranks')))
except Exception as e:
    print('N=$N buckets=$BUCKETS bn=$BN: no line (%s)' % e)"
    done
  done
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$OUT") if l.strip().startswith("{")]
base = [r for r in rows if r["n_gpus"] == 1]
if base:
    b = base[0]["value"]
    for r in rows:
        print("N=%d %-40s value %.0f  x%.2f of N=1  efficiency %.0f %%" % (r["n_gpus"], str(r["config"].get("collectives"))[:40] + " bn=" + str(r["config"].get("bn")), r["value"], r["value"] / b, 100 * r["value"] / b / r["n_gpus"]))
PY
