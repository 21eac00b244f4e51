#!/bin/bash
# Round 3, first GPU session: the new entry points, everything that shipped un-run at the end of round 2, and the data for
# this round's kernel decisions.  usage (repo root): bash tools/gpu_r3a.sh [tag]
TAG=${1:-r3a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-validation"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('$1', 'ms/step', d['ms_per_step'], 'gpu', d['gpu_stream_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], {n:round(v*1e3,1) for n,v in k.items()})"; }
echo "== new GPU tests (prefetcher, library RCCL, callback stream guard)"
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "prefetch or rccl or communicator" 2>&1 | tail -4 | tee $OUT/pytest_new.log
echo "== frame-chunk kernels: parity (tools/gpu_chunk_probe.py), then timing"
timeout 300 python tools/gpu_chunk_probe.py > $OUT/chunk_probe.log 2>&1; tail -4 $OUT/chunk_probe.log
for ch in 0 1 2 3; do
  MWW_BENCH_OPTIONS=graph_frame_chunks=$ch timeout 300 python bench.py --model inception $Q --profile-steps 0 --steps 150 --warmup 15 2>/dev/null | line inception_chunks=$ch
  MWW_BENCH_OPTIONS=graph_frame_chunks=$ch timeout 300 python bench.py --force-generic $Q --profile-steps 0 --steps 150 --warmup 15 2>/dev/null | line generic_chunks=$ch
done 2>&1 | tee $OUT/chunks_ab.txt
echo "== graph-MixedNet checker (tightened: Adam moments re-synchronised every step) on random flag sets"
timeout 400 python tools/gpu_topo_fuzz.py 0 150 2>&1 | tail -5 | tee $OUT/topo_fuzz.log
echo "== default bench: worker-thread sampler vs synchronous, driver-sized (20 steps) and 200 steps"
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | tee $OUT/bench20_$rep.json | line prefetch_20
  timeout 300 python bench.py --steps 20 --warmup 5 $Q --no-prefetch 2>/dev/null | line sync_20
  timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | tee $OUT/bench200_$rep.json | line prefetch_200
  timeout 300 python bench.py --steps 200 --warmup 20 $Q --no-prefetch 2>/dev/null | line sync_200
done 2>&1 | tee $OUT/prefetch_ab.txt
echo "== forced data-parallel path on one GPU: library RCCL (1 / 2 buckets, sync-BN) vs the torch.distributed callback"
for a in "" "--grad-buckets 2" "--sync-bn" "--torch-collectives" "--torch-collectives --grad-buckets 2"; do
  MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 200 --warmup 20 $Q $a 2>$OUT/dp.err | tee "$OUT/bench_dp$(echo $a | tr -d ' -').json" | line "dp[$a]"
done 2>&1 | tee $OUT/dp_ab.txt
echo "== notebook topology"
timeout 300 python bench.py --model notebook --steps 200 --warmup 20 $Q 2>/dev/null | tee $OUT/bench_notebook.json | line notebook
echo "== phase ablation of the backward block kernels (profile build, results invalid by construction)"
for m in 0 1 2 4 32 64 128 192 224 7; do
  MWW_HIP_LIB=$R/microwakeword_amd/libmww_prof.so timeout 300 python bench.py --steps 40 --warmup 5 $Q --no-prefetch --ablate $m 2>/dev/null | line "ablate=$m"
done 2>&1 | tee $OUT/ablation.txt
echo "== variants (slim builds, alternating)"
if [ -f $R/microwakeword_amd/libmww_vbase.so ]; then
  for rep in 1 2 3; do
    for v in $VARIANTS; do
      MWW_HIP_LIB=$R/microwakeword_amd/libmww_$v.so timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | line "variant=$v"
    done
  done 2>&1 | tee $OUT/variants.txt
fi
echo "== done"
