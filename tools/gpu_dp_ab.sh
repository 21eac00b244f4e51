export HSA_ENABLE_IPC_MODE_LEGACY=0
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["gpu_stream_ms_per_step"])'
for pf in 1 0 1 0; do MWW_BENCH_DP_PREFETCH=$pf MWW_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "$P"; done
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "$P"
