export HSA_ENABLE_IPC_MODE_LEGACY=0
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["gpu_stream_ms_per_step"])'
for sd in 1 0 1 0; do echo side=$sd; MWW_BENCH_SIDE_STREAM=$sd timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "$P"; done
