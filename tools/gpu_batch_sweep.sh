export HSA_ENABLE_IPC_MODE_LEGACY=0
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["global_batch"], d["value"], d["ms_per_step"], d["roofline"]["step_frac"])'
for b in 256 512 1024 2048 4096; do timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "$P"; done
