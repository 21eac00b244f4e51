export HSA_ENABLE_IPC_MODE_LEGACY=0
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for gf in 512 768 1024; do for gb in 256 384 512; do echo -n "grid_fwd=$gf grid_bwd=$gb  "; timeout 300 python bench.py --grid-fwd $gf --grid-bwd $gb --steps 150 --warmup 15 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "$P"; done; done
for gh in 256 512; do echo -n "grid_head=$gh  "; timeout 300 python bench.py --grid-head $gh --steps 150 --warmup 15 --no-cpu-baseline --no-validation --profile-steps 0 2>/dev/null | python -c "$P"; done
