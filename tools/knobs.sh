# developer tool (GPU box): scheduling knobs on the bench step (config 2), steady state
run() { env $1 python bench.py --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step %.1f'%d['ms_per_step'], 'deposits/atomic %.2f'%d['roofline']['ray_wavelength_deposits_per_atomic_instruction'])"; }
for kv in "$@"; do run "$kv"; done
