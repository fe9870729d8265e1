# developer tool (GPU box): DRT_HEAVY_DIV (limit of a task = ideal launch time / div) on config 1, the one-wavelength step and config 2
run() { env $1 python bench.py $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 [$2]', 'ms_per_step %.3f'%d['ms_per_step'])"; }
for d in 1 2 3 4 6; do run DRT_HEAVY_DIV=$d "--config 1 --steps 20 --warmup 5"; done
for d in 4 8 16 32; do run DRT_HEAVY_DIV=$d "--max-lvl 5 --nlambda 1 --tau 1.0 --steps 3 --warmup 2"; done
for d in 8 16 32; do run DRT_HEAVY_DIV=$d "--steps 2 --warmup 2"; done
