# developer tool (GPU box): config 1 (TRUST-I tau = 0.01, one wavelength, 21 801 cells) by sectors per task and phase window
run() { env "$@" python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step %.3f'%d['ms_per_step'])"; }
run DRT_X=0
run DRT_SECTORS_WANT=8
run DRT_SECTORS_WANT=16
run DRT_SECTORS_WANT=16 DRT_PHASE_WINDOW=0
run DRT_SECTORS_WANT=8 DRT_PHASE_WINDOW=0
run DRT_SECTORS_WANT=2
