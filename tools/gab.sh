# developer tool (GPU box): the reduced BASELINE configs 3-5, config 1, the one-wavelength step and config 2 with the library as built
for cfg in "shell 3 28" "galaxy 4 16" "nbody 4 28"; do
  MAX_IT=2 python tools/run_config.py $cfg 2>&1 | grep -E "total"
done
for a in "--config 1 --steps 20 --warmup 5" "--max-lvl 5 --nlambda 1 --tau 1.0 --steps 3 --warmup 2" "--steps 3 --warmup 2"; do
  python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', 'ms_per_step %.3f'%d['ms_per_step'])"
done
