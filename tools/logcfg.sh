# developer tool (GPU box): BASELINE configs 3-5 and the one-wavelength steps by atomics and by the deposit log (default log size)
for log in 0 1; do
  echo "== DRT_LOG=$log"
  DRT_LOG=$log MAX_IT=1 python tools/run_config.py shell 3 28 2>&1 | grep -E "dir_|scatt|total"
  DRT_LOG=$log MAX_IT=1 python tools/run_config.py galaxy 3 16 2>&1 | grep -E "dir_|scatt|total"
  DRT_LOG=$log MAX_IT=1 python tools/run_config.py nbody 3 28 2>&1 | grep -E "dir_|scatt|total"
  DRT_LOG=$log python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 1 ms_per_step %.2f'%d['ms_per_step'])"
  DRT_LOG=$log python bench.py --max-lvl 5 --nlambda 1 --tau 1.0 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 lambda 85801 ms_per_step %.2f'%d['ms_per_step'])"
done
