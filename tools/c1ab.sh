# developer tool (GPU box): config 1 under environment settings, e.g. tools/c1ab.sh DRT_HEAVY_DIV=8 DRT_HEAVY_DIV=0
for kv in "$@"; do
  env $kv python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv config 1', 'ms_per_step %.3f'%d['ms_per_step'])"
done
