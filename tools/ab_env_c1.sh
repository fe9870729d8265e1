var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $var=$v python bench.py --config 1 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', 'ms_per_step %.3f kernel_ms %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
done
done
