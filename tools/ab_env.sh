# in-call A/B of an environment variable on the bench step: tools/ab_env.sh <nlambda> <VAR> <value> <value> ...
nl=$1; var=$2; shift; shift
for rep in 1 2; do
for v in "$@"; do
  env $var=$v python bench.py --max-lvl 5 --nlambda $nl --tau 1.0 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', 'ms_per_step %.2f kernel_ms %.2f'%(d['ms_per_step'], d['roofline']['kernel_ms']), 'deposits/atomic %.2f'%d['roofline']['ray_wavelength_deposits_per_atomic_instruction'])"
done
done
