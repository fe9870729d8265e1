# in-call A/B of library builds (DRT_LIB picks one) on the bench grid: tools/ab_1lambda.sh <nlambda> <lib> <lib> ...
nl=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  DRT_LIB=$PWD/dart-ray_amd/csrc/$lib python bench.py --max-lvl 5 --nlambda $nl --tau 1.0 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms_per_step %.2f kernel_ms %.2f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
done
done
