# developer tool (GPU box): the bench step by atomics and by the deposit log, in one call
run() { python bench.py --arena-gb $1 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 arena $1', 'ms_per_step %.1f'%d['ms_per_step'], d['config']['checksum_u_final_first_pass'])"; }
for rep in 1 2; do
DRT_LOG=0 run 200 "atomics"
DRT_LOG=1 DRT_LOG_GB=100 run 120 "log 100"
DRT_LOG=1 DRT_LOG_GB=110 run 110 "log 110"
DRT_LOG=1 DRT_LOG_GB=100 DRT_LIB=$PWD/dart-ray_amd/csrc/libdartray_rt_pow1.so run 120 "log 100 pow1"
done
