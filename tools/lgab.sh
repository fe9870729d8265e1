# developer tool (GPU box): the bench step by atomics and by the deposit log, in one call
run() { python bench.py $1 --steps 3 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1]', 'ms_per_step %.1f'%d['ms_per_step'], d['config']['checksum_u_final_first_pass'])"; }
for rep in 1 2; do
run ""
run "--deposit-log"
run "--deposit-log --log-gb 110 --arena-gb 110"
done
