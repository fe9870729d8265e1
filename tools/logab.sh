# developer tool (GPU box): timing of the trace side of the deposit-log path (DRT_LOG=1) on the bench step, reduce pass skipped
# (DRT_LOG_NOREDUCE) - flags of a -DDRT_DEBUG_FLAGS build: 4 no records, 8 records without their C stores, 16 no u_final sums
export DRT_LOG=1 DRT_LOG_GB=100 DRT_LOG_NOREDUCE=1 DRT_LIB=$PWD/dart-ray_amd/csrc/libdartray_rt_dbg.so
run() { DRT_DBG_FLAGS=$1 DRT_LOG_BATCHES=$2 python bench.py --arena-gb 120 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags $1 batches $2', 'ms_per_step %.1f'%d['ms_per_step'])"; }
run 20 1
run 4 1
run 4 6
run 8 6
run 0 6
run 0 12
