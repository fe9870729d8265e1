#!/bin/bash
# Developer tool (GPU box): instruction and wave counters of the deposit-log kernels on the bench step, one rocprofv3 pass per group.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp DRT_LOG=1 DRT_LOG_GB=100 DRT_LOG_LG=${1:-4}
n=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  n=$((n+1))
  rm -rf gpurun_out/pmclog_$n
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmclog_$n -o p -- python3 bench.py --arena-gb 120 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmclog_$n.log 2>&1 || { echo "pass [$c] failed"; tail -3 gpurun_out/pmclog_$n.log; }
  python3 - $n <<'PY'
import csv,glob,sys,collections
f=glob.glob('gpurun_out/pmclog_%s/**/*counter_collection.csv'%sys.argv[1], recursive=True)
if not f: print('no counter file'); sys.exit()
d=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:28]; d[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in d.items():
    if 'k_log_reduce' in k or 'rt_stage_wide' in k or 'k_log_s' in k or 'k_log_c' in k: print(k, {a:'%.3e'%b for a,b in v.items()})
PY
done
