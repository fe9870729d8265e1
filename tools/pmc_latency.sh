#!/bin/bash
# Developer tool (GPU box): the wave-occupancy / instruction counters of the bench step, one rocprofv3 pass per group.
set -u
tag=$1
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
n=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQC_ICACHE_BUSY_CYCLES"; do
  n=$((n+1))
  rm -rf gpurun_out/${tag}_lat_$n
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_lat_$n -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_lat_$n.log 2>&1 || { echo "pass [$c] failed"; tail -3 gpurun_out/${tag}_lat_$n.log; }
  echo "pass [$c] done"
done
