#!/bin/bash
# Developer tool (GPU box): counter passes of the one-wavelength 85 801-cell step (the headline's kernel), one rocprofv3 pass per group.
set -u
tag=$1
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
n=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$((n+1))
  rm -rf gpurun_out/${tag}_p1_$n
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_p1_$n -o p -- python3 bench.py --max-lvl 5 --nlambda 1 --tau 1.0 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_p1_$n.log 2>&1 || { echo "pass [$c] failed"; tail -3 gpurun_out/${tag}_p1_$n.log; }
  echo "pass [$c] done"
done
