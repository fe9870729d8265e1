#!/bin/bash
# Developer tool (GPU box): the measurement set of the deposit-log variant of the bench (bench.py --deposit-log): bench line, rocprofv3
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC counter passes (separate runs, the program itself after "--").
set -u
tag=$1
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
python3 bench.py --deposit-log --steps 3 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
rm -rf gpurun_out/${tag}_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -o kt -- python3 bench.py --deposit-log --steps 2 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_kt.err || exit 1
n=0
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  n=$((n+1))
  rm -rf gpurun_out/${tag}_pmc_$n
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$n -o p -- python3 bench.py --deposit-log --steps 1 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_pmc_$n.log 2>&1 || { echo "pass $c failed"; tail -3 gpurun_out/${tag}_pmc_$n.log; }
  echo "pmc pass $c done"
done
