#!/bin/bash
# Developer tool (GPU box): rocprofv3 counter passes of one command, one pass per counter group, CSV output under
# gpurun_out/<tag>_<n>/. usage: tools/pmc_passes.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- python3 prog args
# The program itself follows "--" (never a shell or env wrapper: the profiler has initialised the GPU by then).
set -u
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
n=0
for p in "${passes[@]}"; do
  n=$((n+1))
  out=gpurun_out/${tag}_$n
  rm -rf "$out"
  rocprofv3 --pmc $p --output-format csv -d "$out" -o p -- "$@" > "gpurun_out/${tag}_$n.log" 2>&1 || { echo "pass $n failed"; tail -5 "gpurun_out/${tag}_$n.log"; exit 1; }
done
python3 tools/pmc_sum.py gpurun_out/${tag}_*/ 
