#!/bin/bash
# Round-3 randomised campaign on the kernels as shipped: two processes (16 host cores between them), $1 seconds each.
set -u
SEC=${1:-900}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
python -c "import importlib; print('build', importlib.import_module('squigly-trace_amd').build_id())" 2>/dev/null | tee $O/r03_fuzz_a.log > $O/r03_fuzz_b.log
( timeout -k 10 $((SEC + 120)) python tests/fuzz_gpu.py $SEC ${2:-30000000} >> $O/r03_fuzz_a.log 2>&1; echo "rc=$?" >> $O/r03_fuzz_a.log ) &
( timeout -k 10 $((SEC + 120)) python tests/fuzz_gpu.py $SEC ${3:-30500000} big >> $O/r03_fuzz_b.log 2>&1; echo "rc=$?" >> $O/r03_fuzz_b.log ) &
while [ "$(jobs -r | wc -l)" -gt 0 ]; do sleep 60; tail -q -n 1 $O/r03_fuzz_a.log $O/r03_fuzz_b.log; done
wait
grep -E "MISMATCH|fuzz:|rc=|build" $O/r03_fuzz_a.log $O/r03_fuzz_b.log | grep -v amdgpu
