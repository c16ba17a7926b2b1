#!/bin/bash
# flat leaf-open / window tail / idle ballot: parity (GPU suite), A/B against the previous commit's library, then a fuzz campaign on the
# flat-steps build (output straight into files: a run that is silent for 7 minutes is taken to be hung)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03y_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03y_pytest.log | tail -3
[ $rc -ne 0 ] && { tail -40 $O/r03y_pytest.log; exit $rc; }
ROUNDS=3 bash tools/ab_libs.sh lib_prev.so libsquigly_hip.so 2>&1 | tee $O/r03y_flat2_ab.txt
timeout -k 10 420 python tests/fuzz_gpu.py ${1:-300} 42000000 > $O/r03y_fuzz_small.log 2>&1; tail -1 $O/r03y_fuzz_small.log
timeout -k 10 520 python tests/fuzz_gpu.py ${2:-300} 43000000 big > $O/r03y_fuzz_big.log 2>&1; tail -1 $O/r03y_fuzz_big.log
