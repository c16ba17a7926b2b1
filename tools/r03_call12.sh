#!/bin/bash
# flat (single exec region) branch and return steps of the resident pooled kernel: parity, then A/B against the same source without them
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03x_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03x_pytest.log | tail -3
[ $rc -ne 0 ] && { tail -40 $O/r03x_pytest.log; exit $rc; }
ROUNDS=3 bash tools/ab_libs.sh lib_noflat.so libsquigly_hip.so 2>&1 | tee $O/r03x_flat_ab.txt
