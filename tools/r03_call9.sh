#!/bin/bash
# incremental slab test in the resident form: parity (GPU suite) first, then same-process A/B
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03q_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03q_pytest.log | tail -3
[ $rc -ne 0 ] && { tail -40 $O/r03q_pytest.log; exit $rc; }
timeout -k 10 600 python tools/gpu_ab_options.py reps=4 rounds=2 -- incremental=0,descend_extra=2 incremental=1,descend_extra=2 incremental=1,descend_extra=3 incremental=1,descend_extra=4 2>&1 | grep -v amdgpu.ids | tee $O/r03q_incremental_ab.txt
