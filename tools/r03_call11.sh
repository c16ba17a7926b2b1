#!/bin/bash
# merged branch records in the streaming form: parity on the big scenes, then same-process A/B (0 = two tables, 5 = packed, 8 = line-aligned)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 600 python -m pytest tests/test_big_scenes.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/r03t_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03t_pytest.log | tail -3
[ $rc -ne 0 ] && { tail -40 $O/r03t_pytest.log; exit $rc; }
for scene in hf708 blob6; do
  timeout -k 10 400 python tools/gpu_ab_options.py scene=$scene spp=64 reps=3 rounds=2 -- merged_branches=0 merged_branches=5 merged_branches=8 2>&1 | grep -v amdgpu.ids
done | tee $O/r03t_merged_branches_ab.txt
