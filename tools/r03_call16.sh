#!/bin/bash
# tiled enumeration of the primary rays: parity (GPU suite), then A/B of option primary_tiles on the whole frame, on one rank's
# share at 8 ranks and on the two stand-in scenes, with the per-kernel time of the primary pass from a kernel trace
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03zw_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03zw_pytest.log | tail -3
[ $rc -ne 0 ] && { tail -40 $O/r03zw_pytest.log; exit $rc; }
{
timeout -k 10 300 python tools/gpu_ab_options.py reps=4 rounds=2 -- primary_tiles=0 primary_tiles=1
timeout -k 10 300 python tools/gpu_ab_options.py reps=6 rounds=2 shard=2,3,8 -- primary_tiles=0 primary_tiles=1
timeout -k 10 300 python tools/gpu_ab_options.py scene=hf708 spp=64 reps=3 rounds=2 -- primary_tiles=0 primary_tiles=1
timeout -k 10 300 python tools/gpu_ab_options.py scene=blob6 spp=64 reps=3 rounds=2 -- primary_tiles=0 primary_tiles=1
} 2>&1 | grep -v amdgpu.ids | tee $O/r03zw_primary_tiles_ab.txt
for t in 0 1; do bash tools/kernel_times.sh pt$t product scene=obj frames=3 spp=256 shard=2,3,8 primary_tiles=$t 2>&1 | grep -i "primary\|==" ; done | tee -a $O/r03zw_primary_tiles_ab.txt
