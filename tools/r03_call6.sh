#!/bin/bash
# Round 3, GPU call 6: the 33-byte slot layout -- whole GPU suite, then same-box A/B against the 61-byte layout.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest -m gpu"; ( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03f_pytest.log 2>&1; rc=$?; tail -4 $O/r03f_pytest.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  echo "-- 33 B/slot"; timeout -k 10 300 python tools/gpu_frames.py scene=obj frames=5 spp=256 | tail -2
  echo "-- 61 B/slot"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libold61.so timeout -k 10 300 python tools/gpu_frames.py scene=obj frames=5 spp=256 | tail -2
  echo "-- 33 B/slot, one rank's share of eight"; timeout -k 10 300 python tools/gpu_frames.py scene=obj frames=5 spp=256 shard=8,0,8 | tail -2
  echo "-- 61 B/slot, one rank's share of eight"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libold61.so timeout -k 10 300 python tools/gpu_frames.py scene=obj frames=5 spp=256 shard=8,0,8 | tail -2
done 2>&1 | grep -v amdgpu.ids > $O/r03f_slots_ab.txt
cat $O/r03f_slots_ab.txt
bash tools/kernel_times.sh r03f_new product scene=obj frames=3 spp=256
bash tools/kernel_times.sh r03f_old libold61.so scene=obj frames=3 spp=256
