#!/bin/bash
# Round 3, GPU call 7: 45-byte slots (hit and generator words folded into the ray quads, radiance on its own) against 61 and 33.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest -m gpu"; ( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03g_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/r03g_pytest.log | tail -3; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  for lib in libsquigly_hip.so libold61.so libslot33.so; do
    echo "-- $lib"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=obj frames=6 spp=256 | tail -2
  done
done 2>&1 | grep -v amdgpu.ids > $O/r03g_slots_ab.txt
cat $O/r03g_slots_ab.txt
bash tools/kernel_times.sh r03g_new product scene=obj frames=3 spp=256
