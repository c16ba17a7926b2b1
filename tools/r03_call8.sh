#!/bin/bash
# Round 3, GPU call 8: touch-prefetch of the near child's record in the streaming form (SQ_STREAM_TOUCH bits 1, 2, 4).
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== parity libtouch7"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libtouch7.so timeout -k 10 600 python -m pytest tests/test_big_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "streaming or variants or campaign or pooled or axis or origin" 2>&1 | tail -2
for scene in blob6 hf708; do for i in 1 2; do
  for lib in libsquigly_hip.so libtouch1.so libtouch3.so libtouch5.so libtouch7.so; do
    echo "-- $scene $lib"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 | tail -1
  done
done; done 2>&1 | grep -v amdgpu.ids > $O/r03h_touch.txt
cat $O/r03h_touch.txt
