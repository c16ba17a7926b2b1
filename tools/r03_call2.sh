#!/bin/bash
# Round 3, GPU call 2: incremental slab test in the streaming form -- parity first (product build and the cull16 / 6-wave variants),
# then timing with the option on and off, same box.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for lib in libsquigly_hip.so libc16.so libw6c16.so; do
  echo "== parity $lib"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 600 python -m pytest tests/test_big_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "streaming or large or variants or golden or soup or degenerate or campaign or pooled or axis or origin or culling" 2>&1 | tail -3 || exit 1
done > $O/r03b_parity.txt 2>&1
cat $O/r03b_parity.txt
for scene in blob6 hf708; do for i in 1 2; do
  for lib in libsquigly_hip.so libc16.so; do for inc in 0 1; do
    echo "-- $scene $lib incremental=$inc"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 incremental=$inc | tail -1
  done; done
  for inc in 0 1; do
    echo "-- $scene libw6c16.so 3WG incremental=$inc"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libw6c16.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 incremental=$inc trace_blocks_per_cu=3 lds_node_kb=12 | tail -1
  done
done; done 2>&1 | grep -v amdgpu.ids > $O/r03b_incremental.txt
cat $O/r03b_incremental.txt
