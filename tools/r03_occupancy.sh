#!/bin/bash
# occupancy curve of the resident trace kernel on the current build: 1024 / 768 / 512-thread workgroups = 4 / 3 / 2 waves per SIMD
# (diagnostic builds -DSQ_RESIDENT_BLOCK=768 / 512 as lib_w3.so / lib_w2.so); frame and trace-kernel time per frame, each library in its own process
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for r in 1 2; do for lib in libsquigly_hip.so lib_w3.so lib_w2.so; do
  echo "== $lib"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_ab_options.py reps=4 rounds=1 2>&1 | grep "best"
done; done
