#!/bin/bash
# Same-box A/B: the product library against variant builds (python squigly-trace_amd/build.py --out=libX.so -DFLAG),
# all in ONE gpurun call because boxes differ by several per cent.
#   usage: [SCENES="obj blob6"] bash tools/ab_variant.sh libA.so [libB.so ...]
set -e
cd "$(dirname "$0")/.."
for scene in ${SCENES:-obj}; do for i in 1 2; do
  echo "== product"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=5 | tail -2
  for lib in "$@"; do
    echo "== $lib"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=5 | tail -2
  done
done; done
