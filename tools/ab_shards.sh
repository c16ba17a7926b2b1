#!/bin/bash
# Same-box A/B of the product library against one variant on the whole headline frame and on one rank's share at 8 ranks.
#   usage: bash tools/ab_shards.sh libvariant.so
cd "$(dirname "$0")/.."
for sh in 0,0,1 8,0,8; do for i in 1 2; do
  echo "== product shard=$sh"; timeout -k 10 200 python tools/gpu_frames.py scene=obj frames=6 shard=$sh | tail -1
  echo "== $1 shard=$sh"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$1 timeout -k 10 200 python tools/gpu_frames.py scene=obj frames=6 shard=$sh | tail -1
done; done
