#!/bin/bash
# Same-box A/B of variant libraries (python squigly-trace_amd/build.py --out=lib_X.so ...), each in its own process through
# SQ_LIB_PATH, ROUNDS rounds in turn; prints the best and the median frame of every run.
#   usage: [SCENE=obj SPP=256 ROUNDS=2 FRAMES=6 OPTS="key=value ..."] bash tools/ab_libs.sh lib_a.so lib_b.so ...
set -u
cd "$(dirname "$0")/.."
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=${SCENE:-obj} spp=${SPP:-256} frames=${FRAMES:-6} ${OPTS:-} 2>&1 | grep "frame" | tail -n +2 \
      | python3 -c "
import sys, re
t = []; cs = ''
for l in sys.stdin:
    m = re.search(r': ([0-9.]+) ms', l); t.append(float(m.group(1))); cs = l.split()[-1]
t.sort(); print('%-22s best %7.2f ms  median %7.2f ms  checksum %s' % ('$lib', t[0], t[len(t)//2], cs))"
  done
done
