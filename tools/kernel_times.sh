#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a few frames with a given library and shard.
#   usage: bash tools/kernel_times.sh <tag> <lib.so|product> [gpu_frames.py args...]
TAG=$1; LIB=$2; shift 2
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
[ "$LIB" != product ] && export SQ_LIB_PATH=$PWD/squigly-trace_amd/$LIB
rm -rf gpurun_out/kt_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$TAG -- python tools/gpu_frames.py "$@" > gpurun_out/kt_$TAG.log 2>&1
f=$(ls gpurun_out/kt_$TAG/*/*kernel_stats.csv | head -1)
echo "== $TAG ($LIB) $*"; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    if 'sq_' in r['Name']: print(f"  {r['Name'][:44]:46s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
