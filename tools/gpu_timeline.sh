#!/bin/bash
# Start / end of every kernel of the LAST frame of a run (rocprofv3 --kernel-trace), to see gaps between launches.
#   usage: bash tools/gpu_timeline.sh <tag> [gpu_frames.py args...]
TAG=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/tl_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$TAG -- python tools/gpu_frames.py "$@" > gpurun_out/tl_$TAG.log 2>&1
python - "$TAG" "$@" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/tl_{tag}/*/*kernel_trace.csv")
rows = [r for r in csv.DictReader(open(f[0]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prim = [i for i, r in enumerate(rows) if "primary" in r["Kernel_Name"]]
# back up to the memsets that precede the last primary pass
i0 = prim[-1]
while i0 > 0 and "fillBuffer" in rows[i0 - 1]["Kernel_Name"]: i0 -= 1
rows = rows[i0:]; t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
print(f"== {tag}: {' '.join(sys.argv[2:])}")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev_end)/1e3:6.1f}  {r['Kernel_Name'][:52]}")
    busy += e - s; prev_end = max(prev_end, e)
print(f"  total {(prev_end-t0)/1e3:.1f} us, kernels {busy/1e3:.1f} us, gaps {(prev_end-t0-busy)/1e3:.1f} us")
PY
