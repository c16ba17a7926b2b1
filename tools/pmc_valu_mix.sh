#!/bin/bash
# VALU instruction classes of the trace kernel on ONE headline frame (one rocprofv3 --pmc pass per group; counters only).
#   usage (on the GPU box): bash tools/pmc_valu_mix.sh <tag>
set -u
TAG=${1:-mix}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
            "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
            "SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-include-regex "sq_" --output-format csv -d $OUT/pass$i -- python tools/gpu_frames.py scene=obj frames=1 > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot.setdefault((kn, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    o.write("# python tools/gpu_frames.py scene=obj frames=1 ; per kernel and counter: launches, sum over launches (one --pmc pass per group)\n")
    for (kn, k), v in tot.items():
        line = f"{kn:60s} {k:28s} launches={len(v):3d} sum={sum(v):.6g}"
        print(line); o.write(line + "\n")
PY
