#!/bin/bash
# usage: tools/pmc_frames.sh <tag> <gpu_frames.py args...> : rocprofv3 PMC passes (counters only, never combined with tracing)
# over `python tools/gpu_frames.py <args>` for the trace kernel; summary in gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
            "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-include-regex "sq_trace" --output-format csv -d $OUT/pass$i -- python tools/gpu_frames.py "$@" > $OUT/pass$i.log 2>&1
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
    o.write("# python tools/gpu_frames.py $* ; per counter: launches, sum over launches\n")
    for (kn, k), v in tot.items():
        line = f"{kn:60s} {k:28s} launches={len(v):3d} sum={sum(v):.6g}"
        print(line); o.write(line + "\n")
PY
