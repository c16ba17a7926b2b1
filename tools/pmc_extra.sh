#!/bin/bash
# usage: tools/pmc_extra.sh <tag> <gpu_frames.py args...> : LDS queueing / latency / fetch counters of the trace kernel
# (counters only, never combined with tracing); summary in gpurun_out/pmcx_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcx_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
            "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
            "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU"; do
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
