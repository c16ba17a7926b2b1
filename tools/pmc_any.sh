#!/bin/bash
# usage: tools/pmc_any.sh <tag> <python script and args...> : PMC passes for an arbitrary script
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-include-regex "sq_trace" --output-format csv -d $OUT/pass$i -- python "$@" > $OUT/pass$i.log 2>&1
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
    for (kn, k), v in tot.items():
        line = f"{kn:34s} {k:30s} launches={len(v):3d} sum={sum(v):.6g}"
        print(line); o.write(line + "\n")
PY
