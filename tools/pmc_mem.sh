#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <gpu_frames.py args...> : vector-memory path counters of the trace kernel (TA / TCP / TD / TCC busy and
# stall cycles, L1 accesses, L1->L2 requests and their latency), one --pmc pass per group (at most two counters per TA / TCP / TD
# block: more are refused, and a refused rocprofv3 does not exit by itself), never combined with tracing;
# summary in gpurun_out/pmcm_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTRS in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
            "TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TD_TD_BUSY_sum TD_TC_STALL_sum TCC_BUSY_sum TCC_REQ_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 --pmc $CTRS --kernel-include-regex "sq_trace" --output-format csv -d $OUT/pass$i -- python tools/gpu_frames.py "$@" > $OUT/pass$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"; [ $rc -ne 0 ] && { grep -m2 "exceeds\|error" $OUT/pass$i.log; echo "stopping: a counter group was refused or the run failed"; break; }
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
        line = f"{kn:60s} {k:40s} launches={len(v):3d} sum={sum(v):.6g}"
        print(line); o.write(line + "\n")
PY
