#!/bin/bash
# Round profile collection on the GPU box: kernel-trace stats and, in separate passes, HBM PMC counters,
# all for the default `python bench.py` workload.  Outputs under gpurun_out/profiles_<tag>/.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 3 --warmup 1 > $OUT/bench_under_trace.log 2>&1
echo "trace rc=$?"
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-include-regex "sq_" --output-format csv -d $OUT/pmc_$C -- python bench.py --steps 1 --warmup 0 --no-cpu > $OUT/pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
done
python - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        tot[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for kn, d in tot.items():
    res[kn] = {c: {"launches": len(v), "sum_KB": sum(v), "mean_KB_per_launch": sum(v) / len(v)} for c, v in d.items()}
tr = res.get("sq_trace_rays", {})
if "FETCH_SIZE" in tr and "WRITE_SIZE" in tr:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a
    # wide coalesced read -> x2 ; WRITE_SIZE is exact.
    fetch = tr["FETCH_SIZE"]["mean_KB_per_launch"] * 1024 * 2
    write = tr["WRITE_SIZE"]["mean_KB_per_launch"] * 1024
    res["hbm_bytes_per_launch"] = fetch + write
    res["note"] = "sq_trace_rays, default bench.py workload; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, mean per launch"
json.dump(res, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
tail -1 $OUT/bench_under_trace.log
