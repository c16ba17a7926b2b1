#!/bin/bash
# Round 3, GPU call 3: incremental slab with hoisted loads (streaming), kernel timeline of the overlapped schedules,
# shader clock under the headline load, co-residency with one per-sample workgroup per CU.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== parity"; timeout -k 10 600 python -m pytest tests/test_big_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "streaming or variants or campaign or pooled or axis or origin" 2>&1 | tail -2
for scene in blob6 hf708; do for i in 1 2; do
  for lib in libsquigly_hip.so libc16.so; do for inc in 0 1; do
    echo "-- $scene $lib incremental=$inc"; SQ_LIB_PATH=$PWD/squigly-trace_amd/$lib timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 incremental=$inc | tail -1
  done; done
  for inc in 0 1; do
    echo "-- $scene libw6c16.so 3WG incremental=$inc"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libw6c16.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 incremental=$inc trace_blocks_per_cu=3 lds_node_kb=12 | tail -1
  done
done; done 2>&1 | grep -v amdgpu.ids > $O/r03c_incremental.txt
cat $O/r03c_incremental.txt
echo "== clocks under load"
( timeout -k 10 120 python tools/gpu_frames.py scene=obj frames=400 spp=256 > $O/r03c_load.txt 2>&1 & )
sleep 25
for k in 1 2 3; do rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -4; sleep 2; done > $O/r03c_clocks.txt 2>&1
rocm-smi --showperflevel --showpower 2>&1 | grep -v "^=" | head -12 >> $O/r03c_clocks.txt
cat $O/r03c_clocks.txt
wait; sleep 20; tail -2 $O/r03c_load.txt
echo "== timeline overlap=1 and 2"
for ov in 1 2; do
  rm -rf $O/r03c_tl$ov
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03c_tl$ov -- python tools/gpu_frames.py scene=obj frames=2 spp=256 overlap=$ov > $O/r03c_tl$ov.log 2>&1; echo "rc=$?"
done
python - <<'PY'
import csv, glob
for ov in (1, 2):
    f = glob.glob(f"gpurun_out/r03c_tl{ov}/*/*kernel_trace.csv")
    if not f: print("no trace", ov); continue
    rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith(("sq_", "void sq_"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    # last frame only: kernels after the second sq_primary
    prim = [i for i, r in enumerate(rows) if "primary" in r["Kernel_Name"]]
    rows = rows[prim[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    print(f"overlap={ov}: kernels of the last frame (start, end in ms, queue)")
    for r in rows:
        print(f"  {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-t0)/1e6:8.3f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}")
PY
echo "== co-residency, one per-sample workgroup per CU"
python - <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
sqt = importlib.import_module("squigly-trace_amd"); import torch
data = "data"
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); w, h, n = 1920, 1080, 256
for overlap, aux in ((0, 0), (1, 1), (2, 1), (1, 2), (2, 2), (2, 0), (0, 0)):
    ds.set_option("overlap", overlap); ds.set_option("aux_blocks_per_cu", aux); ds.set_option("coresidency", 0)
    ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    ds.set_option("coresidency", 1); ds.stats(reset=True); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); st = ds.stats(reset=True)
    print(f"overlap={overlap} aux={aux}: {best*1e3:.2f} ms | per-sample waves {st[25]}, started beside {100.0*st[26]/max(st[25],1):.1f} %, ended beside {100.0*st[27]/max(st[25],1):.1f} %", flush=True)
PY
