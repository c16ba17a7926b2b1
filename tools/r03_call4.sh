#!/bin/bash
# Round 3, GPU call 4: the streaming form's new defaults (binary16 culling boxes, six-wave build with three workgroups per CU), and the
# overlapped schedules with a low-priority second stream and s_setprio on the trace waves.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== parity"; timeout -k 10 900 python -m pytest tests/test_big_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "streaming or variants or campaign or pooled or axis or origin or large" 2>&1 | tail -2
for scene in blob6 hf708; do for i in 1 2; do
  echo "-- $scene default"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 | tail -1
  echo "-- $scene two workgroups per CU (plain build)"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 trace_blocks_per_cu=2 | tail -1
done; done 2>&1 | grep -v amdgpu.ids > $O/r03d_stream_defaults.txt
cat $O/r03d_stream_defaults.txt
echo "== overlapped schedules with priorities"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d_overlap_priorities.txt
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
sqt = importlib.import_module("squigly-trace_amd"); import torch
data = "data"
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); w, h, n = 1920, 1080, 256
print("build", sqt.build_id())
ref = None
for rep in range(2):
  for overlap, low, prio in ((0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 1, 2), (1, 1, 3), (2, 0, 0), (2, 1, 0), (2, 1, 2), (0, 1, 2), (0, 1, 0)):
    ds.set_option("aux_low_priority", low); ds.set_option("overlap", overlap); ds.set_option("trace_prio", prio); ds.set_option("coresidency", 0)
    a, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    if ref is None: ref = r.clone()
    same = bool((r == ref).all())
    best = 1e9
    for _ in range(4):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    ds.set_option("coresidency", 1); ds.stats(reset=True); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); st = ds.stats(reset=True)
    print(f"overlap={overlap} aux_low_priority={low} trace_prio={prio}: {best*1e3:.2f} ms same_image={same} | per-sample waves {st[25]}, started beside {100.0*st[26]/max(st[25],1):.1f} %, ended beside {100.0*st[27]/max(st[25],1):.1f} %", flush=True)
PY
echo "== timeline overlap=1 low priority prio 2"
rm -rf $O/r03d_tl1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03d_tl1 -- python tools/gpu_frames.py scene=obj frames=2 spp=256 overlap=1 trace_prio=2 > $O/r03d_tl1.log 2>&1; echo "rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03d_tl1/*/*kernel_trace.csv")
rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith(("sq_", "void sq_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prim = [i for i, r in enumerate(rows) if "primary" in r["Kernel_Name"]]
rows = rows[prim[-1]:]; t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-t0)/1e6:8.3f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}")
PY
