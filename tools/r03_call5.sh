#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for scene in blob6 hf708; do for i in 1 2; do
  echo "-- $scene default"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 | tail -1
  echo "-- $scene two workgroups per CU (plain build)"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=64 trace_blocks_per_cu=2 | tail -1
done; done 2>&1 | grep -v amdgpu.ids > $O/r03e_stream_defaults.txt
cat $O/r03e_stream_defaults.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03e_overlap_polite.txt
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
sqt = importlib.import_module("squigly-trace_amd"); import torch
data = "data"
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); w, h, n = 1920, 1080, 256
print("build", sqt.build_id())
ref = None
for rep in range(2):
  for overlap, polite, prio in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 1, 2), (1, 2, 2), (2, 0, 0), (2, 1, 0), (2, 2, 0), (2, 4, 0), (0, 0, 0)):
    ds.set_option("overlap", overlap); ds.set_option("aux_polite", polite); ds.set_option("trace_prio", prio); ds.set_option("coresidency", 0)
    a, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    if ref is None: ref = r.clone()
    same = bool((r == ref).all())
    best = 1e9
    for _ in range(4):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    ds.set_option("coresidency", 1); ds.stats(reset=True); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); st = ds.stats(reset=True)
    print(f"overlap={overlap} aux_polite={polite} trace_prio={prio}: {best*1e3:.2f} ms same_image={same} | per-sample waves {st[25]}, started beside {100.0*st[26]/max(st[25],1):.1f} %, ended beside {100.0*st[27]/max(st[25],1):.1f} %", flush=True)
PY
