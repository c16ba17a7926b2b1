import importlib, os, sys, time
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
for shard in ((None, 0, 1), (8, 0, 8)):
    ref = None
    for pp in (0, 1, 0, 1):
        ds.set_option("primary_pooled", pp)
        best = 1e9
        for i in range(5):
            t = time.time(); _, r = ds.render_rows(cam, 256, 1920, 1080, want_avg=False, shard=shard); torch.cuda.synchronize(); best = min(best, time.time() - t)
        if ref is None: ref = r.clone()
        print(f"shard={shard} primary_pooled={pp}: {best*1e3:.2f} ms same_image={bool((r == ref).all())}", flush=True)
