"""Frame time at the headline configuration vs the straggler cut-off of the trace kernel (lanes still
traversing when a wave turns to its leaves)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, 256
ref = None
for strag in (0, 2, 4, 5, 6, 7, 8, 10, 12, 16, 24):
    ds.set_option("straggler_lanes", strag)
    a, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    if ref is None: ref = r.clone()
    same = bool((r == ref).all())
    best = 1e9
    for _ in range(3):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"straggler_lanes={strag}: {best*1e3:.1f} ms -> {w*h*n/best/1e6:.1f} Msamples/s same_image={same}", flush=True)
