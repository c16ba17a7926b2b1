"""Timing vs workspace size (sample slots per batch) at the headline configuration."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, 256
for slots_m in (48, 192, 256, 384, 512):
    for strag in (6,):
        ds.set_option("slots", slots_m << 20); ds.set_option("straggler_lanes", strag)
        ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
        print(f"slots={slots_m}Mi straggler={strag}: {best*1e3:.1f} ms -> {w*h*n/best/1e6:.1f} Msamples/s", flush=True)
