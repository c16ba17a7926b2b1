"""Parity on small frames + timing on 1080p@64 (development aid)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, pyoracle as O
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ob = O.BIH(O.tris_from_obj(os.path.join(data, "scene.obj"), data)); oc = O.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); ds.enable_timing()
ok = True
for (w, h, n) in [(96, 96, 16), (200, 120, 33)]:
    o_avg, _, _ = ob.render(oc, n, w, h, threads=16)
    for res in (1, 0):
        ds.set_option("resident", res); ds.set_option("slots", w * h * 5)
        a, _ = ds.render_rows(cam, n, w, h); torch.cuda.synchronize()
        same = np.array_equal(a.cpu().numpy().view(np.uint32), o_avg.view(np.uint32)); ok &= same
        print(f"{w}x{h}@{n} resident={res}: parity={same}", flush=True)
ds.set_option("slots", 512 << 20)
w, h, n = 1920, 1080, 64
for res in (1, 0):
    ds.set_option("resident", res)
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing(); ds.stats(reset=True)
    best = 1e9
    for _ in range(3):
        t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); best = min(best, time.time() - t)
    ms, cnt, _ = ds.kernel_timing(); rays = ds.stats(reset=True)[0] / 3
    print(f"resident={res}: {w*h*n/best/1e6:.1f} Msamples/s (trace {ms*cnt/3:.1f} ms of {best*1e3:.1f}; {rays/1e6:.1f} Mrays)", flush=True)
sys.exit(0 if ok else 1)
