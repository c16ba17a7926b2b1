"""BASELINE.json configs[0..4] end to end on ONE GPU: load, BIH build, scene upload, render (resident API,
second frame timed).  C3 and C5 use the procedural stand-ins of tools/gen_scenes.py (no bunny / no 1M-triangle
mesh ships with the reference); C4 and C5 are 8-GPU configurations upstream, run here on one device."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes as G
sqt = importlib.import_module("squigly-trace_amd")
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
data = os.path.join(ROOT, "data")
def scene_obj():
    return open(os.path.join(data, "scene.obj"), "rb").read(), open(os.path.join(data, "scene.sq"), "rb").read(), open(os.path.join(data, "camera"), "rb").read()
cases = [("C1 scene.obj 256x256@4", scene_obj, 256, 256, 4),
         ("C2 scene.obj 1920x1080@256", scene_obj, 1920, 1080, 256),
         ("C3 blob6 (82k tris) 1920x1080@512", lambda: G.blob_scene(6), 1920, 1080, 512),
         ("C4 scene.obj 3840x2160@1024", scene_obj, 3840, 2160, 1024),
         ("C5 heightfield708 (1M tris) 1920x1080@256", lambda: G.heightfield_scene(708), 1920, 1080, 256)]
for name, make, w, h, n in cases:
    obj, sq, camt = make()
    t = time.time(); mesh = sqt.Mesh.from_text(obj, sq); t_load = time.time() - t
    dev = len(mesh) >= 50000
    t = time.time(); bih = sqt.BIH(mesh, device=0 if dev else None); t_build = time.time() - t
    cam = sqt.camera_from_text(camt)
    t = time.time(); ds = sqt.DeviceScene(bih, 0); torch.cuda.synchronize(); t_up = time.time() - t
    ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); t_r = time.time() - t
    rays = ds.stats(reset=True)[0] / 2
    print(f"{name}: {len(mesh)} tris, load {t_load*1e3:.0f} ms, BIH build ({'GPU' if dev else 'host'}) {t_build*1e3:.1f} ms, upload {t_up*1e3:.1f} ms, "
          f"render {t_r*1e3:.1f} ms = {w*h*n/t_r/1e6:.0f} Msamples/s ({rays/t_r/1e6:.0f} Mrays/s traced)", flush=True)
    ds.close()
