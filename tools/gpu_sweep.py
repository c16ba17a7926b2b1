"""Option sweep on one scene, one process (same box, same clocks):
    python tools/gpu_sweep.py scene=obj spp=256 refill_min=8,12,16 flush_min=0,40 [key=v1,v2 ...]
Every combination renders `frames` frames (default 3); the best time and the image checksum are printed."""
import importlib, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sqt = importlib.import_module("squigly-trace_amd")
import torch
kv = dict(a.split("=") for a in sys.argv[1:])
frames = int(kv.pop("frames", 3)); spp = int(kv.pop("spp", 256)); w = int(kv.pop("w", 1920)); h = int(kv.pop("h", 1080))
which = kv.pop("scene", "obj")
if which == "obj":
    data = os.path.join(ROOT, "data")
    obj, sq, camt = (open(os.path.join(data, f), "rb").read() for f in ("scene.obj", "scene.sq", "camera"))
else:
    import gen_scenes as G
    obj, sq, camt = G.blob_scene(int(which[4:])) if which.startswith("blob") else G.heightfield_scene(int(which[2:]))
mesh = sqt.Mesh.from_text(obj, sq)
bih = sqt.BIH(mesh, device=0 if len(mesh) >= 50000 else None)
cam = sqt.camera_from_text(camt)
ds = sqt.DeviceScene(bih, 0)
keys = list(kv)
ds.render_rows(cam, spp, w, h, want_avg=False); torch.cuda.synchronize()
for combo in itertools.product(*[kv[k].split(",") for k in keys]):
    for k, v in zip(keys, combo):
        ds.set_option(k, int(v))
    best = 1e9
    for i in range(frames):
        t = time.time(); _, r = ds.render_rows(cam, spp, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"{which} {dict(zip(keys, combo))}: {best*1e3:.2f} ms -> {w*h*spp/best/1e6:.0f} Msamples/s, checksum {int(r.to(torch.int64).sum())}", flush=True)
