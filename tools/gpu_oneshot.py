"""Wall time of the one-shot drop-in call (sq_render_rgb8: scene upload + workspace + render + copy back),
the number a Haskell host would see, beside the resident-API frame time."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
torch.cuda.init()
for (w, h, n) in ((1920, 1080, 256), (1920, 1080, 16), (540, 540, 10)):
    ts = []
    for _ in range(4):
        t = time.time(); img = sqt.render_rgb8(bih, cam, n, (w, h)); ts.append(time.time() - t)
    print(f"one-shot {w}x{h}@{n}: " + ", ".join(f"{x*1e3:.1f}" for x in ts) + f" ms -> {w*h*n/min(ts)/1e6:.1f} Msamples/s", flush=True)
