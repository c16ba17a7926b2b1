"""Option sweep for the trace kernel (development aid): parity on a small frame, timing on 1080p."""
import importlib, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, pyoracle as O
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ob = O.BIH(O.tris_from_obj(os.path.join(data, "scene.obj"), data)); oc = O.load_camera(os.path.join(data, "camera"))
o_avg, _, _ = ob.render(oc, 16, 96, 96, threads=16)
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, 64
def run(opts):
    for k, v in opts.items(): ds.set_option(k, v)
    a, _ = ds.render_rows(cam, 16, 96, 96); torch.cuda.synchronize()
    same = np.array_equal(a.cpu().numpy().view(np.uint32), o_avg.view(np.uint32))
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing()
    best = 1e9
    for _ in range(2):
        t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); best = min(best, time.time() - t)
    ms, cnt, name = ds.kernel_timing()
    print(f"{opts}: parity={same} {w*h*n/best/1e6:.1f} Msamples/s (trace {ms*cnt/2:.1f} ms of {best*1e3:.1f})", flush=True)
    return same
ok = True
spec = sys.argv[1] if len(sys.argv) > 1 else "default"
if spec == "default":
    for res in (1, 0):
        for st in (4, 8):
            ok &= run({"resident": res, "straggler_lanes": st})
sys.exit(0 if ok else 1)
