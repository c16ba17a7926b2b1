"""Quick GPU-vs-oracle check + timing (development aid; run via gpurun)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as O
sqt = importlib.import_module("squigly-trace_amd")
import torch

data = os.path.join(ROOT, "data")
mesh = sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)
bih = sqt.BIH(mesh)
cam = sqt.load_camera(os.path.join(data, "camera"))
ob = O.BIH(O.tris_from_obj(os.path.join(data, "scene.obj"), data))
oc = O.load_camera(os.path.join(data, "camera"))
ok = True
for (w, h, n, cast) in [(64, 64, 4, False), (48, 80, 3, False), (64, 64, 2, True), (96, 96, 16, False)]:
    g_avg = sqt.render_f32(bih, cam, n, (w, h), cast)
    g_rgb = sqt.render_rgb8(bih, cam, n, (w, h), cast)
    o_avg, o_rgb, _ = ob.render(oc, n, w, h, cast=cast, threads=os.cpu_count())
    same = np.array_equal(g_avg.view(np.uint32), o_avg.view(np.uint32))
    same8 = np.array_equal(g_rgb, o_rgb)
    nd = int((g_avg.view(np.uint32) != o_avg.view(np.uint32)).any(-1).sum())
    print(f"{w}x{h} n={n} cast={cast}: avg bit-equal={same} rgb equal={same8} differing pixels={nd} maxabs={np.abs(g_avg-o_avg).max():.3g}", flush=True)
    ok &= same and same8
ds = sqt.DeviceScene(bih, 0)
for (w, h, n) in [(256, 256, 4), (540, 540, 16), (1920, 1080, 16)]:
    torch.cuda.synchronize(); ds.reset_timing()
    t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); dt = time.time() - t
    ms, cnt, name = ds.kernel_timing()
    print(f"{w}x{h}@{n}: wall {dt*1e3:.1f} ms, kernel {ms:.1f} ms, {w*h*n/ms/1e3:.1f} Msamples/s", flush=True)
sys.exit(0 if ok else 1)
