import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, pyoracle as O
sqt = importlib.import_module("squigly-trace_amd")
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ob = O.BIH(O.tris_from_obj(os.path.join(data, "scene.obj"), data)); oc = O.load_camera(os.path.join(data, "camera"))
for n in (1, 2, 3):
    g = sqt.render_f32(bih, cam, n, (64, 64), True); o, _, _ = ob.render(oc, n, 64, 64, cast=True, threads=8)
    d = (g.view(np.uint32) != o.view(np.uint32)).any(-1)
    print(n, int(d.sum()))
    ys, xs = np.nonzero(d)
    for y, x in list(zip(ys, xs))[:4]:
        print("  ", y, x, g[y, x], o[y, x], [v.hex() for v in g[y, x].astype(float)], [v.hex() for v in o[y, x].astype(float)])
np.save(os.path.join(ROOT, "gpurun_out", "cast_gpu_n1.npy"), sqt.render_f32(bih, cam, 1, (64, 64), True))
