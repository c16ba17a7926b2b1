"""Host vs GPU BIH build time on the stand-in scenes (informational; the trees are compared in
tests/test_gpu_bih_build.py)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes as G
sqt = importlib.import_module("squigly-trace_amd")
import torch
torch.cuda.init()
cases = [("scene.obj", None), ("blob6", G.blob_scene(6)), ("heightfield708", G.heightfield_scene(708))]
if "only1m" in sys.argv: cases = cases[2:]
if "big" in sys.argv: cases.append(("heightfield2000", G.heightfield_scene(2000)))
for name, sc in cases:
    t = time.time()
    mesh = sqt.Mesh.from_obj(os.path.join(ROOT, "data/scene.obj"), os.path.join(ROOT, "data")) if sc is None else sqt.Mesh.from_text(sc[0], sc[1])
    tl = time.time() - t
    t = time.time(); host = sqt.BIH(mesh); th = time.time() - t
    sqt.BIH(mesh, device=0)                                  # first call pays module load / allocation warm-up
    best = 1e9
    for _ in range(3):
        t = time.time(); dev = sqt.BIH(mesh, device=0); best = min(best, time.time() - t)
    t = time.time(); ds = sqt.DeviceScene(dev, 0); torch.cuda.synchronize(); tu = time.time() - t
    ds.close()
    print(f"{name}: {len(mesh)} tris, load {tl*1e3:.0f} ms, host build {th*1e3:.1f} ms, device build {best*1e3:.1f} ms, "
          f"scene upload {tu*1e3:.1f} ms, height {dev.height}, nodes {dev.scene.n_nodes}", flush=True)
