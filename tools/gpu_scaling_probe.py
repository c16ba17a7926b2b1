"""Frame time against spp for the whole headline frame and for one rank's share of it at 2 / 4 / 8 ranks (interleaved
8-row blocks): separates the per-sample cost from the per-frame fixed cost that limits strong scaling."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
for k, v in (a.split("=") for a in sys.argv[1:]):
    ds.set_option(k, int(v))
w, h = 1920, 1080
for ranks in (1, 2, 4, 8):
    shard = (None, 0, 1) if ranks == 1 else (8, ranks // 2, ranks)
    res = []
    for spp in (16, 64, 256):
        ds.render_rows(cam, spp, w, h, want_avg=False, shard=shard); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.time(); ds.render_rows(cam, spp, w, h, want_avg=False, shard=shard); torch.cuda.synchronize(); best = min(best, time.time() - t)
        res.append((spp, best * 1e3))
    b = (res[2][1] - res[1][1]) / (256 - 64); a = res[2][1] - 256 * b
    print(f"ranks={ranks}: " + ", ".join(f"{s} spp {t:.2f} ms" for s, t in res) + f" -> {b*1e3:.1f} us per spp + {a:.2f} ms per frame; "
          f"speedup at 256 spp if every rank took this long: {ranks and 0}", flush=True)
