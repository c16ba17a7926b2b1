"""Frame time at the headline configuration: serial schedule vs the overlapped two-track schedule, and the
number of workgroups per CU given to the per-sample kernels; checks that both give the same image."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, 256
ref = None
for overlap, aux, slots_m in ((0, 0, 512), (1, 4, 512), (2, 0, 512), (2, 4, 512), (1, 2, 512), (1, 8, 512), (2, 2, 512), (0, 0, 512)):
    ds.set_option("overlap", overlap); ds.set_option("aux_blocks_per_cu", aux); ds.set_option("slots", slots_m << 20)
    a, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    if ref is None: ref = r.clone()
    same = bool((r == ref).all())
    best = 1e9
    for _ in range(3):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"overlap={overlap} aux_blocks_per_cu={aux} slots={slots_m}Mi: {best*1e3:.1f} ms -> {w*h*n/best/1e6:.1f} Msamples/s same_image={same}", flush=True)
