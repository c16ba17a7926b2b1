"""Frame time at the headline configuration: serial schedule vs the overlapped schedules, and -- VERDICT round 2, item 6 --
whether the waves of the per-sample kernels (sq_gen_bounce1, sq_shade1) actually run BESIDE the resident trace workgroups
(option "coresidency": the trace kernel keeps a gauge of its live workgroups; a per-sample wave that starts or ends while all
but eight CUs hold one shares its CU with one).  Checks that every schedule gives the same image.

    python tools/gpu_overlap.py [quick]
"""
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
print(f"build {sqt.build_id()}", flush=True)
combos = ((0, 0), (1, 4), (2, 0), (2, 4), (1, 2), (1, 8), (2, 2), (0, 0)) if "quick" not in sys.argv else ((0, 0), (1, 4), (2, 0), (0, 0))
for overlap, aux in combos:
    ds.set_option("overlap", overlap); ds.set_option("aux_blocks_per_cu", aux); ds.set_option("coresidency", 0)
    a, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    if ref is None: ref = r.clone()
    same = bool((r == ref).all())
    best = 1e9
    for _ in range(3):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    # one more frame with the diagnostic on (its atomics are not in the timed frames)
    ds.set_option("coresidency", 1); ds.stats(reset=True)
    ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    st = ds.stats(reset=True)
    gauge, waves, sb, eb = st[24], st[25], st[26], st[27]
    print(f"overlap={overlap} aux_blocks_per_cu={aux}: {best*1e3:.2f} ms -> {w*h*n/best/1e6:.1f} Msamples/s same_image={same} | "
          f"per-sample waves {waves}: started beside a full set of trace workgroups {sb} ({100.0*sb/max(waves,1):.1f} %), "
          f"ended beside {eb} ({100.0*eb/max(waves,1):.1f} %), gauge at rest {gauge}", flush=True)
