"""Timing of the stand-in scenes for BASELINE configs[2]/[4] on one GPU (informational)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes as G
sqt = importlib.import_module("squigly-trace_amd")
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "both"
cases = []
if which in ("both", "blob6"): cases.append(("blob6 (81932 tris)", G.blob_scene(6), (1920, 1080, 16)))
if which in ("both", "hf708"): cases.append(("heightfield708 (1002540 tris)", G.heightfield_scene(708), (1920, 1080, 16)))
for name, (obj, sq, camt), (w, h, n) in cases:
    t = time.time(); bih = sqt.BIH(sqt.Mesh.from_text(obj, sq)); tb = time.time() - t
    cam = sqt.camera_from_text(camt)
    ds = sqt.DeviceScene(bih, 0); ds.enable_timing()
    sweep = [(kb, bpc) for kb in (64, 32, 16, 8, 4, 0) for bpc in (0,)] if "sweep" in sys.argv else [(None, None)]
    for kb, bpc in sweep:
        if kb is not None:
            ds.set_option("lds_node_kb", kb); ds.set_option("trace_blocks_per_cu", bpc)
        ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing(); ds.stats(reset=True)
        t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); dt = time.time() - t
        ms, cnt, kn = ds.kernel_timing(); rays = ds.stats()[0]
        print(f"{name} lds_node_kb={kb}: load+build {tb:.2f}s height {bih.height}; {w}x{h}@{n}: {w*h*n/dt/1e6:.1f} Msamples/s, {rays/dt/1e6:.1f} Mrays/s (trace {ms*cnt:.1f} of {dt*1e3:.1f} ms)", flush=True)
    ds.close()
