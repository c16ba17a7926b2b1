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
ds = sqt.DeviceScene(bih, 0); ds.enable_timing()
for (w, h, n, cast) in [(64, 64, 4, False), (48, 80, 3, False), (64, 64, 2, True), (96, 96, 16, False), (200, 120, 33, False)]:
    o_avg, o_rgb, _ = ob.render(oc, n, w, h, cast=cast, threads=16)
    for variant in (1, 2):
        ds.set_option("variant", variant)
        if variant == 2:
            ds.set_option("slots", w * h * 5)        # force several batches
        a, r = ds.render_rows(cam, n, w, h, cast=cast); torch.cuda.synchronize()
        g_avg, g_rgb = a.cpu().numpy(), r.cpu().numpy()
        same = np.array_equal(g_avg.view(np.uint32), o_avg.view(np.uint32)); same8 = np.array_equal(g_rgb, o_rgb)
        nd = int((g_avg.view(np.uint32) != o_avg.view(np.uint32)).any(-1).sum())
        print(f"{w}x{h} n={n} cast={cast} variant={variant}: avg bit-equal={same} rgb equal={same8} differing pixels={nd} maxabs={np.abs(g_avg-o_avg).max():.3g}", flush=True)
        ok &= same and same8
ds.set_option("slots", 512 << 20)
args = [a for a in sys.argv[1:]]
for variant in (1, 2):
    ds.set_option("variant", variant)
    for (w, h, n) in [(256, 256, 4), (540, 540, 64), (1920, 1080, 64)]:
        ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing()
        t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); dt = time.time() - t
        ms, cnt, name = ds.kernel_timing()
        print(f"variant {variant} {w}x{h}@{n}: wall {dt*1e3:.1f} ms -> {w*h*n/dt/1e6:.1f} Msamples/s   [{name}: {cnt} launches, {ms*cnt:.1f} ms total]", flush=True)
for res in (0, 1):
    ds.set_option("variant", 2); ds.set_option("resident", res)
    a, r = ds.render_rows(cam, 16, 96, 96); torch.cuda.synchronize()
    o_avg, o_rgb, _ = ob.render(oc, 16, 96, 96, threads=16)
    same = np.array_equal(a.cpu().numpy().view(np.uint32), o_avg.view(np.uint32))
    ok &= same
    for (w, h, n) in [(540, 540, 64), (1920, 1080, 64)]:
        ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing()
        t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); dt = time.time() - t
        ms, cnt, name = ds.kernel_timing()
        print(f"resident={res} parity={same} {w}x{h}@{n}: {w*h*n/dt/1e6:.1f} Msamples/s (trace {ms*cnt:.1f} ms of {dt*1e3:.1f})", flush=True)
if "sweep" in args:
    ds.set_option("variant", 2)
    w, h, n = 1920, 1080, 64
    for strag in (0, 2, 4, 6, 8, 12, 16):
        for bpc in (0,):
            ds.set_option("straggler_lanes", strag); ds.set_option("trace_blocks_per_cu", bpc)
            ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.reset_timing()
            t = time.time(); ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); dt = time.time() - t
            ms, cnt, name = ds.kernel_timing()
            print(f"straggler={strag} blocks/cu={bpc}: {w*h*n/dt/1e6:.1f} Msamples/s (trace {ms*cnt:.1f} ms of {dt*1e3:.1f})", flush=True)
sys.exit(0 if ok else 1)
