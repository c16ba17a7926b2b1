"""One rank's share of the headline frame at 8 ranks, for several row-block sizes of the interleaved partition: min / mean / max over the
eight shards (the strong-scaling frame time is the MAX), best of 3 each.  python tools/gpu_share_balance.py [key=value options]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd"); import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); w, h, n = 1920, 1080, 256
for kv in sys.argv[1:]:
    k, v = kv.split("="); ds.set_option(k, int(v))
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
whole = timed(lambda: ds.render_rows(cam, n, w, h, want_avg=False))
print(f"build {sqt.build_id()} options {sys.argv[1:]}: whole frame {whole:.2f} ms", flush=True)
for N in (8, 4, 2):
    for rb in (1, 2, 4, 8, 16, 32):
        ms = [timed(lambda r=r: ds.render_rows(cam, n, w, h, want_avg=False, shard=(rb, r, N))) for r in range(N)]
        print(f"N={N} row_block={rb:2d}: shards min {min(ms):.2f} mean {sum(ms)/N:.2f} max {max(ms):.2f} ms -> {whole/max(ms):.2f}x before gather", flush=True)
