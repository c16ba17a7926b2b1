"""Headline frame: the pooled trace kernel (option pool=1) against the one-ray-per-lane leaf loop (pool=0), with its
two tunables, plus the lane-occupancy counters of the PROFILE build.  Every variant must produce the same image."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, int(os.environ.get("SPP", "256"))
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def frame(opts, reps=3):
    for k, v in opts.items():
        ds.set_option(k, v)
    _, r = ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
    return r, best


ref, t0 = frame({"pool": 0})
print(f"pool=0: {t0*1e3:.1f} ms -> {w*h*n/t0/1e6:.0f} Msamples/s", flush=True)
grid = [(8, 0)] if quick else [(8, 0), (4, 0), (12, 0), (16, 0), (24, 0), (8, 24), (8, 40), (8, 56), (16, 40), (1, 0)]
for refill_min, flush_min in grid:
    r, t = frame({"pool": 1, "refill_min": refill_min, "flush_min": flush_min})
    print(f"pool=1 refill_min={refill_min} flush_min={flush_min}: {t*1e3:.1f} ms -> {w*h*n/t/1e6:.0f} Msamples/s "
          f"same_image={bool((r == ref).all())}", flush=True)

for opts in ({"pool": 1, "refill_min": 8, "flush_min": 0}, {"pool": 1, "refill_min": 8, "flush_min": 40}):
    ds.set_option("profile", 1)
    for k, v in opts.items():
        ds.set_option(k, v)
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.stats(reset=True)
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize()
    st = ds.stats(reset=True)
    ds.set_option("profile", 0)
    rays, iters, unw, desc, wins, pairs, hits, refill, refl = st[:9]
    print(f"profile {opts}: rays={rays}")
    print(f"  per ray: iterations {iters*64/rays:.2f} (lane-iterations), returns {unw/rays:.2f}, branch steps {desc/rays:.2f}, "
          f"pairs {pairs/rays:.2f}, hits folded {hits/rays:.3f}")
    print(f"  per iteration: lanes returning {unw/(iters*64):.3f}, lanes branching {desc/(iters*64):.3f}, windows {wins/iters:.3f}, "
          f"window fill {pairs/max(wins*64,1):.3f}, hits per window {hits/max(wins,1):.3f}")
    print(f"  refill executions {refill} ({refill/iters:.4f} per iteration), lanes per refill {refl/max(refill,1):.2f}")
    comb, rl, rw, sl, sw, hs = st[9:15]
    print(f"  returns: COMBINE pops {comb/rays:.3f} per ray; t recomputed {rl/rays:.3f} per ray, in {rw/iters:.3f} of the iterations; "
          f"distance compare {sl/rays:.3f} per ray, in {sw/iters:.3f} of the iterations; slow compares in the hit fold {hs/rays:.4f} per ray")
    ts = st[16:24]; tot = sum(ts)
    names = ["store+refill", "return step", "branch step", "leaf open+scan", "owner lookup+pulls", "fetch+MT", "hit fold", "tail"]
    print("  wave cycles per iteration: " + ", ".join(f"{nm} {t/iters:.0f} ({100*t/tot:.0f}%)" for nm, t in zip(names, ts)) + f"; total {tot/iters:.0f}", flush=True)
