"""Pooled trace kernel with and without leaf culling: frame time and the PROFILE build's per-ray / per-section counters."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0)
w, h, n = 1920, 1080, int(os.environ.get("SPP", "64"))
names = ["store+refill", "return step", "branch step", "leaf open+scan", "owner lookup+pulls", "fetch+MT", "hit fold", "tail"]
for cull in (0, 1):
    ds.set_option("cull", cull); ds.set_option("profile", 0); ds.enable_timing()
    ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); ds.reset_timing()
    t = time.time(); ds.render_rows(cam, n, w, h, want_avg=False); torch.cuda.synchronize(); dt = time.time() - t
    ms, launches, _ = ds.kernel_timing()
    print(f"cull={cull}: frame {dt*1e3:.1f} ms, trace launches {launches} x {ms:.2f} ms", flush=True)
    ds.set_option("profile", 1)
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.stats(reset=True)
    ds.render_rows(cam, n, w, h); torch.cuda.synchronize()
    st = ds.stats(reset=True)
    ds.set_option("profile", 0)
    rays, iters, unw, desc, wins, pairs, hits, refill, refl = st[:9]
    print(f"  per ray: lane-iterations {iters*64/rays:.2f}, returns {unw/rays:.2f}, branch steps {desc/rays:.2f}, triangle tests {pairs/rays:.2f}, hits folded {hits/rays:.3f}")
    print(f"  per iteration: lanes returning {unw/(iters*64):.3f}, lanes branching {desc/(iters*64):.3f}, windows {wins/iters:.3f}, window fill {pairs/max(wins*128,1):.3f}")
    comb, rl, rw, sl, sw, hs = st[9:15]
    print(f"  returns: COMBINE pops {comb/rays:.3f} per ray; t recomputed {rl/rays:.3f} per ray, in {rw/iters:.3f} of the iterations; "
          f"distance compare {sl/rays:.3f} per ray, in {sw/iters:.3f} of the iterations; slow compares in the hit fold {hs/rays:.4f} per ray")
    ts = st[16:24]; tot = sum(ts)
    print("  wave cycles per iteration: " + ", ".join(f"{nm} {t/iters:.0f} ({100*t/tot:.0f}%)" for nm, t in zip(names, ts)) + f"; total {tot/iters:.0f}", flush=True)
