"""Same-process A/B of sq_set_option settings on the headline frame (or any scene / size):
    python tools/gpu_ab_options.py [scene=obj] [spp=256] [w=1920 h=1080] [reps=5] [rounds=2] [shard=rb,i,n] -- incremental=0 incremental=1 "incremental=1,descend_extra=3"
Every setting is a comma-separated list of key=value (reset to the FIRST setting's keys between settings: name every key
you vary in every setting).  Settings alternate round by round, so clock and box drift hits them alike; prints best and mean
frame time, the dominant kernel's time per frame, and whether the image equals the first setting's bit for bit."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sqt = importlib.import_module("squigly-trace_amd")
import torch
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
kv = dict(a.split("=") for a in args[:cut]); settings = args[cut + 1:] or [""]
spp = int(kv.get("spp", 256)); w = int(kv.get("w", 1920)); h = int(kv.get("h", 1080)); reps = int(kv.get("reps", 5)); rounds = int(kv.get("rounds", 2))
which = kv.get("scene", "obj")
shard = tuple(int(x) for x in kv.get("shard", "0,0,1").split(","))
shard = (None, 0, 1) if shard[0] == 0 else shard
if which == "obj":
    data = os.path.join(ROOT, "data")
    obj, sq, camt = (open(os.path.join(data, f), "rb").read() for f in ("scene.obj", "scene.sq", "camera"))
else:
    import gen_scenes as G
    obj, sq, camt = G.blob_scene(int(which[4:])) if which.startswith("blob") else G.heightfield_scene(int(which[2:]))
mesh = sqt.Mesh.from_text(obj, sq)
bih = sqt.BIH(mesh, device=0 if len(mesh) >= 50000 else None)
cam = sqt.camera_from_text(camt)
ds = sqt.DeviceScene(bih, 0); ds.enable_timing()
print("build", sqt.build_id(), "scene", which, f"{w}x{h}@{spp}", "shard", shard, flush=True)
ref = None
res = {s: [] for s in settings}; ker = {s: [] for s in settings}; same = {s: True for s in settings}
for rnd in range(rounds):
    for s in settings:
        for item in filter(None, s.split(",")):
            k, v = item.split("="); ds.set_option(k, int(v))
        _, r = ds.render_rows(cam, spp, w, h, want_avg=False, shard=shard); torch.cuda.synchronize()
        if ref is None: ref = r.clone()
        same[s] = same[s] and bool((r == ref).all())
        for _ in range(reps):
            ds.reset_timing()
            t = time.time(); ds.render_rows(cam, spp, w, h, want_avg=False, shard=shard); torch.cuda.synchronize(); res[s].append(time.time() - t)
            ms, cnt, name = ds.kernel_timing(); ker[s].append(ms * cnt)
for s in settings:
    a = res[s]; k = ker[s]
    print(f"{s or '(defaults)':48s} best {min(a)*1e3:7.2f} ms  mean {sum(a)/len(a)*1e3:7.2f} ms  | trace kernel best {min(k):6.2f} mean {sum(k)/len(k):6.2f} ms per frame | same image as first: {same[s]}", flush=True)
