"""Render a few frames with given options, for timing and profiling runs:
    python tools/gpu_frames.py scene=obj|blob6|hf708 pool=1 frames=2 spp=256 [w=1920 h=1080 key=value ...]
Unknown keys are passed to sq_set_option."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sqt = importlib.import_module("squigly-trace_amd")
import torch
kv = dict(a.split("=") for a in sys.argv[1:])
frames = int(kv.pop("frames", 2)); spp = int(kv.pop("spp", 256)); w = int(kv.pop("w", 1920)); h = int(kv.pop("h", 1080))
which = kv.pop("scene", "obj")
shard = tuple(int(x) for x in kv.pop("shard", "0,0,1").split(","))      # row_block,shard,n_shards (row_block 0 = whole frame)
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
ds = sqt.DeviceScene(bih, 0)
for k, v in kv.items():
    ds.set_option(k, int(v))
ref = None
for i in range(frames):
    t = time.time(); _, r = ds.render_rows(cam, spp, w, h, want_avg=False, shard=shard); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"{which} {kv} frame {i}: {dt*1e3:.1f} ms -> {w*h*spp/dt/1e6:.0f} Msamples/s, image checksum {int(r.to(torch.int64).sum())}", flush=True)
