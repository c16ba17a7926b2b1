"""Lane-occupancy profile of the trace kernel (PROFILE build): where do lanes idle?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sqt = importlib.import_module("squigly-trace_amd")
import torch
data = os.path.join(ROOT, "data")
bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data)); cam = sqt.load_camera(os.path.join(data, "camera"))
ds = sqt.DeviceScene(bih, 0); ds.enable_timing()
w, h, n = 1920, 1080, 256
for res in (1, 0):
    for strag in (8,):
        ds.set_option("resident", res); ds.set_option("profile", 1); ds.set_option("straggler_lanes", strag)
        ds.render_rows(cam, n, w, h); torch.cuda.synchronize(); ds.stats(reset=True)
        ds.render_rows(cam, n, w, h); torch.cuda.synchronize()
        st = ds.stats(reset=True)
        rays, adv, unw, desc, leafit, tri, outer, refill, refl = st[:9]
        print(f"resident={res} straggler={strag}: rays={rays}")
        print(f"  per ray: descend steps {desc/rays:.2f}, unwind steps {unw/rays:.2f}, tri tests {tri/rays:.2f}, refills {refl/rays:.3f}")
        print(f"  advance loop: {adv} wave-iterations; lanes busy unwind {unw/(adv*64):.3f}, descend {desc/(adv*64):.3f}")
        print(f"  leaf loop: {leafit} wave-iterations; lanes busy {tri/(leafit*64):.3f}")
        print(f"  outer iterations {outer}, refill executions {refill} ({refill/outer:.3f} of outer), lanes refilled per execution {refl/max(refill,1):.2f}")
        ds.set_option("profile", 0)
