"""Command line mirroring the reference's `squigly-trace` executable (app/Main.hs:13-75).

    python -m squigly-trace_amd.cli --samples 100 --dimensions 540,540 --savepath render/result.png

Flags and defaults are the reference's (cmdargs long names; `-s -d -p -c` short names as declared in
app/Main.hs:15-23).  The material file named by `mtllib` is read from ./data/, as in src/Obj.hs:52.
"""
import argparse
import os
import sys
import time

from . import BIH, Mesh, Settings, lib, load_camera, render


def parse_dimensions(text):
    try:
        w, h = (int(v) for v in text.strip("()").split(","))
    except ValueError:
        raise argparse.ArgumentTypeError("expected W,H")
    return (w, h)


def build_parser():
    p = argparse.ArgumentParser(prog="squigly-trace", description="A cute raytracer",
                                epilog="squigly-trace was made by Ruko (https://github.com/rukokarasu/)")
    p.add_argument("-s", "--samples", type=int, default=10, help="How many samples per pixel to trace")
    p.add_argument("-d", "--dimensions", type=parse_dimensions, default=(540, 540),
                   help="Dimensions of the resulting image")
    p.add_argument("-p", "--savepath", default="./render/result.png", help="Where to save the output")
    p.add_argument("--objpath", default="./data/scene.obj", help="File to load .obj from")
    p.add_argument("-c", "--camerapath", default="./data/camera", help="File to load camera data from")
    p.add_argument("--debug", action="store_true", help="Run in debug mode")
    p.add_argument("--debugpath", default="", help="File to write debug info to")
    p.add_argument("--cast", action="store_true", help="Raycast instead of raytracing (i.e. don't bounce rays)")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    settings = Settings(samples=a.samples, dimensions=a.dimensions, savePath=a.savepath, objPath=a.objpath,
                        cameraPath=a.camerapath, debug=a.debug, debugPath=a.debugpath, cast=a.cast)
    cam = load_camera(settings.cameraPath)                      # app/Main.hs:38
    mesh = Mesh.from_obj(settings.objPath, "./data")            # loadTris, app/Main.hs:58-61 + src/Obj.hs:52
    if settings.debug:                                          # src/Obj.hs:55-57: print (head objs); print mats
        first, mats = mesh.debug_show()
        if not first:                                           # `head objs` of a file without objects throws in the reference
            print("squigly-trace: Prelude.head: empty list", file=sys.stderr)
            return 1
        print(first)
        print(mats)
    # loadBIH, app/Main.hs:63-75.  Both builds give the same arrays; the GPU one wins from a few 10^4 triangles up.
    bih = BIH(mesh, device=0 if (len(mesh) >= 50000 and lib().sq_device_count() > 0) else None)
    if settings.debug:
        if settings.debugPath:
            nodes = bih.nodes
            with open(settings.debugPath, "w") as f:            # flattened dump (the Haskell `show bih` text is not reproduced)
                f.write(f"BIH bounds={bih.bounds.tolist()} nodes={len(nodes)} tris={len(bih.tris)}\n")
                for i, nd in enumerate(nodes):
                    f.write(f"{i} kind={int(nd['kind']) & 3} count={int(nd['kind']) >> 2} lmax={nd['lmax']!r} "
                            f"rmin={nd['rmin']!r} link={int(nd['link'])}\n")
            print(f"Wrote BIH to {settings.debugPath}")
        print(f"BIH height is {bih.height}")
        print(f"Length of longest leaf is {bih.longest_leaf}")
        print(f"Number of leaves is {bih.num_leaves}")
    print("Rendering scene...")
    t0 = time.time()
    print("Started at " + time.strftime("%H:%M:%S%p UTC", time.gmtime(t0)).lower().replace("utc", "UTC"))
    os.makedirs(os.path.dirname(os.path.abspath(settings.savePath)), exist_ok=True)
    render(bih, cam, settings)
    t1 = time.time()
    print("Finished at " + time.strftime("%H:%M:%S%p UTC", time.gmtime(t1)).lower().replace("utc", "UTC"))
    print(f"Took {t1 - t0:.6f}s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
