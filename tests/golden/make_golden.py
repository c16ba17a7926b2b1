#!/usr/bin/env python3
"""Generate tests/golden/* from the CPU oracle (oracle/libsq_oracle.so).

The reference cannot be run here (Haskell, no GHC in the image), so these vectors are outputs of
the oracle restatement; they pin the GPU path and guard the oracle against regressions.
example_png_patches.json is derived from the reference's own rendered artefact
(/root/reference/render/example.png) when that file is present.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")


def main():
    os.makedirs(G, exist_ok=True)
    tris = O.tris_from_obj(os.path.join(DATA, "scene.obj"), DATA)
    bih = O.BIH(tris)
    cam = O.load_camera(os.path.join(DATA, "camera"))
    thr = os.cpu_count() or 1
    # (1) fp32 avg framebuffer 64x64 @ 4 spp + rgb8
    avg, rgb, cnt = bih.render(cam, 4, 64, 64, threads=thr)
    np.save(os.path.join(G, "scene_64x64_4spp_avg.npy"), avg)
    np.save(os.path.join(G, "scene_64x64_4spp_rgb8.npy"), rgb)
    # non-square (A.1 quirk): 40 rows x 72 columns, 3 spp
    avg2, rgb2, _ = bih.render(cam, 3, 40, 72, threads=thr)
    np.save(os.path.join(G, "scene_40x72_3spp_avg.npy"), avg2)
    # (2) --cast 64x64
    cavg, crgb, _ = bih.render(cam, 2, 64, 64, cast=True, threads=thr)
    np.save(os.path.join(G, "scene_64x64_cast_rgb8.npy"), crgb)
    np.save(os.path.join(G, "scene_64x64_cast_avg.npy"), cavg)
    # per-sample radiance for 16 pixels x 4 samples (64x64 @ 4)
    px = [(y, x) for y in (20, 30, 40, 50) for x in (16, 28, 36, 48)]
    rad = np.array([[bih.sample_radiance(cam, 4, 64, 64, y, x, k) for k in range(4)] for (y, x) in px], np.float32)
    np.save(os.path.join(G, "scene_64x64_4spp_samples.npy"), rad)
    # (3) BIH stats + hashes of the flattened arrays
    kind, a, b, c = bih.preorder()
    flat = bih.flatten()
    stats = {
        "n_tris": int(bih.n_tris), "n_nodes": int(bih.n_nodes), "height": int(bih.height),
        "num_leaves": int(bih.num_leaves), "longest_leaf": int(bih.longest_leaf),
        "bounds": [float(v) for v in bih.bounds()],
        "sha256_flatten_vertices": hashlib.sha256(np.ascontiguousarray(
            np.stack([flat["a"], flat["b"], flat["c"]], 1)).tobytes()).hexdigest(),
        "sha256_preorder": hashlib.sha256(kind.tobytes() + a[kind != 3].tobytes() + b[kind != 3].tobytes()
                                          + c.tobytes()).hexdigest(),
        "sample_pixels": px,
        "counters_64x64_4spp": cnt,
    }
    # (4) Threefish KATs (public Skein 1.3 vectors) and TFGen words
    key2 = [int.from_bytes(bytes(range(0x10 + 8 * i, 0x18 + 8 * i)), "little") for i in range(4)]
    tw2 = [int.from_bytes(bytes(range(8 * i, 8 * i + 8)), "little") for i in range(2)]
    pt2 = [int.from_bytes(bytes(range(0xFF - 8 * i, 0xFF - 8 * i - 8, -1)), "little") for i in range(4)]
    stats["threefish_kat"] = [
        {"key": [0] * 4, "tweak": [0] * 2, "pt": [0] * 4,
         "ct_hex": ["94EEEA8B1F2ADA84", "ADF103313EAE6670", "952419A1F4B16D53", "D83F13E63C9F6B11"]},
        {"key": key2, "tweak": tw2, "pt": pt2,
         "ct_hex": ["DF8FEA0EFF91D0E0", "D50AD82EE69281C9", "76F48D58085D869D", "DF975E95B5567065"]},
    ]
    stats["tfgen_words"] = {str(s): O.tfgen_words(s) for s in (0, 1, 2, 2 ** 32 + 5, 8493465599)}
    with open(os.path.join(G, "scene_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)
    # statistical pin against the reference's own artefact
    ex = "/root/reference/render/example.png"
    if os.path.exists(ex):
        from PIL import Image
        img = np.array(Image.open(ex).convert("RGB")).astype(np.float64)
        mx, mn = img.max(-1), img.min(-1)
        light = np.tan(mx / 255.0 * np.pi / 2)
        ratio = np.where(mx > 0, mn / np.maximum(mx, 1), 0)
        rad_ex = img / np.maximum(mx, 1)[..., None] * (2 * light / (1 + ratio))[..., None]   # inverse of Lib.hs:93-104
        patches = {"back": (180, 250), "left": (70, 130), "right": (410, 470), "box": (300, 360)}
        out = {"rows": [250, 256], "size": [540, 540], "black_fraction": float((img.sum(-1) == 0).mean()),
               "black_fraction_9x9_cells": [[float((img[r * 60:(r + 1) * 60, c * 60:(c + 1) * 60].sum(-1) == 0).mean())
                                             for c in range(9)] for r in range(9)],
               "patches": {k: {"cols": list(v), "radiance": rad_ex[250:256, v[0]:v[1]].mean((0, 1)).tolist()}
                           for k, v in patches.items()}}
        with open(os.path.join(G, "example_png_patches.json"), "w") as f:
            json.dump(out, f, indent=1)
    print("golden fixtures written to", G)


if __name__ == "__main__":
    main()
