#!/usr/bin/env python3
"""Round profile collection on the GPU box (run through gpurun): rocprofv3 kernel-trace stats of the default bench
command, then PMC counters in SEPARATE passes (never combined with tracing), for the headline workload and for the
C3 / C5 stand-in scenes.  Writes gpurun_out/profiles_<tag>/ : kernel_stats.csv, pmc_headline.txt, latest_pmc.json,
pmc_c3.txt, pmc_c5.txt, latest_other_configs.json -- copy them into profiles/ (tracked) afterwards.

    python tools/collect_profiles.py <tag> [headline] [c3] [c5]
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
what = set(sys.argv[2:]) or {"headline", "c3", "c5"}
OUT = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

SQ_A = "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ_B = "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
TCC = "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"


def pmc(name, counters, cmd, regex="sq_"):
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    full = ["timeout", "-k", "10", "400", "rocprofv3", "--pmc"] + counters.split() + \
           ["--kernel-include-regex", regex, "--output-format", "csv", "-d", d, "--"] + cmd
    rc = subprocess.call(full, cwd=ROOT, env=env, stdout=open(d + ".log", "w"), stderr=subprocess.STDOUT)
    print(f"pmc pass {name}: rc={rc}", flush=True)
    tot = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot.setdefault((kn, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return tot


def write_summary(path, header, tots):
    with open(path, "w") as o:
        o.write(header + "\n")
        for tot in tots:
            for (kn, k), v in tot.items():
                o.write(f"{kn:62s} {k:26s} launches={len(v):3d} sum={sum(v):.6g} mean_per_launch={sum(v)/len(v):.6g}\n")


def trace_sum(tots, counter):
    for tot in tots:
        for (kn, k), v in tot.items():
            if kn.startswith("sq_trace_rays") and k == counter:
                return sum(v), len(v)
    return None, 0


if "headline" in what:
    bench = ["python", "bench.py", "--no-other"]
    d = os.path.join(OUT, "trace")
    shutil.rmtree(d, ignore_errors=True)
    rc = subprocess.call(["timeout", "-k", "10", "500", "rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] +
                         bench + ["--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                         stdout=open(os.path.join(OUT, "bench_under_trace.log"), "w"), stderr=subprocess.STDOUT)
    print(f"kernel-trace: rc={rc}", flush=True)
    for f in glob.glob(os.path.join(d, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(OUT, "kernel_stats.csv"))
    one = bench + ["--no-cpu", "--no-oneshot", "--steps", "1", "--warmup", "0"]
    tots = [pmc("h_sqa", SQ_A, one), pmc("h_sqb", SQ_B, one), pmc("h_tcc", TCC, one), pmc("h_fetch", "FETCH_SIZE", one), pmc("h_write", "WRITE_SIZE", one)]
    write_summary(os.path.join(OUT, "pmc_headline.txt"),
                  "# rocprofv3 --pmc <one group per pass> -- python bench.py --no-other --no-cpu --no-oneshot --steps 1 --warmup 0 ; every sq_ kernel of ONE headline frame", tots)
    res = {"command": "python bench.py --no-other --no-cpu --no-oneshot --steps 1 --warmup 0", "workload": [1920, 1080, 256],
           "what": "sq_trace_rays, sums over the launches of ONE frame (1920x1080 @ 256 spp); one rocprofv3 --pmc pass per counter group"}
    for c in SQ_A.split() + SQ_B.split():
        v, n = trace_sum(tots, c)
        if v is not None:
            res[c] = v; res["launches"] = n
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v, n = trace_sum(tots, c)
        if v is not None:
            res[c + "_KB"] = v
    json.dump(res, open(os.path.join(OUT, "latest_pmc.json"), "w"), indent=1)
    print(json.dumps(res)[:600], flush=True)

other = {}
opath = os.path.join(OUT, "latest_other_configs.json")
if os.path.exists(opath):
    other = json.load(open(opath))
for key, scene, spp in (("c3", "blob6", 512), ("c5", "hf708", 256)):
    if key not in what:
        continue
    cmd = ["python", "tools/gpu_frames.py", f"scene={scene}", f"spp={spp}", "frames=1"]
    tots = [pmc(f"{key}_fetch", "FETCH_SIZE", cmd, "sq_trace"), pmc(f"{key}_write", "WRITE_SIZE", cmd, "sq_trace"),
            pmc(f"{key}_sq", "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS", cmd, "sq_trace"),
            pmc(f"{key}_tcc", TCC, cmd, "sq_trace")]
    write_summary(os.path.join(OUT, f"pmc_{key}.txt"),
                  f"# rocprofv3 --pmc <one group per pass> -- python tools/gpu_frames.py scene={scene} spp={spp} frames=1 ; sq_trace_rays launches of ONE 1920x1080 frame", tots)
    f, _ = trace_sum(tots, "FETCH_SIZE"); wv, _ = trace_sum(tots, "WRITE_SIZE"); rd, _ = trace_sum(tots, "SQ_INSTS_VMEM_RD")
    hit, _ = trace_sum(tots, "TCC_HIT_sum"); miss, _ = trace_sum(tots, "TCC_MISS_sum")
    wc, _ = trace_sum(tots, "SQ_WAVE_CYCLES"); wa, _ = trace_sum(tots, "SQ_WAIT_ANY")
    if f is not None and wv is not None:
        other[key] = {"scene": scene, "spp": spp, "hbm_bytes_per_frame": (f * 2 + wv) * 1024, "vmem_rd_per_frame": rd,
                      "l2_hit_rate": round(hit / (hit + miss), 4) if hit is not None and miss is not None and hit + miss > 0 else None,
                      "wait_fraction_of_wave_cycles": round(wa / wc, 3) if wa and wc else None,
                      **{c: trace_sum(tots, c)[0] for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES")},
                      "note": "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KB -> bytes; separate --pmc passes"}
    json.dump(other, open(opath, "w"), indent=1)
    print(key, json.dumps(other.get(key)), flush=True)
