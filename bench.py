#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on data/scene.obj, 1920x1080 @ 256 spp (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--config c2|c4] [--scaling strong|weak]

With N > 1 and no WORLD_SIZE in the environment this process starts N rank processes itself (a
`python -m torch.distributed.run` child, before anything here touches a GPU) and relays rank 0's JSON line;
under torchrun (WORLD_SIZE set) it IS one of the ranks.  It exits non-zero if fewer than N devices are visible.

A step is one pass of the hot path over one frame: every pixel's `samples` paths are traced, accumulated in
sample order and tonemapped (src/Lib.hs:68-137).  The scene is already resident in HBM when the timed region
starts.  With N ranks the SAME frame (strong scaling, the default: 1080p @ 256 spp in total) is cut into
interleaved blocks of 2 rows, each rank renders its rows on its GPU, and one RCCL all_gather_into_tensor
reassembles the RGB8 framebuffer inside the timed region.  `--scaling weak` renders the frame at 256*N spp instead
(per-GPU work fixed).  `--config c4` is BASELINE.json configs[3]: 3840x2160 @ 1024 spp.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMD-32, max clock (same guide)
VALU_PEAK_TLANEOPS = N_SIMD * 32 * CLOCK_HZ / 1e12      # 78.6: one wave64 VALU instruction per 2 cycles per SIMD (= 157.3 TFLOP/s of FMAs)
CONFIGS = {"c2": (1920, 1080, 256), "c4": (3840, 2160, 1024)}   # first number = image ROWS (src/Lib.hs:70-71)


# What one VALU wave-instruction of each class costs a SIMD, in cycles, with four waves resident (tools/ubench/op_rate.hip on an
# MI355X): only fp32 add / mul / select / move issue at the 2-cycle class that `roofline.peak` is priced on.
# SQ_INSTS_VALU_<class> counters give the classes below; the rest of SQ_INSTS_VALU ("other": compares, min/max, selects, moves,
# DPP, lane reads) has no counter of its own, so it is priced between its cheapest member (v_cndmask / v_mov, 2.5) and its most
# expensive ones (v_cmp, v_min/max, DPP, v_readlane: 4.3), with the mean of the shipped kernel's static mix as the point value.
# Measured again in round 3 (profiles/r03a_op_rate.txt): v_add/sub 2.50, v_mul 2.5-2.7, v_mov 2.25, v_xor 2.33, v_fma 3.76, v_cmp 4.1-4.35,
# v_min/max 4.40, v_max3/min3 4.57, integer mix 4.22, v_fma_mix 4.39, v_cvt 4.11, v_lshl_add_u64 6.05, v_rcp 8.54; 1, 2 and 8 waves per
# SIMD price the 4-cycle classes the same (4.1-4.5), so these are pipeline rates, not a shortage of waves.
# Per opcode (tools/ubench/op_rate2.hip, profiles/r03p_op_rate2.txt): the 32-bit integer class is not one price -- v_and / v_or / v_xor /
# v_not / v_add_u32 / v_sub_u32 / v_lshrrev_b32 issue at 2.2-2.5 cycles, v_lshlrev_b32 / v_bfe / v_add3 / v_mul_lo / v_perm / v_bfi / v_bcnt /
# lshl_add / integer min-max at 4.2-4.5; two thirds of the shipped trace kernel's integer instructions are of the fast kind (static
# count), hence 3.1 with 2.5 / 4.3 as its bounds.  A fast-class instruction that reads an SGPR operand costs 4.4 (v_mul_f32 v, s, v), which
# no counter shows: the `high` bound is the honest one wherever the compiler keeps wave-uniform values in SGPRs.
VALU_PRICE = {"ADD_F32": 2.5, "MUL_F32": 2.6, "FMA_F32": 3.8, "TRANS_F32": 8.5, "INT32": 3.1, "INT64": 6.0, "CVT": 4.1,
              "ADD_F64": 4.2, "MUL_F64": 4.2, "FMA_F64": 4.2}
VALU_PRICE_INT32 = (2.5, 3.1, 4.3)     # low / point / high of the 32-bit integer class (see above)
VALU_PRICE_OTHER = (2.3, 3.6, 4.4)     # low (all moves), point, high (all compares / min-max / DPP / lane reads)


def valu_mix_weighted(p, secs):
    """Fraction of the SIMDs' time that the VALU pipe is occupied when every instruction is priced at what its class really
    costs: sum_class(count x cycles) / (1024 SIMDs x clock x duration).  None without the class counters."""
    if not p or "SQ_INSTS_VALU_ADD_F32" not in p or secs <= 0:
        return None
    classed = {k: float(p.get("SQ_INSTS_VALU_" + k, 0.0)) for k in VALU_PRICE}
    other = max(0.0, float(p["SQ_INSTS_VALU"]) - sum(classed.values()))
    base = sum(classed[k] * VALU_PRICE[k] for k in VALU_PRICE if k != "INT32")
    simd_cycles = N_SIMD * CLOCK_HZ * secs
    lo, mid, hi = ((base + classed["INT32"] * ci + other * c) / simd_cycles for ci, c in zip(VALU_PRICE_INT32, VALU_PRICE_OTHER))
    return {"value": round(mid, 3), "low": round(lo, 3), "high": round(hi, 3),
            "fp32_add_mul_fma_share": round((classed["ADD_F32"] + classed["MUL_F32"] + classed["FMA_F32"]) / float(p["SQ_INSTS_VALU"]), 3),
            "unclassed_share": round(other / float(p["SQ_INSTS_VALU"]), 3),
            "formula": "sum over SQ_INSTS_VALU_<class> of count x cycles-per-instruction-per-SIMD (tools/ubench/op_rate.hip and op_rate2.hip, 4 waves "
                       "per SIMD: add 2.5, mul 2.6, fma 3.8, transcendental 8.5, int32 2.5 (low) / 3.1 / 4.3 (high: and/or/add/sub/lshr cost 2.5, "
                       "lshl/bfe/mul/perm 4.3), int64 6.0, cvt 4.1, f64 4.2; unclassed = compares, min/max, selects, moves, DPP, lane reads at "
                       "2.3 (low) / 3.6 / 4.4 (high)) / (1024 SIMDs x 2.4 GHz x kernel seconds)"}


def alg_bytes(c, spp):
    """per ray N_branch*16 B + N_tri*40 B + N_hit*32 B, per pixel 12 B + 3 B."""
    per_sample = (c["branch_visits"] * 16 + c["tri_tests"] * 40 + c["hits"] * 32) / c["samples"]
    return per_sample + 15.0 / spp


def cpu_baseline(w, h, spp, budget_rows, scene_text=None):
    """The oracle (kind 'port': C restatement of the reference CPU algorithm; GHC cannot run here) timed on
    all host cores over a bounded sample of the same frame: every (w // budget_rows)-th row at full spp.
    scene_text = (obj, sq, camera) bytes for a generated scene; default: the reference's data/ files."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    data = os.path.join(ROOT, "data")
    if scene_text is None:
        ob = O.BIH(O.tris_from_obj(os.path.join(data, "scene.obj"), data))
        cam = O.load_camera(os.path.join(data, "camera"))
    else:
        ob = O.BIH(O.tris_from_text(scene_text[0], scene_text[1]))
        cam = O.camera_from_text(scene_text[2])
    cores = min(os.cpu_count() or 1, 16)          # the GPU box gives one GPU a 16-core share
    step = max(1, w // budget_rows)
    t0 = time.time()
    _, _, cnt = ob.render(cam, spp, w, h, threads=cores, rows=(step // 2, w), row_step=step, want_avg=False)
    dt = time.time() - t0
    rows = len(range(step // 2, w, step))
    return {"value": round(cnt["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{rows} rows (every {step}th) of the {w}x{h} @ {spp} spp frame = {cnt['samples']} samples in {dt:.1f} s"}, cnt


def algorithmic(cnt, spp, rays, secs):
    """the reference's visits priced at 16 B per branch, 40 B per triangle test, 32 B per hit, from the oracle's
    counters for bounce rays (depth >= 1), and the rate at which the GPU gets through them for the rays it traced."""
    bytes_per_ray = (cnt["b_branch_visits"] * 16 + cnt["b_tri_tests"] * 40 + cnt["b_hits"] * 32) / max(cnt["b_rays"], 1)
    return {"bytes_per_ray": round(bytes_per_ray, 1), "bytes_per_sample_reference": round(alg_bytes(cnt, spp), 1),
            "reference_branch_visits_per_ray": round(cnt["b_branch_visits"] / max(cnt["b_rays"], 1), 1),
            "reference_triangle_tests_per_ray": round(cnt["b_tri_tests"] / max(cnt["b_rays"], 1), 1),
            "GBps_for_rays_traced": round(bytes_per_ray * rays / secs / 1e9, 1) if secs > 0 else None}


def load_profile(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator comes up; rank 0's stdout must carry the JSON line only."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_ranks(args):
    """--gpus N without torchrun: start the N ranks as a child job and relay rank 0's line.  The parent makes no HIP
    call (torch.cuda.device_count() only counts devices on this image) and never replaces itself."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.oversubscribe and have >= 1):
        print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) are visible; refusing to report a "
              f"{args.gpus}-GPU number from fewer devices", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank job failed (rc {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def stale_note(prof, build_id):
    """None if the counter file was collected on the library that is loaded now, else why it must not be used."""
    if prof is None:
        return "no counter file under profiles/"
    have = prof.get("build_id")
    if have != build_id:
        return (f"counters under profiles/ were collected on build {have!r}, the loaded library is {build_id!r}: "
                f"re-run tools/collect_profiles.py on the GPU box")
    return None


OTHER = (("c3", "BASELINE configs[2] stand-in (procedural blob, no Stanford Bunny offline)", "blob", 6, 1920, 1080, 512),
         ("c5", "BASELINE configs[4] stand-in (jittered height-field)", "heightfield", 708, 1920, 1080, 256))


def other_scene_text(kind, size):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_scenes as G
    return G.blob_scene(size) if kind == "blob" else G.heightfield_scene(size)


def time_other_config(sqt, torch, name, text, w, h, spp, traffic_key, cpu_leg):
    """One stand-in configuration on this GPU through the resident API, second frame timed (the first one allocates).
    cpu_leg = (cpu_baseline record, oracle counters) measured before the GPU phase, or None."""
    obj, sq, camt = text
    cpu, cnt = cpu_leg if cpu_leg is not None else (None, None)
    mesh = sqt.Mesh.from_text(obj, sq)
    t = time.perf_counter(); bih = sqt.BIH(mesh, device=0 if len(mesh) >= 50000 else None); t_build = time.perf_counter() - t
    cam = sqt.camera_from_text(camt)
    ds = sqt.DeviceScene(bih, 0)
    ds.enable_timing()
    ds.render_rows(cam, spp, w, h, want_avg=False); torch.cuda.synchronize()
    ds.reset_timing(); ds.stats(reset=True)
    t = time.perf_counter(); ds.render_rows(cam, spp, w, h, want_avg=False); torch.cuda.synchronize(); dt = time.perf_counter() - t
    kern_ms, launches, kname = ds.kernel_timing()
    rays = ds.stats()[0]
    ds.close()
    out = {"workload": f"{name}: {len(mesh)} triangles, BIH height {bih.height}, {w}x{h} @ {spp} spp, one GPU",
           "value": round(w * h * spp / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 1),
           "bih_build_ms": round(t_build * 1e3, 1), "rays_traced": rays,
           "kernel": kname, "kernel_ms_total": round(kern_ms * launches, 1), "launches": launches}
    secs = kern_ms * launches * 1e-3
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["algorithmic"] = algorithmic(cnt, spp, rays, secs)
    prof = load_profile("latest_other_configs.json")
    why = stale_note(prof, sqt.build_id())
    if why is not None or traffic_key not in prof or kern_ms <= 0:
        out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                           "pmc_stale": why is not None, "note": why or "no counters for this configuration"}
        return out
    p = prof[traffic_key]      # per frame, sq_trace_rays launches: separate --pmc passes (tools/collect_profiles.py)
    traffic = p["hbm_bytes_per_frame"]
    mem = {"achieved": round(traffic / secs / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(traffic / secs / 1e9 / HBM_PEAK_GBS, 4),
           "what": "memory-side bytes of the trace launches of one frame (FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are "
                   "included, the guide's counters cannot separate them) / their measured duration",
           "l2_hit_rate": p.get("l2_hit_rate"), "wait_fraction_of_wave_cycles": p.get("wait_fraction_of_wave_cycles"),
           "vmem_load_wave_instructions_per_ray": round(p["vmem_rd_per_frame"] / max(rays, 1), 2) if p.get("vmem_rd_per_frame") else None}
    valu = None
    if p.get("SQ_INSTS_VALU"):
        lane_util = p["SQ_THREAD_CYCLES_VALU"] / (64.0 * p["SQ_ACTIVE_INST_VALU"])
        issue = p["SQ_INSTS_VALU"] / secs / (N_SIMD * CLOCK_HZ / 2)
        achieved = p["SQ_INSTS_VALU"] * 64 * lane_util / secs / 1e12
        valu = {"achieved": round(achieved, 2), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s", "frac": round(achieved / VALU_PEAK_TLANEOPS, 4),
                "valu_issue_frac": round(issue, 4), "valu_lane_utilisation": round(lane_util, 4), "valu_mix_weighted": valu_mix_weighted(p, secs)}
    # the larger fraction names the bound; both are reported
    if valu and valu["frac"] >= mem["frac"]:
        out["roofline"] = {"bound": "valu", **valu, "traffic": traffic, "memory": mem, "pmc_stale": False}
    else:
        out["roofline"] = {"bound": "hbm", **{k: mem[k] for k in ("achieved", "peak", "unit", "frac")}, "traffic": traffic, "memory": mem, "valu": valu, "pmc_stale": False}
    return out


def scaling_probe(sqt, torch, dist, d, scene, cam, w, h, spp, reps=3):
    """A PROJECTION, measured on this one GPU: how long one rank's share of the headline frame takes at N = 2, 4, 8 ranks
    (every shard of each partition rendered in turn; min and max over the shards, best of `reps`), and what one RCCL
    all_gather_into_tensor of the RGB8 frame costs with a world of one.  No multi-GPU run is behind these numbers."""
    out = {"what": "projection from ONE GPU: per-shard render time of the strong-scaling partition (interleaved blocks of "
                   f"{d.ROW_BLOCK} rows), every shard of each N timed alone on this device; not a multi-GPU measurement",
           "frame": [w, h, spp], "shares": {}}
    def timed(fn):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            best = dt if best is None else min(best, dt)
        return best
    whole = timed(lambda: scene.render_rows(cam, spp, w, h, want_avg=False))
    out["whole_frame_ms"] = round(whole, 3)
    for n in (2, 4, 8):
        ms = [timed(lambda r=r: scene.render_rows(cam, spp, w, h, want_avg=False, shard=(d.ROW_BLOCK, r, n))) for r in range(n)]
        out["shares"][str(n)] = {"per_shard_ms": [round(min(ms), 3), round(max(ms), 3)],
                                 "speedup_before_gather": round(whole / max(ms), 2)}
    # one collective of the frame over RCCL, world size 1 (what the library call itself costs; no link is crossed)
    try:
        created = False
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            with stdout_to_stderr():
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
            created = True
        frame = torch.zeros((w, h, 3), dtype=torch.uint8, device="cuda")
        with stdout_to_stderr():
            d.gather_frame(frame, w); torch.cuda.synchronize()         # first call: communicator set-up
        out["rccl_all_gather_world1_ms"] = round(timed(lambda: d.gather_frame(frame, w)), 3)
        out["rccl_world_size"] = dist.get_world_size()
        if created:
            dist.destroy_process_group()
    except Exception as e:                                             # the probe must never cost the headline line
        out["rccl_all_gather_world1_ms"] = None
        out["rccl_error"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="c2 = 1920x1080 @ 256 spp (headline), c4 = 3840x2160 @ 1024 spp")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the configuration's frame is shared by the ranks; weak: spp is multiplied by the rank count")
    ap.add_argument("--width", type=int, default=0)         # overrides of the configuration (first dimension = image ROWS)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--cpu-rows", type=int, default=240, help="rows of the frame timed on the CPU oracle (240 rows = 66 M samples: about 10 s on 16 cores)")
    ap.add_argument("--other-cpu-rows", type=int, default=6, help="rows of the C3 / C5 stand-in frames timed on the CPU oracle (6 rows spread over the frame at full spp: 10-20 s each on 16 cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the C3/C5 stand-ins, the C4 frame and the scaling probe (N = 1 only)")
    ap.add_argument("--no-oneshot", action="store_true", help="skip the one-shot call's wall time (profiling runs: keeps the launch count per step)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and gather even with one rank (self-test)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (the product path); gloo only for plumbing tests")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: let ranks share devices (rank r uses device r %% visible) so that the N-rank path can be "
                         "exercised on a one-GPU box with --backend gloo; the line it prints is marked and is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be positive")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    single = world_env == 1 and not args.force_dist          # CPU baseline, roofline and the other configurations are N = 1 figures
    cw, ch, cspp = CONFIGS[args.config]
    w, h, base_spp = args.width or cw, args.height or ch, args.spp or cspp
    cpu = cnt = None
    with_other = single and not args.no_other and args.config == "c2" and not (args.width or args.height or args.spp)
    other_text, other_cpu = {}, {}
    if not args.no_cpu and single:
        # The CPU legs run FIRST, before anything touches the GPU: they are ~10 s of host work each, and with them out of the way
        # the GPU phase of the run is one contiguous block that an outside utilisation sampler can see.
        cpu, cnt = cpu_baseline(w, h, base_spp, args.cpu_rows)
    if with_other:
        for key, _, kind, size, ow, oh, ospp in OTHER:
            other_text[key] = other_scene_text(kind, size)
            if not args.no_cpu and args.other_cpu_rows > 0:
                other_cpu[key] = cpu_baseline(ow, oh, ospp, args.other_cpu_rows, scene_text=other_text[key])
    import torch
    import torch.distributed as dist
    sqt = importlib.import_module("squigly-trace_amd")
    d = importlib.import_module("squigly-trace_amd.dist")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.oversubscribe and torch.cuda.device_count() >= 1:
        local_rank %= torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    use_dist = world_env > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
            dist.barrier()                                    # brings the communicator up (and its banner) before anything is timed
    world = dist.get_world_size() if use_dist else 1          # the ranks RCCL actually connected

    data = os.path.join(ROOT, "data")
    bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(data, "scene.obj"), data))
    cam = sqt.load_camera(os.path.join(data, "camera"))
    scene = sqt.DeviceScene(bih, local_rank)                 # resident in HBM before the timed region
    scene.enable_timing()                                    # hipEvents around every sq_trace_rays launch, on its stream
    spp = base_spp * world if args.scaling == "weak" else base_spp
    events = [] if use_dist else None

    def step():
        return d.render_frame(scene, cam, spp, w, h, want="rgb", events=events)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    scene.reset_timing()
    scene.stats(reset=True)
    if events is not None:
        events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step()
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own render and gather time per frame (events on its stream), so that a shortfall against N x the
        # one-GPU number can be attributed: imbalance (spread of render_ms), fixed cost (min render_ms x N against the
        # one-GPU frame) or the collective (gather_ms)
        mine = [sum(a.elapsed_time(b) for a, b, _ in events) / max(len(events), 1),
                sum(b.elapsed_time(c) for _, b, c in events) / max(len(events), 1)]
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        per_rank = [[round(float(x[0]), 3), round(float(x[1]), 3)] for x in allr]
    kern_ms, launches, kname = scene.kernel_timing()
    stats = scene.stats()
    samples_per_step = w * h * spp                           # all ranks together
    value = samples_per_step * args.steps / elapsed / 1e6

    if rank == 0:
        out = {
            "metric": "Msamples/s (whole node) at 1080p/256spp on data/scene.obj",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            **({"not_a_measurement": "ranks share devices (--oversubscribe) or gather over gloo: plumbing test only"}
               if (args.oversubscribe or args.backend != "nccl") else {}),
            "vs_baseline": None, "dtype": "f32", "data": "data/scene.obj + data/scene.sq + data/camera (the reference's sample scene)",
            "parity": "bit-identical to this repository's C restatement of the Haskell (oracle/); unpinned against a GHC build of the reference",
            "build_id": sqt.build_id(),
            "config": {"workload": f"data/scene.obj {w}x{h} @ {spp} spp (BASELINE configs[{1 if args.config == 'c2' else 3}]"
                                   f"{'' if (w, h, base_spp) == CONFIGS[args.config] else ', size overridden'}), "
                                   f"rows sharded over {world} GPU(s) in blocks of {d.ROW_BLOCK}, RGB8 all_gather_into_tensor",
                       "samples_per_step": samples_per_step, "row_block": d.ROW_BLOCK,
                       "nonblack_pixels": int((frame.sum(-1) > 0).sum().item())},
        }
        if per_rank is not None:
            rms = [x[0] for x in per_rank]; gms = [x[1] for x in per_rank]
            out["ranks"] = {"backend": args.backend, "world_size": world,
                            "per_rank_ms": [min(rms), max(rms)], "gather_ms": [min(gms), max(gms)],
                            "what": "per frame, from events on each rank's stream: its own rows rendered (per_rank_ms) and the "
                                    "all_gather + de-interleave that follows (gather_ms; includes waiting for the slowest rank); [min, max] over ranks"}
        if single and not args.no_oneshot and args.config == "c2" and not (args.width or args.height or args.spp):
            # SURVEY 8(d) asks for the rate with the scene upload included as well: the one-shot drop-in call
            # (upload + workspace + render + copy back over PCIe), second call timed.  Reported beside `value`, never as it.
            sqt.render_rgb8(bih, cam, spp, (w, h))
            t1 = time.perf_counter()
            sqt.render_rgb8(bih, cam, spp, (w, h))
            dt = time.perf_counter() - t1
            out["one_shot_call"] = {"ms": round(dt * 1e3, 2), "msamples_per_s": round(samples_per_step / dt / 1e6, 1),
                                    "what": "sq_render_rgb8: scene upload + render + 6.2 MB copy back, host buffers in and out"}
        if single and with_other and not args.no_oneshot and args.config == "c2" and not (args.width or args.height or args.spp):
            # (not in --no-other runs: those are the profiling commands, whose per-kernel averages must stay those of the serial schedule)
            # The same frames with the library's two-pipeline schedule (option overlap = 2: even and odd sample batches on two
            # streams, one pipeline's launches filling the other's ramp-downs and per-sample kernels).  Same image bit for bit.
            # Reported beside `value`, not as it: `value` and the roofline are measured on the serial schedule, where a launch
            # has the GPU to itself and per-kernel durations mean what they say.
            try:
                scene.set_option("overlap", 2)
                same = bool((scene.render_rows(cam, spp, w, h, want_avg=False)[1] == frame).all().item())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    scene.render_rows(cam, spp, w, h, want_avg=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 5
                out["overlapped_schedule"] = {"option": "overlap=2", "ms_per_step": round(dt * 1e3, 3), "msamples_per_s": round(samples_per_step / dt / 1e6, 1),
                                              "same_image": same, "what": "two sample-batch pipelines on two streams; not `value` (see roofline note)"}
            except Exception as e:
                out["overlapped_schedule"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                scene.set_option("overlap", 0)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if single and launches:
            # Roofline of the dominant kernel, sq_trace_rays.  The 0.3 MB scene lives in LDS, so HBM is not what binds it
            # (measured traffic is ~2 % of the HBM peak); the resource that binds is the VALU pipe.  VALU wave-instructions by class,
            # active cycles and thread cycles come from separate rocprofv3 --pmc passes of this same command on THIS build
            # (profiles/latest_pmc.json carries the build id it was collected on; the counts depend only on the workload), the
            # duration is measured live here with hipEvents on the launch stream.
            launches_per_step = launches / args.steps
            trace_s_per_step = kern_ms * 1e-3 * launches_per_step
            rays = stats[0] / args.steps
            pmc = load_profile("latest_pmc.json")
            why = stale_note(pmc, sqt.build_id())
            roof = {"kernel": kname, "kernel_ms": round(kern_ms, 3), "launches": launches, "launches_per_step": launches_per_step,
                    "rays_per_launch": int(rays / launches_per_step)}
            if why is None and pmc.get("workload") == [w, h, spp]:
                insts, active, threads = pmc["SQ_INSTS_VALU"], pmc["SQ_ACTIVE_INST_VALU"], pmc["SQ_THREAD_CYCLES_VALU"]
                lane_util = threads / (64.0 * active)
                issue = insts / trace_s_per_step / (N_SIMD * CLOCK_HZ / 2)
                achieved = insts * 64 * lane_util / trace_s_per_step / 1e12
                traffic = (pmc["FETCH_SIZE_KB"] * 2 + pmc["WRITE_SIZE_KB"]) * 1024 / launches_per_step
                roof.update({"bound": "valu", "achieved": round(achieved, 2), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s",
                             "frac": round(achieved / VALU_PEAK_TLANEOPS, 4), "pmc_stale": False, "counters_from_build": pmc.get("build_id"),
                             "valu_issue_frac": round(issue, 4), "valu_lane_utilisation": round(lane_util, 4),
                             "valu_wave_instructions_per_launch": int(insts / launches_per_step),
                             # the same instruction stream priced per class: how full the VALU pipe is, as one number
                             "valu_mix_weighted": valu_mix_weighted(pmc, trace_s_per_step),
                             # rocprofv3's VALUBusy expression (4 x SQ_ACTIVE_INST_VALU / SIMDs / cycles), for information
                             "valu_busy_rocprof_expr": round(4.0 * active / (N_SIMD * CLOCK_HZ * trace_s_per_step), 3),
                             "wave_cycles": {k: round(pmc[c] / pmc["SQ_WAVE_CYCLES"], 3) for k, c in
                                             (("waiting", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"))
                                             if pmc.get(c) and pmc.get("SQ_WAVE_CYCLES")},
                             # the other co-limiting unit: the CU's LDS (scene, stacks and the ds_bpermute ray pulls all go
                             # through it).  array_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles of the trace launches).
                             "lds": {"array_busy_frac": round(pmc["SQ_LDS_IDX_ACTIVE"] / (N_SIMD / 4 * CLOCK_HZ * trace_s_per_step), 4),
                                     "bank_conflict_share": round(pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"], 4),
                                     "instructions_per_launch": int(pmc["SQ_INSTS_LDS"] / launches_per_step)},
                             "traffic": int(traffic),
                             "hbm": {"achieved": round(traffic / (kern_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                             "note": "frac = valu_issue_frac x valu_lane_utilisation: busy VALU lanes / (1024 SIMDs x 32 lanes x 2.4 GHz), i.e. against "
                                     "the fp32-FMA issue rate; valu_mix_weighted prices each instruction class at its measured cost instead "
                                     "(the time elasticity of injected VALU instructions was 0.5 in round 2, DESIGN.md 4.7: not re-measured in this run). "
                                     "Counters: profiles/latest_pmc.json (separate --pmc passes, stamped with the build id); duration: live hipEvents. "
                                     "traffic = HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE): ray fetch and hit store only."})
            else:
                roof.update({"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s", "frac": None,
                             "traffic": None, "pmc_stale": why is not None,
                             "note": why or "profiles/latest_pmc.json was collected for another workload"})
            if cnt is not None:
                # SURVEY 8(d)'s algorithmic bytes, kept for reference: they are served from LDS, not HBM, so they
                # are not a roofline fraction.  Unit = one bounce ray (oracle counters for rays at depth >= 1).
                roof["algorithmic"] = {**algorithmic(cnt, base_spp, rays, trace_s_per_step), "served_from": "LDS (scene resident per CU)"}
            out["roofline"] = roof
        if with_other:
            out["other_configs"] = {}
            try:
                out["other_configs"]["scaling_probe"] = scaling_probe(sqt, torch, dist, d, scene, cam, w, h, spp)
            except Exception as e:
                out["other_configs"]["scaling_probe"] = {"error": f"{type(e).__name__}: {e}"}
            try:                                       # BASELINE configs[3] on ONE GPU: the anchor an 8-GPU C4 run is compared with
                c4w, c4h, c4s = CONFIGS["c4"]
                scene.render_rows(cam, c4s, c4w, c4h, want_avg=False); torch.cuda.synchronize()
                t1 = time.perf_counter(); scene.render_rows(cam, c4s, c4w, c4h, want_avg=False); torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                out["other_configs"]["c4"] = {"workload": f"data/scene.obj {c4w}x{c4h} @ {c4s} spp (BASELINE configs[3]) on ONE GPU, second frame timed",
                                              "value": round(c4w * c4h * c4s / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 1)}
            except Exception as e:
                out["other_configs"]["c4"] = {"error": f"{type(e).__name__}: {e}"}
            scene.close()
            scene = None
            sqt.release_cached_memory()
            for key, name, _, _, ow, oh, ospp in OTHER:
                try:
                    out["other_configs"][key] = time_other_config(sqt, torch, name, other_text[key], ow, oh, ospp, key, other_cpu.get(key))
                except Exception as e:            # the headline line must not be lost to a side measurement
                    out["other_configs"][key] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if scene is not None:
        scene.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
