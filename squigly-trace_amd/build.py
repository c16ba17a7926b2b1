"""Build libsquigly_hip.so (HIP kernels for gfx950 + host side) in-tree with hipcc.

    python squigly-trace_amd/build.py [--force]

The flags that matter for bit-exact parity with the CPU oracle:
  -ffp-contract=off                               no FMA contraction (host and device)
  -fhip-fp32-correctly-rounded-divide-sqrt        IEEE fp32 divide / sqrt on the device
  -fno-slp-vectorize                              performance only: packed fp32 VALU ops are slow on gfx950
  no -ffast-math, denormals kept (hipcc default for gfx9)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsquigly_hip.so")
SOURCES = ["sq_device.hip", "sq_bih_device.hip", "sq_host.cpp"]
HEADERS = ["sq_math.h", "sq_error.h", "sq_scene.h", "sq_host_types.h", "cli_main.cpp", "../../include/squigly_hip.h", "../../include/squigly_host.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         # v_pk_mul_f32 / v_pk_add_f32 issue at ~9.5 cycles per wave on gfx950 against ~2.6 for the scalar forms
         # (tools/ubench/valu_rate.hip): keep the SLP vectoriser from packing the fp32 vector math.
         "-fno-slp-vectorize",
         # performance only (scheduling; bits unchanged): the trace kernel is bound by the latency of its dependent steps
         # at 4 waves per SIMD, and the ILP-first machine scheduler measured -0.5 ... -1.3 % on all three scenes against
         # the default (same gpurun call, tools/ab_variant.sh); max-memory-clause +-0, iterative-minreg +2 %
         "-mllvm", "-amdgpu-sched-strategy=max-ilp",
         "-Wall", "-Wno-unused-function"]


def source_id(extra=()):
    """First 16 hex digits of SHA-256 over every source and header of the library, the flags and the extra -D options:
    what sq_build_id() returns, and what tools/collect_profiles.py stamps its counter files with."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(f.encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update("\0".join(FLAGS + sorted(extra)).encode())
    return h.hexdigest()[:16]


def id_flag(extra=()):
    return ['-DSQ_BUILD_ID="%s"' % source_id(extra)]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    """True unless the in-tree library was built from exactly the sources and flags that are here now: it must carry the
    current source_id() (file times alone can lie after a checkout or a copy)."""
    if not os.path.exists(OUT):
        return True
    with open(OUT, "rb") as f:
        if source_id().encode() not in f.read():
            return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


CLI = os.path.join(HERE, "bin", "squigly-trace")


def build(force=False, extra=(), out=None):
    """out: build an experimental variant (extra -D flags) beside the product library; SQ_LIB_PATH selects it at load time."""
    if out is not None:
        subprocess.check_call([hipcc()] + FLAGS + list(extra) + id_flag(extra) + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out])
        return out
    if not force and not stale() and os.path.exists(CLI):
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + id_flag(extra) + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    subprocess.check_call(cmd)
    # the reference's executable (app/Main.hs) over the C-ABI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(CSRC, "cli_main.cpp"), "-o", CLI,
                           "-L" + HERE, "-lsquigly_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])
    return OUT


if __name__ == "__main__":
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    # -Dname[=v] / -Rpass / -save-temps go to hipcc as they are; --flag=<anything> passes <anything> (e.g. --flag=-mllvm --flag=-amdgpu-skip-threshold=24)
    print(build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a[:2] in ("-R", "-D") or a.startswith("-save")] + [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--flag=")],
                out=os.path.join(HERE, outs[0]) if outs else None))
