"""bench.py's pure helpers (no GPU): the class-priced VALU figure, the staleness check of counter files, the build id."""
import importlib
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("sq_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_valu_mix_weighted_is_the_priced_sum_over_simd_cycles():
    b = _bench()
    p = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_ADD_F32": 200.0, "SQ_INSTS_VALU_MUL_F32": 100.0, "SQ_INSTS_VALU_FMA_F32": 50.0,
         "SQ_INSTS_VALU_TRANS_F32": 10.0, "SQ_INSTS_VALU_INT32": 140.0}
    secs = 1e-9
    m = b.valu_mix_weighted(p, secs)
    other = 1000 - 500
    base = 200 * b.VALU_PRICE["ADD_F32"] + 100 * b.VALU_PRICE["MUL_F32"] + 50 * b.VALU_PRICE["FMA_F32"] + 10 * b.VALU_PRICE["TRANS_F32"] + 140 * b.VALU_PRICE["INT32"]
    cyc = b.N_SIMD * b.CLOCK_HZ * secs
    assert abs(m["value"] - (base + other * b.VALU_PRICE_OTHER[1]) / cyc) < 1e-3
    assert m["low"] <= m["value"] <= m["high"] and m["unclassed_share"] == 0.5 and m["fp32_add_mul_fma_share"] == 0.35
    assert b.valu_mix_weighted({"SQ_INSTS_VALU": 5.0}, 1.0) is None and b.valu_mix_weighted(None, 1.0) is None
    # the committed counter file prices the headline kernel between its bounds
    prof = b.load_profile("latest_pmc.json")
    if prof and "SQ_INSTS_VALU_ADD_F32" in prof:
        m = b.valu_mix_weighted(prof, 0.043)
        assert 0.4 < m["low"] < m["value"] < m["high"] < 1.3


def test_counter_files_are_tied_to_a_build():
    b = _bench()
    assert b.stale_note(None, "x") is not None
    assert b.stale_note({"build_id": "abc"}, "abc") is None
    why = b.stale_note({"build_id": "abc"}, "def")
    assert "abc" in why and "def" in why
    assert "build_id" not in {} and b.stale_note({}, "def") is not None            # a file from before the stamp is stale by definition
    for name in ("latest_pmc.json", "latest_other_configs.json"):
        prof = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert re.fullmatch(r"[0-9a-f]{16}", prof["build_id"]), name


def test_build_id_is_the_hash_of_the_sources_and_is_in_the_library():
    sqt = importlib.import_module("squigly-trace_amd")
    spec = importlib.util.spec_from_file_location("sq_build", os.path.join(ROOT, "squigly-trace_amd", "build.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    assert sqt.build_id() == bm.source_id() and not bm.stale()                     # the in-tree library is built from the sources that are here
    assert bm.source_id(("-DX=1",)) != bm.source_id()                              # a variant build has another identity
