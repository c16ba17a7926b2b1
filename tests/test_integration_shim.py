"""The reference-side binding (integration/LibHip.hs + integration/reference.patch) checked by inspection, mechanically.

No GHC exists here, so the shim cannot be compiled.  What CAN be checked: that reference.patch applies to the
reference as it is, and that every name LibHip.hs and DumpGolden.hs import from a reference module is in that module's
export list once the patch is applied (the shim named constructors src/BIH.hs does not export).
Runs only where /root/reference exists (the build container); skipped on the GPU box.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine")


def exports_of(path):
    """Names in a module's export list ('Name' and 'Type(..)' both as 'Name'); None = no list, everything is exported."""
    text = open(path).read()
    text = re.sub(r"\{-.*?-\}", "", text, flags=re.S)
    text = re.sub(r"--.*", "", text)
    m = re.search(r"\bmodule\s+[\w.]+\s*(\((.*?)\))?\s*where", text, flags=re.S)
    assert m, path
    if m.group(1) is None:
        return None
    body = re.sub(r"\(\.\.\)", "", m.group(2))
    return {n.strip() for n in body.split(",") if n.strip()}


def imports_of(path):
    """[(module, [names])] for every `import M (a, B (..), c)` with an explicit list."""
    text = open(path).read()
    text = re.sub(r"--.*", "", text)
    out = []
    for m in re.finditer(r"^import\s+(?:qualified\s+)?([\w.]+)(?:\s+as\s+\w+)?\s*\(((?:[^()]|\([^()]*\))*)\)", text, flags=re.M):
        names = [re.sub(r"\s*\(.*\)\s*$", "", n).strip() for n in re.split(r",(?![^()]*\))", m.group(2))]
        out.append((m.group(1), [n for n in names if n]))
    return out


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref")
    for rel in ("src", "app", "squigly-trace.cabal"):
        src = os.path.join(REF, rel)
        (shutil.copytree if os.path.isdir(src) else shutil.copy)(src, os.path.join(d, rel))
    r = subprocess.run(["patch", "-p1", "--batch", "-i", os.path.join(ROOT, "integration", "reference.patch")], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return d


def test_patch_applies_and_is_three_small_hunks(patched):
    text = open(os.path.join(ROOT, "integration", "reference.patch")).read()
    files = re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M)
    assert sorted(files) == ["app/Main.hs", "squigly-trace.cabal", "src/BIH.hs"]
    added = [l for l in text.splitlines() if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in text.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert len(added) <= 10 and len(removed) <= 3          # "what a maintainer changes" stays a handful of lines
    assert "renderHip scene cam settings" in open(os.path.join(patched, "app", "Main.hs")).read()
    assert "LibHip" in open(os.path.join(patched, "squigly-trace.cabal")).read()


@pytest.mark.parametrize("shim", ["LibHip.hs", "DumpGolden.hs"])
def test_every_name_imported_from_the_reference_is_exported(patched, shim):
    ref_modules = {os.path.splitext(f)[0]: os.path.join(patched, "src", f) for f in os.listdir(os.path.join(patched, "src")) if f.endswith(".hs")}
    checked = 0
    for module, names in imports_of(os.path.join(ROOT, "integration", shim)):
        if module not in ref_modules:
            continue
        exp = exports_of(ref_modules[module])
        if exp is None:
            continue                                        # e.g. V3: no export list
        for n in names:
            assert n in exp, f"{shim}: `{n}` is not exported by {module} ({sorted(exp)})"
            checked += 1
    assert checked >= 5


def test_unpatched_bih_lacks_the_constructors_the_patch_adds():
    """The reason for hunk 1: without it LibHip.hs cannot see Leaf / Branch / BIHN."""
    exp = exports_of(os.path.join(REF, "src", "BIH.hs"))
    assert not ({"Tree", "BIHNode", "BIHTree"} & exp)


def test_shim_uses_the_extensions_and_imports_it_needs():
    text = open(os.path.join(ROOT, "integration", "LibHip.hs")).read()
    assert "RecordWildCards" in text and "Settings {..}" in text             # src/Lib.hs:13 enables it for the same pattern
    assert "ForeignFunctionInterface" in text
    assert re.search(r"import qualified Graphics\.ColorSpace\s+as M", text)   # Pixel / RGB / PixelRGB, as src/Lib.hs:39-40
    assert "M.PixelRGB" in text and "M.Pixel M.RGB Word8" in text
    # sq_scene offsets used by the pokes = the C header's layout
    hdr = open(os.path.join(ROOT, "include", "squigly_hip.h")).read()
    assert re.search(r"sq_bounds\s+root;.*const sq_node\*\s+nodes;\s+int32_t n_nodes;.*const sq_tri\*\s+tris;\s+int32_t n_tris;.*"
                     r"const sq_material\*\s+mats;\s+int32_t n_mats;.*int32_t\s+height;", hdr, flags=re.S)
    # ... and the ctypes mirror of the same struct agrees on every offset the shim pokes at (x86-64)
    import ctypes
    import importlib
    nat = importlib.import_module("squigly-trace_amd._native")
    S = nat.Scene
    assert (S.nodes.offset, S.n_nodes.offset, S.tris.offset, S.n_tris.offset, S.mats.offset, S.n_mats.offset, S.height.offset,
            ctypes.sizeof(S), ctypes.sizeof(nat.Camera)) == (24, 32, 40, 48, 56, 64, 68, 72, 48)
    for off in ("pScene 24", "pScene 32", "pScene 40", "pScene 48", "pScene 56", "pScene 64", "pScene 68"):
        assert off in text
