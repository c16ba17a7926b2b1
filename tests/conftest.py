import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _load_native_code_first():
    """Build/load both shared libraries before any test can initialise the GPU (no process spawning after that)."""
    import pyoracle
    pyoracle.lib()
    mod = importlib.import_module("squigly-trace_amd")
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    mod.lib()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker only)."""
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def sqt():
    """The product package; libsquigly_hip.so must already be built (__graft_entry__.build)."""
    mod = importlib.import_module("squigly-trace_amd")
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def oracle_scene(O):
    tris = O.tris_from_obj(os.path.join(DATA, "scene.obj"), DATA)
    return O.BIH(tris), O.load_camera(os.path.join(DATA, "camera")), tris


@pytest.fixture(scope="session")
def product_scene(sqt):
    mesh = sqt.Mesh.from_obj(os.path.join(DATA, "scene.obj"), DATA)
    return sqt.BIH(mesh), sqt.load_camera(os.path.join(DATA, "camera")), mesh
