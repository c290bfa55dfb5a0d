import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MESH = os.path.join(GOLDEN, "meshes")
SPOT = os.path.join(MESH, "spot_triangulated.obj")
BLUB = os.path.join(MESH, "blub_triangulated.obj")
DEFAULT_CAM = (3, 5, 3, 0, 0, 0, 0, -1, 0, 45)

# the scenes of the golden vectors (tests/golden/make_golden.py): name -> meshes, W, H, camera
CASES = {
    "spot_640x480": ([(SPOT, True, (255, 0, 0), (0, 0, 0))], 640, 480, DEFAULT_CAM),
    "main_512x512": ([(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (255, 0, 0), (1.5, 0, 0))], 512, 512, DEFAULT_CAM),
    "main_odd_333x217": ([(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (255, 0, 0), (1.5, 0, 0))], 333, 217, DEFAULT_CAM),
    "two_colors_200x150": ([(SPOT, True, (10, 200, 30), (0, 0, 0)), (BLUB, True, (0, 90, 255), (1.5, 0, 0))], 200, 150, DEFAULT_CAM),
    "spot_inside_97x61": ([(SPOT, True, (255, 0, 0), (0, 0, 0))], 97, 61, (0.1, 0.2, 0.3, 0, 0, 1, 0, -1, 0, 70)),
    "plane10_480x270": ([("PLANE10", True, (255, 0, 0), (0, 0, 0))], 480, 270, DEFAULT_CAM),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product package (ctypes binding of librt_amd.so)."""
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def O():
    """The oracle bindings (test infrastructure)."""
    import oracle_py
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        pytest.fail("oracle/liboracle.so is missing: run `make -C oracle` or __graft_entry__.build()")
    return oracle_py


@pytest.fixture(scope="session")
def plane10(rt, tmp_path_factory):
    p = str(tmp_path_factory.mktemp("plane") / "plane10.obj")
    rt.make_plane_obj(p, 10)
    return p


def cam_kwargs(cam):
    return dict(cam_pos=tuple(cam[0:3]), cam_target=tuple(cam[3:6]), cam_up=tuple(cam[6:9]), fov=float(cam[9]))


def load_case(scene_add, name, plane10):
    """Feeds the meshes of a golden case to scene_add(path, has_vt, color, offset)."""
    meshes, W, H, cam = CASES[name]
    for path, vt, col, off in meshes:
        scene_add(plane10 if path == "PLANE10" else path, vt, col, off)
    return W, H, cam


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["ids"], z["rgb"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
