"""Generates the golden vectors under tests/golden/ from the reference itself (oracle/_ref/libref.so,
compiled from /root/reference/Serial by oracle/Makefile). Runs only where /root/reference exists.
    python tests/golden/make_golden.py
The vectors are data: per-pixel primary-hit triangle ids (int32, -1 = miss) and the float image
(3 floats per pixel) that render() would hand to its PPM writer. The two OBJ meshes under
tests/golden/meshes/ are the reference's input data files (Serial/*.obj), copied byte for byte.
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

MESH = os.path.join(HERE, "meshes")
SPOT = os.path.join(MESH, "spot_triangulated.obj")
BLUB = os.path.join(MESH, "blub_triangulated.obj")
DEFAULT_CAM = (3, 5, 3, 0, 0, 0, 0, -1, 0, 45)

# name -> (meshes [(path, has_vt, color, offset)], W, H, camera)
CASES = {
    "spot_640x480": ([(SPOT, True, (255, 0, 0), (0, 0, 0))], 640, 480, DEFAULT_CAM),
    "main_512x512": ([(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (255, 0, 0), (1.5, 0, 0))], 512, 512, DEFAULT_CAM),
    "main_odd_333x217": ([(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (255, 0, 0), (1.5, 0, 0))], 333, 217, DEFAULT_CAM),
    "two_colors_200x150": ([(SPOT, True, (10, 200, 30), (0, 0, 0)), (BLUB, True, (0, 90, 255), (1.5, 0, 0))], 200, 150, DEFAULT_CAM),
    "spot_inside_97x61": ([(SPOT, True, (255, 0, 0), (0, 0, 0))], 97, 61, (0.1, 0.2, 0.3, 0, 0, 1, 0, -1, 0, 70)),
    "plane10_480x270": ([("PLANE10", True, (255, 0, 0), (0, 0, 0))], 480, 270, DEFAULT_CAM),
}


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    plane = os.path.join(HERE, "_plane10.obj")
    rt = importlib.import_module("ray-tracer_amd")
    rt.make_plane_obj(plane, 10)
    ref_plane = "/root/reference/Parallel/plane.obj"
    assert open(plane, "rb").read() == open(ref_plane, "rb").read(), "plane generator does not reproduce Parallel/plane.obj"
    print("plane generator output == Parallel/plane.obj, md5", hashlib.md5(open(plane, "rb").read()).hexdigest())
    summary = {}
    for name, (meshes, W, H, cam) in CASES.items():
        r = O.RefScene()
        for path, vt, col, off in meshes:
            r.add_obj(plane if path == "PLANE10" else path, vt, col, off)
        rgb, ids, _ = r.render(W, H, cam=cam)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=ids, rgb=rgb, cam=np.array(cam, np.float32))
        summary[name] = dict(hits=int((ids >= 0).sum()), ids_md5=md5(ids), rgb8_md5=md5(O.quantize(rgb)))
        print(name, summary[name])
    os.remove(plane)
    with open(os.path.join(HERE, "SUMMARY.txt"), "w") as f:
        f.write("plane10_obj_md5 %s\n" % hashlib.md5(open(ref_plane, "rb").read()).hexdigest())
        for k, v in summary.items():
            f.write("%s hits=%d ids_md5=%s rgb8_md5=%s\n" % (k, v["hits"], v["ids_md5"], v["rgb8_md5"]))


if __name__ == "__main__":
    main()
