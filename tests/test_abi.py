"""CPU: the C-ABI library loads, exports everything include/rt_amd.h declares, and its host logic
(parameters, OBJ reader, partition arithmetic, PPM writer, plane generator) matches the reference's
behaviour as pinned by the oracle. No compute call runs here."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import BLUB, GOLDEN, ROOT, SPOT, bits


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read() + open(os.path.join(ROOT, "include", "rt_amd_diag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    L = rt.lib()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"librt_amd.so does not export {n}"
    assert sorted(rt.EXPORTS) == names


def test_default_params_are_the_reference_literals(rt):
    p = rt.default_params()
    assert (p.width, p.height) == (512, 512)
    assert list(p.cam_pos) == [3, 5, 3] and list(p.cam_target) == [0, 0, 0] and list(p.cam_up) == [0, -1, 0]
    assert p.fov == 45 and list(p.light_pos) == [5, -5, 2]
    assert p.kd == 2.0 and p.ks == np.float32(5e11) and p.ka == np.float32(0.2)
    assert p.spec_alpha == 4 and p.light_intensity == 255 and p.shadow_scale == np.float32(0.1) and p.eps == np.float32(0.1)
    assert p.walk_mode == rt.RT_WALK_EARLY and (p.part_index, p.part_count) == (0, 1)


def test_part_rows(rt):
    p = rt.default_params(100, 2160)
    assert rt.part_rows(p) == 2160
    for n, rows in [(2, 1080), (4, 544), (8, 272), (3, 720), (7, 312)]:
        p.part_count = n
        assert rt.part_rows(p) == rows
    p = rt.default_params(10, 61, part_count=3)
    assert rt.part_rows(p) == 24      # 8 bands of 8 rows over 3 parts -> 3 bands each


def test_obj_reader_matches_oracle(rt, O):
    for off in [(0, 0, 0), (1.5, 0, 0), (0.1, -0.7, 1e-3)]:
        s, o = rt.Scene(), O.OracleScene()
        assert s.load_mesh(SPOT, True, (255, 0, 0), off) == o.add_obj(SPOT, True, (255, 0, 0), off) == 5856
        assert s.load_mesh(BLUB, True, (1, 2, 3), off) == o.add_obj(BLUB, True, (1, 2, 3), off) == 14208
        assert np.array_equal(bits(s.triangles()), bits(o.triangles()))


def test_obj_reader_edge_cases(rt, O, tmp_path):
    # comments, blank lines, vn/vt records, CRLF, faces without slashes, faces with v/vt/vn, a quad (tail dropped),
    # leading whitespace, exponents
    text = ("# comment\n\n  v 0 0 0\nv 1e0 0 0\r\nv 0 1 0\nv 0 0 1.5e+0\nvn 0 0 1\nvt 0.5 0.5\n"
            "f 1 2 3\nf 1/1 2/1 4/1\r\nf 2/1/1 3/1/1 4/1/1\nf 1 2 3 4\ng group\ns off\nusemtl x\n")
    p = tmp_path / "e.obj"
    p.write_bytes(text.encode())
    s, o = rt.Scene(), O.OracleScene()
    assert s.load_mesh(str(p), False, (1, 1, 1), (0.25, 0, 0)) == o.add_obj(str(p), False, (1, 1, 1), (0.25, 0, 0)) == 4
    assert np.array_equal(bits(s.triangles()), bits(o.triangles()))
    # a missing file is an empty mesh, like the reference
    assert s.load_mesh(str(tmp_path / "missing.obj")) == 0
    # an index beyond the vertices read so far is an error here (the reference reads out of bounds)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\n")
    with pytest.raises(rt.RtError):
        s.load_mesh(str(bad))
    assert s.num_triangles == 4
    # an empty file
    empty = tmp_path / "empty.obj"
    empty.write_text("")
    assert s.load_mesh(str(empty)) == 0


def test_plane_generator_reproduces_shipped_plane(rt, tmp_path):
    p = tmp_path / "plane.obj"
    rt.make_plane_obj(str(p), 10)
    data = p.read_bytes()
    want = open(os.path.join(GOLDEN, "SUMMARY.txt")).read().split()[1]
    assert hashlib.md5(data).hexdigest() == want        # md5 of the reference's Parallel/plane.obj
    assert data.count(b"\nf ") == 20402 and data.count(b"v ") == 10404 and data.count(b"vt ") == 1


def test_plane_generator_spu1(rt, O, tmp_path):
    p = tmp_path / "p1.obj"
    rt.make_plane_obj(str(p), 1)
    s = O.OracleScene()
    n = s.add_obj(str(p))
    assert n == 2 * 11 * 11                            # -5..6 in steps of 1: the stepper overshoots by one step
    t = s.triangles()
    assert t[:, 1::3].min() == t[:, 1::3].max() == -1.0


def test_ppm_writer_matches_reference_quantisation(rt, O, tmp_path):
    rng = np.random.default_rng(7)
    img = (rng.random((5, 7, 3), dtype=np.float32) * 300).astype(np.float32)
    img[0, 0] = [np.nan, np.inf, 0.0]
    img[1, 1] = [255.0, 254.99998, 51.0]
    p = tmp_path / "a.ppm"
    rt.write_ppm(str(p), img)
    data = p.read_bytes()
    assert data.startswith(b"P6\n7 5\n255\n")
    assert np.array_equal(np.frombuffer(data[len(b"P6\n7 5\n255\n"):], np.uint8), O.quantize(img).ravel())


def test_compute_fails_loudly_without_a_gpu(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    s = rt.Scene()
    s.load_mesh(SPOT)
    with pytest.raises(rt.RtError):
        s.render(rt.default_params(16, 16))
    with pytest.raises(rt.RtError):
        s.build_grid()


def test_subdivide_is_watertight_and_deterministic(rt):
    v = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [1, 0, 0, 1, 1, 0, 0, 1, 0]], np.float32)
    a = rt.subdivide(v, 3)
    assert a.shape == (2 * 64, 9)
    assert np.array_equal(a, rt.subdivide(v, 3))
    # the shared edge (1,0,0)-(0,1,0) gets the same midpoints from both triangles
    pts = {tuple(p) for p in a.reshape(-1, 3).tolist()}
    assert len(pts) == 81
