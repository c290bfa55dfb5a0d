"""CPU, only where oracle/_ref/libref.so exists (built from /root/reference/Serial by oracle/Makefile):
the C restatement against the reference itself, beyond what the committed golden vectors cover."""
import numpy as np
import pytest

from conftest import BLUB, SPOT, bits

import oracle_py

pytestmark = pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref/libref.so not built")


def _pair(O, meshes):
    r, o = O.RefScene(), O.OracleScene()
    for path, vt, col, off in meshes:
        assert r.add_obj(path, vt, col, off) == o.add_obj(path, vt, col, off)
    return r, o


def test_loader_and_grid_bit_exact(O):
    r, o = _pair(O, [(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (255, 0, 0), (1.5, 0, 0))])
    assert np.array_equal(bits(r.triangles()), bits(o.triangles()))
    r.build_grid(); o.build_grid()
    gr, go = r.grid_info(), o.grid_info()
    assert np.array_equal(gr["n"], go["n"])
    for k in ("lohi", "width", "inv_width"):
        assert np.array_equal(bits(gr[k]), bits(go[k]))
    sr, tr = r.grid_csr(); so, to = o.grid_csr()
    assert np.array_equal(sr, so) and np.array_equal(tr, to)


def test_offset_rounding_in_double(O):
    # (float)(double + (double)offset): differs from float + float for these offsets
    r, o = _pair(O, [(SPOT, True, (255, 0, 0), (0.1, -0.7, 1e-3))])
    assert np.array_equal(bits(r.triangles()), bits(o.triangles()))


@pytest.mark.parametrize("W,H,cam", [(160, 120, (3, 5, 3, 0, 0, 0, 0, -1, 0, 45)),
                                     (97, 33, (0.2, 0.1, -0.3, 1, 0, 0, 0, -1, 0, 80)),
                                     (64, 64, (-2, 1, 4, 0.5, 0, 0, 0.2, -1, 0.1, 30))])
def test_render_bit_exact(O, W, H, cam):
    r, o = _pair(O, [(SPOT, True, (255, 0, 0), (0, 0, 0)), (BLUB, True, (20, 30, 240), (1.5, 0, 0))])
    rgb_r, ids_r, _ = r.render(W, H, cam=cam)
    p = O.default_params(W, H, cam_pos=cam[0:3], cam_target=cam[3:6], cam_up=cam[6:9], fov=float(cam[9]))
    rgb_o, ids_o, _ = o.render(p)
    assert np.array_equal(ids_r, ids_o)
    assert np.array_equal(bits(rgb_r), bits(rgb_o))
    assert np.array_equal(O.quantize(rgb_r), O.quantize(rgb_o))
