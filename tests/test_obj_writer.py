"""CPU: the OBJ writer of the test infrastructure (oracle_py.write_obj), which bench.py uses to hand the benchmarked scene
to the compiled reference: what load_mesh (Serial/raytracer.cpp:220-287, format_has_vt = false) reads back is the same floats."""
import numpy as np
import pytest

from conftest import BLUB, bits

import oracle_py


def _soup(O, seed=0):
    o = O.OracleScene()
    o.add_obj(BLUB)
    t = o.triangles()
    rng = np.random.default_rng(seed)
    extra = rng.standard_normal((500, 9)).astype(np.float32) * np.float32(1e3)
    extra[::7, 2] = 0.0                                         # zeros, tiny and large magnitudes among the coordinates
    extra[::11, 4] = np.float32(1.17549435e-38)
    extra[::13, 5] = np.float32(3.4e38)
    return np.ascontiguousarray(np.concatenate([t, extra]), dtype=np.float32)


def test_written_obj_reads_back_bit_for_bit_in_the_oracle_loader(O, tmp_path):
    v9 = _soup(O)
    path = str(tmp_path / "soup.obj")
    nv, nf = O.write_obj(path, v9)
    assert nf == v9.shape[0] and nv < 3 * nf                    # shared vertices are written once
    o = O.OracleScene()
    assert o.add_obj(path, False) == v9.shape[0]
    assert np.array_equal(bits(o.triangles()), bits(v9))


@pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref/libref.so not built")
def test_written_obj_reads_back_bit_for_bit_in_the_reference_loader(O, tmp_path):
    v9 = _soup(O, 1)
    path = str(tmp_path / "soup.obj")
    O.write_obj(path, v9)
    r = O.RefScene()
    assert r.add_obj(path, False) == v9.shape[0]
    assert np.array_equal(bits(r.triangles()), bits(v9))
