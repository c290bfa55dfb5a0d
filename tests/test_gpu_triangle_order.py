"""GPU: a large mesh whose file lists its triangles in arbitrary order. The grid build then visits the triangles along a
space-filling curve (rt_scene_triangle_order), which no result may depend on: hit ids and colours against the oracle on the
same shuffled triangle list (Serial/grid.h:118-150 appends in file order, Serial/geometry.h:162-175 updates in that order),
the voxel lists read back in the reference's order, and both visiting orders against each other."""
import numpy as np
import pytest

from conftest import BLUB, bits

pytestmark = pytest.mark.gpu


def _shuffled(rt, k=3, seed=5):
    tmp = rt.Scene(); tmp.load_mesh(BLUB)
    v9 = rt.subdivide(tmp.triangles(), k); tmp.close()
    return np.ascontiguousarray(v9[np.random.default_rng(seed).permutation(v9.shape[0])])


def test_shuffled_mesh_is_built_in_spatial_order_with_the_oracles_results(rt, O, monkeypatch):
    v9 = _shuffled(rt)                                           # 909,312 triangles: crowded voxels, run culling, index lists
    W, H = 640, 360
    o = O.OracleScene(); o.add_triangles(v9)
    o_rgb, o_ids, _ = o.render(O.default_params(W, H, walk_mode=0))
    assert int((o_ids >= 0).sum()) > 5000
    frames = {}
    for setting in ("", "0", "1"):
        if setting:
            monkeypatch.setenv("RT_TRI_ORDER", setting)
        else:
            monkeypatch.delenv("RT_TRI_ORDER", raising=False)
        s = rt.Scene(); s.add_triangles(v9); s.upload()
        area_file, area_sorted, _ = s.record_order()
        area_spatial, spatial = s.triangle_order()
        if setting == "":
            # the file's order is incoherent: the build visits the triangles along the curve, whose runs are no wider
            assert spatial and 0 < area_spatial < 1.1 * area_file
        else:
            assert spatial == (setting == "1")
        rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL))
        assert np.array_equal(ids, o_ids), "RT_TRI_ORDER=%r: hit ids differ from the oracle" % setting
        assert np.array_equal(bits(rgb), bits(o_rgb)), "RT_TRI_ORDER=%r: colours differ from the oracle" % setting
        rgb_e, _, ids_e = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_EARLY))
        assert int((ids_e != o_ids).sum()) <= 4
        frames[setting] = (ids_e, rgb_e)
        if setting == "":
            start, tris = s.grid_csr()
            o_start, o_tris = o.grid_csr()
            assert np.array_equal(start, o_start) and np.array_equal(tris, o_tris)
        s.close()
    for setting in ("0", "1"):
        assert np.array_equal(frames[setting][0], frames[""][0]) and np.array_equal(bits(frames[setting][1]), bits(frames[""][1]))


def test_coherent_file_keeps_its_order(rt):
    tmp = rt.Scene(); tmp.load_mesh(BLUB)
    v9 = rt.subdivide(tmp.triangles(), 3); tmp.close()
    s = rt.Scene(); s.add_triangles(v9); s.upload()
    area_file, _, _ = s.record_order()
    area_spatial, spatial = s.triangle_order()
    assert area_spatial > 0 and not spatial and area_spatial >= 0.9 * area_file
    s.close()


def test_scene_spanning_triangles_beside_a_detailed_mesh(rt, O, monkeypatch):
    """A ground quad under a detailed mesh: two triangles whose boxes cover thousands of voxels among 909,312 that cover a
    few. The thread-per-triangle build kernels hand such a triangle to the whole wavefront; lists, hit ids and colours must be
    what Serial/grid.h:118-150 and Serial/geometry.h:162-175 give, in every list layout and in the sorted build."""
    tmp = rt.Scene(); tmp.load_mesh(BLUB)
    v9 = rt.subdivide(tmp.triangles(), 3); tmp.close()
    p = v9.reshape(-1, 3)
    lo, hi = p.min(0), p.max(0)
    y = hi[1] + 0.01                                             # the camera's up is (0,-1,0): the ground is at +y
    a, b, c, d = (lo[0] - 1, y, lo[2] - 1), (hi[0] + 1, y, lo[2] - 1), (hi[0] + 1, y, hi[2] + 1), (lo[0] - 1, y, hi[2] + 1)
    quad = np.array([a + b + c, a + c + d], dtype=np.float32)
    # the quad in the middle of the file: both its neighbours in a wavefront are small triangles
    n = v9.shape[0] // 2 + 17
    v9 = np.ascontiguousarray(np.concatenate([v9[:n], quad, v9[n:]]))
    W, H = 640, 360
    o = O.OracleScene(); o.add_triangles(v9)
    o_rgb, o_ids, _ = o.render(O.default_params(W, H, walk_mode=0))
    assert int(np.isin(o_ids, (n, n + 1)).sum()) > 20000         # the ground is in view
    o_start, o_tris = o.grid_csr()
    assert int(np.isin(o_tris, (n, n + 1)).sum()) > 1000         # and its triangles are in thousands of voxel lists
    for env in ({}, {"RT_INDEXED_RECS": "0"}, {"RT_INDEXED_RECS": "1"}, {"RT_SORTED_BUILD": "1"}):
        for k in ("RT_INDEXED_RECS", "RT_SORTED_BUILD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = rt.Scene(); s.add_triangles(v9); s.upload()
        rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL))
        assert np.array_equal(ids, o_ids), "%r: hit ids differ from the oracle" % env
        assert np.array_equal(bits(rgb), bits(o_rgb)), "%r: colours differ from the oracle" % env
        _, _, ids_e = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_EARLY))
        assert int((ids_e != o_ids).sum()) <= 4
        start, tris = s.grid_csr()
        assert np.array_equal(start, o_start) and np.array_equal(tris, o_tris), "%r: voxel lists differ" % env
        s.close()
