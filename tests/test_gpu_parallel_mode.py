"""GPU: the second shading mode, fast_trace of the reference's Parallel tree (Parallel/raytracer.cu:445-524: material table,
nearest hit with t > eps, shadow ray towards the light, mirror reflection to depth 3) and its loader
(Parallel/raytracer.cu:805-873: scale * (vertex + offset)), against the CPU restatement in oracle/rt_oracle.c.

PARITY UNPINNED: the Parallel tree is CUDA (nvcc, thrust) and cannot be compiled where these tests run, and the reference
ships no image of it, so neither side of this comparison has ever been checked against the reference's own output. What
the tests do pin is that the HIP path and an independent, recursive CPU restatement of the same source lines agree: hit
ids bit for bit in full-walk mode, colours within 1/255 of the 0..255 scale on >= 99.9 % of the pixels (the specular term
goes through powf, which differs in the last bits between the two maths libraries)."""
import numpy as np
import pytest

from conftest import BLUB, SPOT, bits

pytestmark = pytest.mark.gpu

# the material table of Parallel/raytracer.cu:450-453: base colour, kd, ks, spec_alpha, ka, reflective, km
TABLE = [((0, 0, 255), 1.0, 1.5, 1.25, 0.3, True, 0.6), ((255, 0, 0), 10.0, 10.0, 1.25, 0.3, False, 0.0),
         ((0, 20, 0), 10.0, 10.0, 1.25, 0.3, True, 0.9999), ((255, 0, 0), 10.0, 10.0, 1.25, 0.3, False, 0.0)]
# camera, light and constants of Parallel/raytracer.cu:470,531-534,505 and Parallel/geometry.cuh:3
PAR = dict(cam_pos=(18.0, 18.0, 19.0), fov=60.0, light_pos=(2.0, 5.0, 0.0), eps=1e-4, shadow_scale=0.5)


def _mat(mod, k):
    c, kd, ks, a, ka, refl, km = TABLE[k]
    return mod.Material(c, kd, ks, a, ka, km, refl)


def _main_scene_of_the_parallel_tree(rt, O, plane10):
    """main() of Parallel/raytracer.cu:777-780: plane x3, blub, spot, blub, scale 5, material indices 0, 1, 1, 3."""
    s, o = rt.Scene(), O.OracleScene()
    for path, k, off, scale in [(plane10, 0, (0, 0, 0), 3), (BLUB, 1, (0, 0, 0), 5), (SPOT, 1, (1.5, 0, 0), 5), (BLUB, 3, (-1.5, 0.2, 0.5), 5)]:
        n1 = s.load_mesh_material(path, True, _mat(rt, k), off, scale)
        n2 = o.add_obj_material(path, True, _mat(O, k), off, scale)
        assert n1 == n2 > 0
    assert np.array_equal(bits(s.triangles()), bits(o.triangles()))       # scale * (vertex + offset), rounded once
    return s, o


def _compare(rgb, ids, o_rgb, o_ids, exact_ids):
    n_id = int((ids != o_ids).sum())
    if exact_ids:
        assert n_id == 0, f"{n_id} hit ids differ"
    a = np.minimum(np.nan_to_num(rgb.astype(np.float64), nan=1e9), 255.0)
    b = np.minimum(np.nan_to_num(o_rgb.astype(np.float64), nan=1e9), 255.0)
    agree = float((np.abs(a - b).max(axis=-1) <= 1.0).mean())
    assert agree >= 0.999, f"only {agree:.6f} of the pixels within 1/255"
    rel = np.abs(rgb.astype(np.float64) - o_rgb) / np.maximum(np.abs(o_rgb.astype(np.float64)), 1.0)
    assert float(np.nanpercentile(rel, 99.9)) < 1e-3
    return n_id


@pytest.mark.parametrize("depth", [3, 0, 1])
def test_parallel_tree_scene_against_its_restatement(rt, O, plane10, depth):
    s, o = _main_scene_of_the_parallel_tree(rt, O, plane10)
    W, H = 320, 240
    o_rgb, o_ids, _ = o.render_parallel(O.default_params(W, H, walk_mode=0, reflect_depth=depth, **PAR))
    assert (o_ids >= 0).mean() > 0.3
    rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL, shading_mode=rt.RT_SHADE_PARALLEL, reflect_depth=depth, **PAR))
    _compare(rgb, ids, o_rgb, o_ids, True)
    rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_EARLY, shading_mode=rt.RT_SHADE_PARALLEL, reflect_depth=depth, **PAR))
    assert _compare(rgb, ids, o_rgb, o_ids, False) <= 2
    if depth == 3:
        # the reflections are really there: the mirror plane's pixels change with the depth
        o0, _, _ = o.render_parallel(O.default_params(W, H, walk_mode=0, reflect_depth=0, **PAR))
        assert (np.abs(o0 - o_rgb).max(axis=-1) > 1.0).mean() > 0.05


def test_reflective_meshes_seen_from_the_serial_camera(rt, O):
    """Both meshes mirrors (km 0.5 and 0.9), camera and light of the Serial tree, eps as small as the Parallel tree's:
    rays bounce between the two meshes, reflected rays start on a surface and must not hit it again below eps."""
    s, o = rt.Scene(), O.OracleScene()
    for mod, sc in ((rt, s), (O, o)):
        add = sc.load_mesh_material if mod is rt else sc.add_obj_material
        add(SPOT, True, mod.Material((200, 180, 40), 0.02, 0.5, 8.0, 0.2, 0.5, True), (0, 0, 0), 1)
        add(BLUB, True, mod.Material((30, 120, 255), 0.05, 0.1, 2.0, 0.1, 0.9, True), (1.5, 0, 0), 1)
    kw = dict(eps=1e-4, shadow_scale=0.5, light_pos=(5.0, -5.0, 2.0))
    W, H = 256, 256
    o_rgb, o_ids, _ = o.render_parallel(O.default_params(W, H, walk_mode=0, **kw))
    rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL, shading_mode=rt.RT_SHADE_PARALLEL, **kw))
    assert (o_ids >= 0).sum() > 5000
    _compare(rgb, ids, o_rgb, o_ids, True)


def test_serial_mode_ignores_materials_and_bad_modes_are_rejected(rt, plane10):
    from conftest import golden
    s = rt.Scene()
    s.load_mesh_material(SPOT, True, rt.Material((255, 0, 0), 1, 2, 3, 4, 0.5, True))
    s.load_mesh_material(BLUB, True, rt.Material((255, 0, 0), 9, 9, 9, 9, 0.9, True), (1.5, 0, 0), 1)
    rgb, _, ids = s.render(rt.default_params(512, 512))
    g_ids, g_rgb = golden("main_512x512")
    assert np.array_equal(ids, g_ids) and np.array_equal(bits(rgb), bits(g_rgb))
    with pytest.raises(rt.RtError, match="shading_mode"):
        s.render(rt.default_params(8, 8, shading_mode=7))
    with pytest.raises(rt.RtError, match="reflect_depth"):
        s.render(rt.default_params(8, 8, shading_mode=rt.RT_SHADE_PARALLEL, reflect_depth=4))
