"""CPU: the oracle (C restatement) against the golden vectors generated from the compiled reference."""
import numpy as np
import pytest

from conftest import CASES, bits, cam_kwargs, golden, load_case


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_full_walk_matches_golden(O, plane10, name):
    s = O.OracleScene()
    W, H, cam = load_case(lambda p, vt, c, o: s.add_obj(p, vt, c, o), name, plane10)
    rgb, ids, _ = s.render(O.default_params(W, H, walk_mode=0, **cam_kwargs(cam)))
    g_ids, g_rgb = golden(name)
    assert np.array_equal(ids, g_ids)                  # hit ids: bit-exact
    assert np.array_equal(bits(rgb), bits(g_rgb))      # float colours: bit-exact


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_early_exit_matches_golden(O, plane10, name):
    """Early exit is the GPU default; on these scenes it changes nothing."""
    s = O.OracleScene()
    W, H, cam = load_case(lambda p, vt, c, o: s.add_obj(p, vt, c, o), name, plane10)
    cnt_full, cnt_early = O.Counters(), O.Counters()
    s.render(O.default_params(W, H, walk_mode=0, **cam_kwargs(cam)), want_rgb=False, want_ids=False, counters=cnt_full)
    rgb, ids, _ = s.render(O.default_params(W, H, walk_mode=1, **cam_kwargs(cam)), counters=cnt_early)
    g_ids, g_rgb = golden(name)
    assert np.array_equal(ids, g_ids)
    assert np.array_equal(bits(rgb), bits(g_rgb))
    a, b = cnt_full.as_dict(), cnt_early.as_dict()
    for k in ("primary_rays", "primary_entering", "hits", "shadow_rays", "shadow_entering", "shadowed"):
        assert a[k] == b[k]
    assert b["tests_primary"] <= a["tests_primary"] and b["tests_shadow"] <= a["tests_shadow"]


def test_quantize_edge_cases(O):
    x = np.array([0.0, 254.999, 255.0, 256.0, 1e30, np.inf, np.nan, 51.0, 50.999996, 0.5], np.float32)
    q = O.quantize(x)
    assert list(q) == [0, 254, 255, 255, 255, 255, 255, 51, 50, 0]


def test_empty_scene_is_black(O):
    s = O.OracleScene()
    assert s.add_obj("/nonexistent/file.obj") == 0
    rgb, ids, _ = s.render(O.default_params(16, 8))
    assert (ids == -1).all() and (rgb == 0).all()
