"""GPU: rt_render_device_noalloc (the entry point that allocates nothing: the frame's working memory is a fixed device
array inside the library) against rt_render_device on the same scenes, through the C ABI. Two calls in a row, the
second with a smaller scene and a smaller frame: the array keeps what the first call left in it."""
import numpy as np
import pytest

from conftest import BLUB, SPOT

pytestmark = pytest.mark.gpu


def _frame(rt, torch, s, p, noalloc):
    rows, W = rt.part_rows(p), p.width
    rgb = torch.full((rows, W, 3), -1.0, dtype=torch.float32, device="cuda")
    u8 = torch.full((rows, W, 3), 9, dtype=torch.uint8, device="cuda")
    ids = torch.full((rows, W), -9, dtype=torch.int32, device="cuda")
    if noalloc:
        s.render_device_noalloc(p, d_rgb=rgb.data_ptr(), d_u8=u8.data_ptr(), d_ids=ids.data_ptr())
    else:
        s.render_device(p, rebuild_grid=True, d_rgb=rgb.data_ptr(), d_u8=u8.data_ptr(), d_ids=ids.data_ptr())
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), u8.cpu().numpy(), ids.cpu().numpy()


def test_noalloc_entry_point_matches_rt_render_device(rt):
    import torch
    big = rt.Scene(); big.load_mesh(SPOT); big.load_mesh(BLUB, True, (10, 200, 30), (1.5, 0, 0))
    small = rt.Scene(); small.load_mesh(SPOT, True, (0, 90, 255))
    big.upload(); small.upload()
    calls = [(big, rt.default_params(1024, 768)),
             (small, rt.default_params(333, 217, cam_pos=(2.0, 3.0, 4.0), walk_mode=rt.RT_WALK_FULL)),      # smaller input: stale contents must not show
             (big, rt.default_params(640, 480, part_index=1, part_count=3))]
    got = [_frame(rt, torch, s, p, True) for s, p in calls]          # the no-alloc calls first, back to back
    for (s, p), (rgb, u8, ids) in zip(calls, got):
        w_rgb, w_u8, w_ids = _frame(rt, torch, s, p, False)
        assert (w_ids >= 0).sum() > 500
        assert np.array_equal(ids, w_ids)
        assert np.array_equal(rgb.view(np.uint32), w_rgb.view(np.uint32))
        assert np.array_equal(u8, w_u8)


def test_noalloc_entry_point_rejects_what_does_not_fit(rt):
    import torch
    s = rt.Scene(); s.load_mesh(SPOT)
    out = torch.empty((8, 8, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(rt.RtError, match="rt_scene_upload"):
        s.render_device_noalloc(rt.default_params(8, 8), d_u8=out.data_ptr())
    s.upload()
    with pytest.raises(rt.RtError, match="tiles"):
        s.render_device_noalloc(rt.default_params(3840, 2160), d_u8=out.data_ptr())
    tmp = rt.Scene(); tmp.load_mesh(BLUB); v9 = rt.subdivide(tmp.triangles(), 3); tmp.close()
    crowded = rt.Scene(); crowded.add_triangles(v9); crowded.upload()
    with pytest.raises(rt.RtError, match="references"):
        crowded.render_device_noalloc(rt.default_params(8, 8), d_u8=out.data_ptr())


def test_noalloc_entry_point_under_graph_capture(rt):
    """What the entry point is for: a frame captured into a HIP graph (no allocation may happen under capture) and replayed,
    also after the outputs were overwritten, gives the frame of the plain call."""
    import torch
    s = rt.Scene(); s.load_mesh(SPOT); s.load_mesh(BLUB, True, (10, 200, 30), (1.5, 0, 0))
    s.upload()
    p = rt.default_params(640, 480)
    w_rgb, w_u8, w_ids = _frame(rt, torch, s, p, False)
    rgb = torch.full((480, 640, 3), -1.0, dtype=torch.float32, device="cuda")
    u8 = torch.full((480, 640, 3), 9, dtype=torch.uint8, device="cuda")
    ids = torch.full((480, 640), -9, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        # the scene's first call of the entry point is the captured one: rt_scene_upload() has looked the array's address up
        g.capture_begin()
        s.render_device_noalloc(p, d_rgb=rgb.data_ptr(), d_u8=u8.data_ptr(), d_ids=ids.data_ptr(), stream=st.cuda_stream)
        g.capture_end()
    for _ in range(2):
        rgb.fill_(-1.0); u8.fill_(9); ids.fill_(-9)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(ids.cpu().numpy(), w_ids)
        assert np.array_equal(rgb.cpu().numpy().view(np.uint32), w_rgb.view(np.uint32))
        assert np.array_equal(u8.cpu().numpy(), w_u8)
