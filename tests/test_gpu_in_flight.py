"""GPU: several frames of one scene in flight (rt_scene_set_frames_in_flight, rt_streams_create): frames issued back to
back on different streams, each on its own working set, must give bit for bit what the same frames give one at a time."""
import numpy as np
import pytest

from conftest import BLUB, SPOT

pytestmark = pytest.mark.gpu


def _views(rt, n):
    cams = [(3.0, 5.0, 3.0), (2.0, 3.0, 4.0), (-3.0, 4.0, 2.5), (1.0, 6.0, -3.0), (4.0, 2.0, 0.5), (0.5, 5.5, 4.0)]
    sizes = [(640, 480), (333, 217), (512, 512), (800, 600), (256, 320), (640, 360)]
    return [rt.default_params(*sizes[k % 6], cam_pos=cams[k % 6], walk_mode=rt.RT_WALK_FULL if k % 2 else rt.RT_WALK_EARLY) for k in range(n)]


def _buffers(rt, torch, p):
    rows, W = rt.part_rows(p), p.width
    return (torch.full((rows, W, 3), -1.0, dtype=torch.float32, device="cuda"), torch.full((rows, W, 3), 9, dtype=torch.uint8, device="cuda"),
            torch.full((rows, W), -9, dtype=torch.int32, device="cuda"))


def test_streams_run_side_by_side(rt):
    hs = rt.create_streams(3)
    try:
        assert 1 <= len(hs) <= 3 and len(set(hs)) == len(hs)
        # the default configuration of the runtime has four hardware queues: three streams beside each other exist
        assert len(hs) == 3
    finally:
        rt.destroy_streams(hs)
    with pytest.raises(rt.RtError):
        rt.create_streams(99)


@pytest.mark.parametrize("depth,rebuild", [(3, True), (2, False), (4, True)])
def test_frames_in_flight_equal_frames_one_at_a_time(rt, depth, rebuild):
    import torch
    s = rt.Scene(); s.load_mesh(SPOT); s.load_mesh(BLUB, True, (10, 200, 30), (1.5, 0, 0))
    s.upload()
    views = _views(rt, 9)
    want = []
    for p in views:                                  # one working set, one stream, one frame at a time
        b = _buffers(rt, torch, p)
        s.render_device(p, rebuild_grid=True, d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr())
        torch.cuda.synchronize()
        want.append([t.cpu().numpy() for t in b])
    s.set_frames_in_flight(depth)
    assert s.frames_in_flight() == depth
    hs = rt.create_streams(depth)
    try:
        bufs = [_buffers(rt, torch, p) for p in views]
        torch.cuda.synchronize()                     # the fills above ran on torch's stream
        for k, p in enumerate(views):                # nine frames back to back, no host wait in between
            b = bufs[k]
            s.render_device(p, rebuild_grid=rebuild, d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr(), stream=hs[k % len(hs)])
        torch.cuda.synchronize()
        for k in range(len(views)):
            rgb, u8, ids = [t.cpu().numpy() for t in bufs[k]]
            assert (ids >= 0).sum() > 500
            assert np.array_equal(ids, want[k][2]), k
            assert np.array_equal(rgb.view(np.uint32), want[k][0].view(np.uint32)), k
            assert np.array_equal(u8, want[k][1]), k
        st = s.stats_at(0)
        assert st["ms_render"] > 0 and st["n_refs"] > 0
        # more streams than sets: the frame on the extra stream waits for a set and is still right
        b = _buffers(rt, torch, views[0]); torch.cuda.synchronize()
        extra = torch.cuda.Stream()
        s.render_device(views[0], rebuild_grid=True, d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr(), stream=extra.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(b[2].cpu().numpy(), want[0][2])
        # back to one set
        s.set_frames_in_flight(1)
        b = _buffers(rt, torch, views[1]); torch.cuda.synchronize()
        s.render_device(views[1], d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(b[1].cpu().numpy(), want[1][1])
    finally:
        torch.cuda.synchronize()
        rt.destroy_streams(hs)
    with pytest.raises(rt.RtError):
        s.set_frames_in_flight(0)


def test_crowded_grid_in_flight(rt):
    """A mesh of 909,312 triangles (crowded voxels: run culling, the kernel variant without the small-voxel path, the unit
    sizes and the wavefront share chosen for frames in flight) from two cameras, four frames in flight against one at a time."""
    import torch
    tmp = rt.Scene(); tmp.load_mesh(BLUB); v9 = rt.subdivide(tmp.triangles(), 3); tmp.close()
    s = rt.Scene(); s.add_triangles(v9); s.upload()
    views = [rt.default_params(1920, 1080), rt.default_params(1280, 720, cam_pos=(1.0, 1.6, 1.0), cam_target=(0.0, 0.2, -0.45)),
             rt.default_params(1920, 1080, walk_mode=rt.RT_WALK_FULL), rt.default_params(800, 600, cam_pos=(-2.0, 3.0, 2.0))]
    want = []
    for p in views:
        b = _buffers(rt, torch, p)
        s.render_device(p, rebuild_grid=True, d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr())
        torch.cuda.synchronize()
        want.append([t.cpu().numpy() for t in b])
    s.set_frames_in_flight(4)
    hs = rt.create_streams(4)
    try:
        order = [0, 1, 2, 3, 1, 0, 3, 2, 2, 1]
        bufs = [_buffers(rt, torch, views[v]) for v in order]
        torch.cuda.synchronize()
        for k, v in enumerate(order):
            b = bufs[k]
            s.render_device(views[v], rebuild_grid=True, d_rgb=b[0].data_ptr(), d_u8=b[1].data_ptr(), d_ids=b[2].data_ptr(), stream=hs[k % len(hs)])
        torch.cuda.synchronize()
        for k, v in enumerate(order):
            rgb, u8, ids = [t.cpu().numpy() for t in bufs[k]]
            assert (ids >= 0).sum() > 5000
            assert np.array_equal(ids, want[v][2]), k
            assert np.array_equal(rgb.view(np.uint32), want[v][0].view(np.uint32)), k
            assert np.array_equal(u8, want[v][1]), k
    finally:
        torch.cuda.synchronize()
        rt.destroy_streams(hs)


@pytest.mark.gpu
def test_cost_estimate_is_reused_only_for_the_same_view_and_changes_no_pixel(rt):
    """A frame with the camera, light, frame size and triangles of the last estimated frame on its working set skips the
    tile cost estimate; any change of the view or of the triangle list estimates again; the images are those of a scene
    that estimates every frame (the estimate only orders and sizes the work units)."""
    import numpy as np
    from conftest import SPOT, BLUB
    views = [dict(), dict(), dict(cam_pos=(2.5, 4.0, 3.5)), dict(cam_pos=(2.5, 4.0, 3.5)), dict(), dict(light_pos=(4.0, -5.0, 2.0)), dict(light_pos=(4.0, -5.0, 2.0))]
    fresh = [True, False, True, False, True, True, False]
    a = rt.Scene(); a.load_mesh(SPOT); a.load_mesh(BLUB, True, (255, 0, 0), (1.5, 0, 0))
    b = rt.Scene(); b.load_mesh(SPOT); b.load_mesh(BLUB, True, (255, 0, 0), (1.5, 0, 0))
    b.set_estimate_reuse(False)
    for v, f in zip(views, fresh):
        before = a.estimates_run()
        p = rt.default_params(320, 240, **v)
        ra, ua, ia = a.render(p, want_u8=True)
        rb, ub, ib = b.render(p, want_u8=True)
        assert a.estimates_run() - before == (1 if f else 0), (v, f)
        assert np.array_equal(ia, ib) and np.array_equal(ra.view(np.uint32), rb.view(np.uint32)) and np.array_equal(ua, ub)
    assert b.estimates_run() == len(views)
    # another frame size, and a changed triangle list, estimate again
    n0 = a.estimates_run()
    a.render(rt.default_params(336, 240))
    a.render(rt.default_params(336, 240))
    assert a.estimates_run() == n0 + 1
    a.load_mesh(SPOT, True, (0, 255, 0), (-1.5, 0, 0))
    a.render(rt.default_params(336, 240))
    assert a.estimates_run() == n0 + 2
