"""GPU: the exchange step behind the C ABI, rt_comm_create + rt_gather_bands (single process, RCCL, ncclCommInitAll /
ncclGather), against the frame rendered in one piece. The GPU box of the test run has one GPU: there the communicator
has one device and the gather is a copy onto itself; with more GPUs visible every device renders its part."""
import numpy as np
import pytest

from conftest import BLUB, SPOT, bits

pytestmark = pytest.mark.gpu


def _scene(rt):
    s = rt.Scene(); s.load_mesh(SPOT); s.load_mesh(BLUB, True, (255, 0, 0), (1.5, 0, 0))
    return s


def test_gather_bands_on_the_visible_gpus(rt):
    import torch
    n = min(torch.cuda.device_count(), 8)
    assert n >= 1
    W, H = 333, 217
    torch.cuda.set_device(0)
    whole = _scene(rt)
    rgb, u8, ids = whole.render(rt.default_params(W, H), want_u8=True)
    comm = rt.Comm(list(range(n)))
    scenes, parts_u8, parts_id, streams = [], [], [], []
    rows = rt.part_rows(rt.default_params(W, H, part_count=n))
    for r in range(n):
        torch.cuda.set_device(r)
        s = _scene(rt)
        p = rt.default_params(W, H, part_index=r, part_count=n)
        pu = torch.full((rows, W, 3), 3, dtype=torch.uint8, device="cuda:%d" % r)
        pi = torch.full((rows, W), -3, dtype=torch.int32, device="cuda:%d" % r)
        st = torch.cuda.current_stream(r).cuda_stream
        s.render_device(p, rebuild_grid=True, d_u8=pu.data_ptr(), d_ids=pi.data_ptr(), stream=st)
        scenes.append(s); parts_u8.append(pu); parts_id.append(pi); streams.append(st)
    torch.cuda.set_device(0)
    p = rt.default_params(W, H, part_count=n)
    stacked_u8 = torch.empty((n, rows, W, 3), dtype=torch.uint8, device="cuda:0")
    stacked_id = torch.empty((n, rows, W), dtype=torch.int32, device="cuda:0")
    frame_u8 = torch.full((H, W, 3), 9, dtype=torch.uint8, device="cuda:0")
    frame_id = torch.full((H, W), -9, dtype=torch.int32, device="cuda:0")
    for _ in range(2):                                            # twice: the communicator is reusable
        comm.gather_bands(p, [t.data_ptr() for t in parts_u8], stacked_u8.data_ptr(), frame_u8.data_ptr(), 3, streams)
        comm.gather_bands(p, [t.data_ptr() for t in parts_id], stacked_id.data_ptr(), frame_id.data_ptr(), 4, streams)
    for r in range(n):
        torch.cuda.synchronize(r)
    assert np.array_equal(frame_id.cpu().numpy(), ids)
    assert np.array_equal(frame_u8.cpu().numpy(), u8)
    comm.close()


def test_comm_rejects_bad_arguments(rt):
    import torch
    with pytest.raises(rt.RtError):
        rt.Comm([torch.cuda.device_count() + 3])
    with pytest.raises(rt.RtError):
        rt.Comm([0, 0])
    comm = rt.Comm([0])
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(rt.RtError, match="part_count"):
        comm.gather_bands(rt.default_params(8, 8, part_count=2, part_index=0), [buf.data_ptr()], buf.data_ptr(), buf.data_ptr(), 1)
    comm.close()
