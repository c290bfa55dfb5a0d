"""CPU, world_size 2 over gloo: the N > 1 path of bench.py without a GPU. Every rank produces the bands the
partition assigns to it (here with the oracle standing in for the render kernel, as the checker), rank 0
gathers them with the same collective bench.py uses and puts the bands back in frame order; the result must be
the single-process frame. Covers rt_part_rows(), the band layout and the gather plumbing."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SPOT, bits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    rt = importlib.import_module("ray-tracer_amd")
    import bench
    import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = rt.default_params(W, H, part_index=rank, part_count=world)
        rows = rt.part_rows(p)
        rowmap = rt.part_row_map(p)
        o = O.OracleScene()
        o.add_obj(SPOT)
        o.build_grid()
        op = O.default_params(W, H, walk_mode=1)
        part = np.zeros((rows, W, 3), np.float32)
        for local, y in enumerate(rowmap):
            if y >= 0:
                part[local] = o.render(op, y0=int(y), y1=int(y) + 1, want_ids=False)[0][0]
        gathered = bench.gather_parts(torch.from_numpy(part), world, rank)
        if rank == 0:
            frame = rt.deinterleave_host(p, gathered.numpy())
            np.save(out_path, frame)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(64, 61), (40, 16)])
def test_two_rank_band_partition_and_gather(O, tmp_path, W, H):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, out), nprocs=2, join=True)
    frame = np.load(out)
    o = O.OracleScene()
    o.add_obj(SPOT)
    want = o.render(O.default_params(W, H, walk_mode=1), want_ids=False)[0]
    assert np.array_equal(bits(frame), bits(want))


def test_row_map_matches_part_rows(rt):
    for H in (1, 7, 8, 9, 61, 2160):
        for n in (1, 2, 3, 4, 8):
            seen = []
            for k in range(n):
                p = rt.default_params(16, H, part_index=k, part_count=n)
                m = rt.part_row_map(p)
                assert len(m) == rt.part_rows(p)
                seen += [int(y) for y in m if y >= 0]
            assert sorted(seen) == list(range(H))      # every frame row belongs to exactly one part


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher: main() must start the N ranks as a child process under
    torch.distributed.run (never replace itself), pass its own arguments on and exit with the child's code, before torch or
    the GPU is touched."""
    import subprocess
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    cmd = bench.launch_command(2, ["--gpus", "2", "--steps", "5"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"]
    seen = {}

    def fake_run(c, env=None, **kw):
        seen["cmd"], seen["env"] = c, env
        return subprocess.CompletedProcess(c, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(os, "execv", lambda *a: (_ for _ in ()).throw(AssertionError("exec is not allowed")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "4"
    assert seen["cmd"][-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE set) nothing is started: main() goes on to its own rank's work
    monkeypatch.setenv("WORLD_SIZE", "4")
    seen.clear()
    # (stopped at its first GPU call, so that no process group is waited for on a box that has a GPU)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a: (_ for _ in ()).throw(RuntimeError("reached the rank's own work")))
    try:
        bench.main()
    except BaseException:
        pass
    assert "cmd" not in seen
