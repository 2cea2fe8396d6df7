"""CPU: the N>1 path (row sharding + one gather of the framebuffer) with the gloo backend, world_size 2 and 3.
The per-rank framebuffers are the rows of one oracle frame (no GPU here); the test checks that sharding, padding,
gather and de-interleave reproduce the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, strip, frame_path, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from path_tracer_amd import api
    from path_tracer_amd import scenes
    from path_tracer_amd.dist import gather_framebuffer, rows_of_rank
    frame = np.load(frame_path)
    # libptmi's own row assignment for this rank (host logic, no GPU) must agree with the gather's
    r = api.Renderer(scenes.cornell_box(width, height), width, height, rank=rank, world_size=world, strip_rows=strip)
    rows = r.local_rows()
    assert np.array_equal(rows, rows_of_rank(height, rank, world, strip))
    local = torch.from_numpy(frame[rows].copy())
    full = gather_framebuffer(local, height, width, rank, world, strip, dst=0)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,strip", [(2, 30, 4), (3, 37, 4), (2, 8, 16)])
def test_gather_reassembles_the_frame(tmp_path, oracle_mod, world, height, strip):
    from path_tracer_amd import scenes
    width = 40
    o = oracle_mod.Oracle(scenes.cornell_box(width, height))
    frame, _, _, _ = o.render(width, height, 2, max_bounces=3)
    fp, op = str(tmp_path / "frame.npy"), str(tmp_path / "out.npy")
    np.save(fp, frame)
    mp.spawn(_worker, args=(world, _free_port(), height, width, strip, fp, op), nprocs=world, join=True)
    out = np.load(op)
    assert np.array_equal(out.view(np.uint32), frame.view(np.uint32))
