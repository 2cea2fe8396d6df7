#!/usr/bin/env python3
"""Rehearsal of bench.py's N>1 flow (sharded render + gather) with every rank on cuda:0 (one-GPU box).  RCCL refuses two
ranks on one device, so this uses gloo for the collective and stages the framebuffer through the host: it checks the
sharding / gather / assembly logic end to end against a single-rank render, not RCCL itself."""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
from path_tracer_amd import dist as ptdist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W, H, SPP, strip = 320, 180, 8, 4
r = api.Renderer(scenes.cornell_box(W, H), W, H, max_bounces=8, rank=rank, world_size=world, strip_rows=strip, device=0)
r.render_device(0, SPP)
ptr, _ = r.accum_device_ptr()
fb = ptdist.wrap_device_framebuffer(ptr, len(r.local_rows()), W, torch.device("cuda", 0))
torch.cuda.synchronize()
full = ptdist.gather_framebuffer(fb.cpu(), H, W, rank, world, strip, dst=0)
if rank == 0:
    ref = api.Renderer(scenes.cornell_box(W, H), W, H, max_bounces=8, device=0).render(0, SPP)[0]
    ok = np.array_equal(full.numpy().view(np.uint32), ref.view(np.uint32))
    print("sharded == single:", ok, flush=True)
    assert ok
dist.barrier()
dist.destroy_process_group()
