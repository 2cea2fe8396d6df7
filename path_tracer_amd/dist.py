"""Multi-GPU sharding of the pixel loop: one process per GPU, rows dealt to ranks in strips, no data-path collective
while rendering (pixels are independent, src/main.rs:181-207), one gather of the final framebuffer to rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def rows_of_rank(height: int, rank: int, world_size: int, strip_rows: int) -> np.ndarray:
    """Global rows rendered by `rank` (same rule as libptmi's compute_rows / global_row)."""
    y = np.arange(height, dtype=np.int64)
    return y[(y // strip_rows) % world_size == rank]


def max_rows(height: int, world_size: int, strip_rows: int) -> int:
    return max(len(rows_of_rank(height, r, world_size, strip_rows)) for r in range(world_size))


class _DevPtr:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can view libptmi's framebuffer in place."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def wrap_device_framebuffer(ptr: int, rows: int, width: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevPtr(ptr, (rows, width, 4)), device=device)


class _Plan:
    """What a gather of one frame geometry needs again and again: the padded send buffer, ONE receive buffer whose chunks are the
    gather list, and the row permutation that de-interleaves it with a single index_select (per call: no allocation, no host-to-device
    copy; the first version rebuilt and uploaded one index tensor per rank per call, ~0.5 ms on a 10 ms step at 8 ranks)."""

    def __init__(self, height, width, world_size, strip_rows, dtype, device, is_dst):
        self.pad_rows = max_rows(height, world_size, strip_rows)
        self.send = None  # allocated on first use by ranks that have to pad
        if is_dst:
            self.recv = torch.empty((world_size * self.pad_rows, width, 4), dtype=dtype, device=device)
            self.parts = list(self.recv.view(world_size, self.pad_rows, width, 4).unbind(0))
            perm = np.empty(height, dtype=np.int64)
            for r in range(world_size):
                rows = rows_of_rank(height, r, world_size, strip_rows)
                perm[rows] = r * self.pad_rows + np.arange(len(rows))
            self.perm = torch.from_numpy(perm).to(device)


_plans: dict = {}


def _plan(height, width, world_size, strip_rows, dtype, device, is_dst) -> _Plan:
    key = (height, width, world_size, strip_rows, dtype, str(device), is_dst)
    pl = _plans.get(key)
    if pl is None:
        pl = _plans[key] = _Plan(height, width, world_size, strip_rows, dtype, device, is_dst)
    return pl


def pad_strips(local: torch.Tensor, height: int, width: int, world_size: int, strip_rows: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """A rank's (local_rows, width, 4) strips padded with zero rows to the largest rank's row count: gather wants equal chunks.
    `out` (pad_rows, width, 4), if given, is reused: its padding rows are zero from the first call on."""
    pad_rows = max_rows(height, world_size, strip_rows)
    if local.shape[0] == pad_rows:
        return local.contiguous()
    send = out if out is not None else torch.zeros((pad_rows, width, 4), dtype=local.dtype, device=local.device)
    send[: local.shape[0]] = local
    return send


def assemble_strips(parts, height: int, width: int, world_size: int, strip_rows: int) -> torch.Tensor:
    """De-interleave the gathered (padded) strip buffers of all ranks into the (height, width, 4) frame, on their device."""
    full = torch.empty((height, width, 4), dtype=parts[0].dtype, device=parts[0].device)
    for r in range(world_size):
        rows = torch.from_numpy(rows_of_rank(height, r, world_size, strip_rows)).to(parts[0].device)
        full[rows] = parts[r][: len(rows)]
    return full


def gather_framebuffer(local: torch.Tensor, height: int, width: int, rank: int, world_size: int, strip_rows: int, dst: int = 0):
    """Gather the per-rank (local_rows, width, 4) accumulation buffers on `dst` and de-interleave into (height, width, 4).
    Returns the full framebuffer on dst and None elsewhere.  Equal-size chunks: ranks with fewer rows pad.  One collective (into one
    receive buffer) and one index_select per call; buffers and the row permutation are kept per frame geometry."""
    if world_size == 1:
        return local
    pl = _plan(height, width, world_size, strip_rows, local.dtype, local.device, rank == dst)
    if local.shape[0] != pl.pad_rows and pl.send is None:
        pl.send = torch.zeros((pl.pad_rows, width, 4), dtype=local.dtype, device=local.device)
    send = pad_strips(local, height, width, world_size, strip_rows, out=pl.send)
    if rank == dst:
        dist.gather(send, pl.parts, dst=dst)
        return pl.recv.index_select(0, pl.perm)
    dist.gather(send, None, dst=dst)
    return None
