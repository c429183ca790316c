"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in CPU tests).  The path shards by independent candidates, so the only collective is
ONE broadcast of the POD model blob; afterwards every rank works on its own contiguous shard."""
import ctypes

import numpy as np

from . import Model
from .sweep import shard_bounds


def broadcast_model(model, src=0, device=None):
    """Rank `src` passes a towr_amd.Model, the others pass None; everyone returns the same Model."""
    import torch
    import torch.distributed as dist

    nbytes = ctypes.sizeof(Model)
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(bytes(model)), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    out = Model()
    ctypes.memmove(ctypes.addressof(out), buf.cpu().numpy().tobytes(), nbytes)
    return out


def my_shard(weights, rank, world):
    b = shard_bounds(weights, world)
    return b[rank], b[rank + 1]
