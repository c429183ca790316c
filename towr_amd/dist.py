"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in CPU tests).  The path shards by independent candidates, so the only collective on the evaluation
path is ONE broadcast of the POD model blob; afterwards every rank works on its own contiguous shard.  A planner-style
sweep that wants one decision from all shards adds an all-gather of the per-candidate scores (16 doubles per
candidate, twr_batch_score) -- SURVEY section 8e."""
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


def broadcast_grid(grid, src=0, device=None):
    """The gridded terrain of a sweep ("W x H x 4 B if gridded"): rank `src` passes a
    towr_amd.TerrainGrid (CSV heights) or towr_amd.GridMap (the float elevation layer + resolution + position fpowr
    hands the solver), the others pass None; everyone returns an equal handle.  Two broadcasts beside the model blob: a
    48-byte header (kind, sizes, resolution, position) and the cell data."""
    import torch
    import torch.distributed as dist

    from . import GridMap, TerrainGrid

    hdr = torch.zeros(6, dtype=torch.float64, device=device)
    if dist.get_rank() == src:
        if isinstance(grid, GridMap):
            hdr.copy_(torch.tensor([1.0, grid.elevation.shape[0], grid.elevation.shape[1], grid.resolution, grid.position[0],
                                    grid.position[1]], dtype=torch.float64))
        else:
            hdr.copy_(torch.tensor([0.0, grid.heights.shape[0], grid.heights.shape[1], 0.0, 0.0, 0.0], dtype=torch.float64))
    dist.broadcast(hdr, src=src)
    kind, n0, n1, res, px, py = [float(v) for v in hdr.cpu()]
    n0, n1 = int(n0), int(n1)
    data = torch.zeros(n0 * n1, dtype=torch.float32 if kind == 1.0 else torch.float64, device=device)
    if dist.get_rank() == src:
        # grid_map's layer is column-major [size_x][size_y]: ship it in storage order
        flat = grid.elevation.reshape(-1, order="F") if kind == 1.0 else grid.heights.reshape(-1)
        data.copy_(torch.from_numpy(np.ascontiguousarray(flat)))
    dist.broadcast(data, src=src)
    if dist.get_rank() == src:
        return grid
    a = data.cpu().numpy()
    if kind == 1.0:
        return GridMap(a.reshape((n0, n1), order="F"), res, (px, py))
    return TerrainGrid(a.reshape(n0, n1))


def my_shard(weights, rank, world):
    b = shard_bounds(weights, world)
    return b[rank], b[rank + 1]


def gather_scores(local, shard_sizes):
    """All-gather of the per-candidate score rows: `local` is this rank's (n_local, k) tensor (twr_batch_score gives
    k = 16), shard_sizes the candidates per rank in rank order; every rank returns the (sum(shard_sizes), k) table in
    candidate order.  One collective of equal blocks (shards padded to the largest)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    assert len(shard_sizes) == world and local.shape[0] == shard_sizes[rank]
    k, n_max = local.shape[1], max(shard_sizes)
    block = torch.zeros((n_max, k), dtype=local.dtype, device=local.device)
    block[:local.shape[0]] = local
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    return torch.cat([blocks[r][:shard_sizes[r]] for r in range(world)], dim=0)


def best_candidate(table, families=(0, 1, 3, 4)):
    """Index and score of the candidate with the smallest summed inf-norm violation over the given constraint
    families (bit indices of TWR_SET_*: 0 terrain, 1 dynamic, 3 rangeofmotion, 4 force); NaN scores lose."""
    import torch

    total = sum(table[:, 2 * f] for f in families)
    total = torch.where(torch.isnan(total), torch.full_like(total, float("inf")), total)
    idx = int(torch.argmin(total))
    return idx, float(total[idx])


def gather_best(local_best):
    """All-gather of every rank's 16-byte decision (twr_batch_score_best with index_offset = the shard's first candidate:
    [global candidate index, total]); returns the winner (index, total) on every rank -- the smallest total, on a tie the
    smallest index (ranks hold ascending candidate ranges, so the first minimum is it); a NaN total loses."""
    import torch
    import torch.distributed as dist

    rows = [torch.empty_like(local_best) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, local_best)
    table = torch.stack(rows)
    total = torch.where(torch.isnan(table[:, 1]), torch.full_like(table[:, 1], float("inf")), table[:, 1])
    r = int(torch.argmin(total))
    return int(table[r, 0]), float(total[r])
