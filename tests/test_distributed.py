"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): model-blob broadcast, candidate
sharding, per-rank structures.  The kernels themselves need a GPU (tests/test_gpu_parity.py)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, os.environ["TWR_ROOT"])
import numpy as np, torch, torch.distributed as dist
import towr_amd as ta
from towr_amd import sweep
from towr_amd.dist import broadcast_model, broadcast_grid, my_shard, gather_scores, best_candidate
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
model = broadcast_model(ta.model_preset("anymal", "stairs") if rank == 0 else None)
ref = ta.model_preset("anymal", "stairs")
assert bytes(model) == bytes(ref), "model blob differs after broadcast"
# the gridded terrain travels the same way (SURVEY 5 / 8e): a float elevation layer + resolution + position, then CSV heights
elev = (np.arange(40 * 28, dtype=np.float32).reshape(40, 28) % 17) * 0.01
gm = broadcast_grid(ta.GridMap(elev, 0.05, (0.7, -0.1)) if rank == 0 else None)
assert isinstance(gm, ta.GridMap) and gm.elevation.shape == (40, 28) and np.array_equal(np.asarray(gm.elevation), elev)
assert gm.resolution == 0.05 and gm.position == (0.7, -0.1)
csv = broadcast_grid(ta.TerrainGrid(np.arange(6 * 9, dtype=np.float64).reshape(6, 9) * 0.02) if rank == 0 else None)
assert not isinstance(csv, ta.GridMap) and np.array_equal(csv.heights, np.arange(54, dtype=np.float64).reshape(6, 9) * 0.02)
mg = ta.model_preset("anymal", "grid_map")
gs = sweep.candidate_structures(mg, sweep.enumerate_candidates(6), threads=2, grid=gm)   # every rank, on ITS copy of the map
hg = [None] * world
dist.all_gather_object(hg, hashlib.sha256(b"".join(s_.col_idx.tobytes() for s_ in gs)).digest())
assert all(x == hg[0] for x in hg)
cands = sweep.enumerate_candidates(48)
# SURVEY 8e: shard by a cheap weight, then every rank builds the structures of ITS shard only (threaded library call)
weights = [int(sweep.candidate_weight(c)) for c in cands]
a, b = my_shard(weights, rank, world)
structs_mine = sweep.candidate_structures(model, cands[a:b], threads=2)
assert len(structs_mine) == b - a
structs = {a + i: s for i, s in enumerate(structs_mine)}
if 7 not in structs:
    structs[7] = sweep.candidate_structure(model, cands[7])   # (one common candidate for the cross-rank pattern check)
ref7 = sweep.candidate_structure(model, cands[a])
assert np.array_equal(ref7.col_idx, structs[a].col_idx) and ref7.nnz == structs[a].nnz   # threaded == one by one
mine = torch.tensor([a, b, sum(weights[a:b])], dtype=torch.int64)
allr = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allr, mine)
allr = [t.tolist() for t in allr]
assert allr[0][0] == 0 and allr[-1][1] == len(cands)
for r in range(world - 1):
    assert allr[r][1] == allr[r + 1][0], "shards must tile the candidate list"
assert sum(t[2] for t in allr) == sum(weights)
assert max(t[2] for t in allr) - min(t[2] for t in allr) <= 2 * max(weights)
# every rank derives the identical pattern for the same candidate (no rank-dependent state)
h = hashlib.sha256(structs[7].col_idx.tobytes() + structs[7].row_ptr.tobytes()).digest()
hs = [None] * world
dist.all_gather_object(hs, h)
assert all(x == hs[0] for x in hs)
# planner-style exchange: every rank scores its shard, one all-gather gives everyone the table in candidate order
sizes = [t[1] - t[0] for t in allr]
local = torch.stack([torch.full((16,), float(c), dtype=torch.float64) for c in range(a, b)])
local[:, 2] = torch.tensor([abs(c - 29) + 0.25 for c in range(a, b)], dtype=torch.float64)   # "dynamic" inf-norm: best is 29
if a <= 5 < b:
    local[5 - a, 2] = float("nan")                                                          # a failed candidate never wins
table = gather_scores(local, sizes)
assert table.shape == (len(cands), 16)
assert torch.equal(table[:, 0], torch.arange(len(cands), dtype=torch.float64))
best, score = best_candidate(table, families=(1,))
assert best == 29 and score == 0.25
oks = [None] * world
dist.all_gather_object(oks, (rank, int(a), int(b)))
dist.barrier()
if rank == 0:
    sys.stdout.write("ALL RANKS OK %s\n" % sorted(oks))   # one writer: no interleaved output
    sys.stdout.flush()
dist.destroy_process_group()
'''


def test_two_rank_gloo_broadcast_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TWR_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL RANKS OK [(0, 0, " in out.stdout and "(1, " in out.stdout, out.stdout
