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
from towr_amd.dist import broadcast_model, broadcast_grid, my_shard, gather_scores, best_candidate, gather_best
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
# SURVEY 8e: shard by BYTES per callback -- every rank computes the same exact list without building device tables
# (twr_candidate_bytes) -- then every rank builds the structures of ITS shard only (threaded library call)
weights = [int(w) for w in sweep.candidate_bytes(model, cands, threads=2)]
hw = [None] * world
dist.all_gather_object(hw, weights)
assert all(x == hw[0] for x in hw), "the ranks must cut the shards from identical weights"
a, b = my_shard(weights, rank, world)
structs_mine = sweep.candidate_structures(model, cands[a:b], threads=2)
assert len(structs_mine) == b - a
structs = {a + i: s for i, s in enumerate(structs_mine)}
if 7 not in structs:
    structs[7] = sweep.candidate_structure(model, cands[7])   # (one common candidate for the cross-rank pattern check)
ref7 = sweep.candidate_structure(model, cands[a])
assert np.array_equal(ref7.col_idx, structs[a].col_idx) and ref7.nnz == structs[a].nnz   # threaded == one by one
assert weights[a] == ref7.algorithmic_bytes                                             # the weight IS 8 (n + m + nnz)
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
# the lighter exchange: every rank's own 16-byte decision (what twr_batch_score_best leaves on the device: global index,
# total), all-gathered -- the same winner on every rank; an all-NaN shard loses
mine_best = best_candidate(local, families=(1,))
winner = gather_best(torch.tensor([a + mine_best[0], mine_best[1]], dtype=torch.float64))
assert winner == (29, 0.25), winner
nan_row = torch.tensor([float(a), float("nan")], dtype=torch.float64) if rank == 0 else torch.tensor([a + 1.0, 7.5], dtype=torch.float64)
w2 = gather_best(nan_row)
assert w2[1] == 7.5 and w2[0] >= 1, w2
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


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["TWR_ROOT"])
import numpy as np, torch, torch.distributed as dist
import towr_amd as ta
from towr_amd import sweep
from towr_amd.dist import broadcast_model, broadcast_grid, gather_scores, best_candidate
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)            # exactly what bench.py does on a multi-GPU node
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
model = broadcast_model(ta.model_preset("anymal", "stairs"), src=0, device=dev)
assert bytes(model) == bytes(ta.model_preset("anymal", "stairs"))
elev = (np.arange(24 * 18, dtype=np.float32).reshape(24, 18) % 11) * 0.01
gm = broadcast_grid(ta.GridMap(elev, 0.05, (0.7, -0.1)), src=0, device=dev)
assert np.array_equal(np.asarray(gm.elevation), elev)
cands = sweep.enumerate_candidates(12)
structs = sweep.candidate_structures(model, cands, threads=2)
batch = ta.Batch(structs, list(range(len(structs))), device=0)
ee = [[model.nominal_stance[e][0], model.nominal_stance[e][1], 0.0] for e in range(model.n_ee)]
z = -model.nominal_stance[0][2]
x = torch.from_numpy(np.concatenate([s.initial_guess([0, 0, z], [0, 0, 0], [2.0, 0, z], [0, 0, 0], ee) for s in structs])).to(dev)
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
scores = torch.empty((len(structs), 16), dtype=torch.float64, device=dev)
batch.score_device(g.data_ptr(), scores.data_ptr(), st)
table = gather_scores(scores, [len(structs)])           # device tensors through RCCL's all-gather
assert table.is_cuda and torch.equal(table, scores)
from towr_amd.dist import gather_best
best_d = torch.zeros(2, dtype=torch.float64, device=dev)
batch.score_best_device(g.data_ptr(), scores.data_ptr(), best_d.data_ptr(), stream=st)
assert gather_best(best_d) == best_candidate(table)     # the 16-byte decision through RCCL
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
best = best_candidate(table)[0]
assert 0 <= best < len(structs)
dist.destroy_process_group()
print("RCCL_OK", best)
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_backend_single_rank(tmp_path):
    """The collectives of towr_amd.dist and bench.py on DEVICE tensors through the RCCL backend (world size 1 -- the
    pool's boxes have one GPU): a helper that hands RCCL a host tensor, or an init that the ROCm build rejects, fails
    here instead of on the 8-GPU node."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TWR_ROOT=ROOT, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
