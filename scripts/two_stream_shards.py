"""a shard evaluated as ONE batch against the same candidates as two half-batches alternating on two streams
(one's tail over the other's head).  Event-free step time of values + Jacobian, 128 / 256 / 512 candidates."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
structs_all = sweep.candidate_structures(model, sweep.enumerate_candidates(max(sizes)))


def make(structs, first):
    batch = ta.Batch(structs, list(range(len(structs))), device=0)
    xh = np.concatenate([perturbed_inputs(s, model, 1, first + i)[0] for i, s in enumerate(structs)])
    x = torch.from_numpy(xh).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    return batch, x, g, j


def run(parts, streams, n=400):
    def step():
        for (b, x, g, j), s in zip(parts, streams):
            b.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, s.cuda_stream)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B in sizes:
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    whole = make(structs_all[:B], 0)
    halves = [make(structs_all[:B // 2], 0), make(structs_all[B // 2:B], B // 2)]
    t_one = run([whole], [s1])
    t_two = run(halves, [s1, s2])
    t_two_same = run(halves, [s1, s1])
    nbytes = whole[0].algorithmic_bytes
    print("B=%4d  one batch %.1f us (%.2f TB/s)   two halves on two streams %.1f us (%.2f TB/s)   two halves on one stream %.1f us"
          % (B, t_one, nbytes / t_one / 1e6, t_two, nbytes / t_two / 1e6, t_two_same), flush=True)
