"""Where a planner step of the sweep goes (values only -> twr_batch_score -> twr_batch_best), piece by piece, event-free:
each piece alone back to back on a stream, the three together, and the three replayed from a hipGraph."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
cands = sweep.enumerate_candidates(max(sizes))
structs_all = sweep.candidate_structures(model, cands)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B in sizes:
    structs = structs_all[:B]
    batch = ta.Batch(structs, list(range(B)), device=0)
    xh = np.concatenate([perturbed_inputs(s, model, 1, i)[0] for i, s in enumerate(structs)])
    x = torch.from_numpy(xh).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    scores = torch.empty((B, 16), dtype=torch.float64, device="cuda")
    best = torch.zeros(2, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ev = lambda: batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
    sc = lambda: batch.score_device(g.data_ptr(), scores.data_ptr(), st)
    be = lambda: batch.best_device(scores.data_ptr(), B, best.data_ptr(), stream=st)

    def all3():
        ev(); sc(); be()

    def fused2():
        ev()
        batch.score_best_device(g.data_ptr(), scores.data_ptr(), best.data_ptr(), stream=st)

    t_ev, t_sc, t_be, t_all = timed(ev), timed(sc), timed(be), timed(all3)
    t_f2 = timed(fused2)
    t_both = timed(lambda: batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s2 = side.cuda_stream
        for _ in range(3):
            batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, s2)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(10):
                cs = torch.cuda.current_stream().cuda_stream
                batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, cs)
                batch.score_device(g.data_ptr(), scores.data_ptr(), cs)
                batch.best_device(scores.data_ptr(), B, best.data_ptr(), stream=cs)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            graph.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 300 * 1e6
    print("B=%4d  values-only eval %.1f us, score %.1f us, best %.1f us, the three %.1f us, from a graph %.1f us, values + score_best %.1f us;  values + Jacobian %.1f us"
          % (B, t_ev, t_sc, t_be, t_all, t_graph, t_f2, t_both), flush=True)
