"""Small-batch step time of the ragged sweep (no per-kernel events): 32 -> 1024 candidates on one GPU,
launched call by call and replayed from a captured hipGraph."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 512, 1024]
cands = sweep.enumerate_candidates(max(sizes))
structs_all = sweep.candidate_structures(model, cands)
for B in sizes:
    structs = structs_all[:B]
    batch = ta.Batch(structs, list(range(B)), device=0)
    tb = batch.table_bytes()
    print("B=%4d  tables: %.1f MB resident, dyn layout %.1f MB built -> %.1f MB distinct" % (B, tb["resident"] / 1e6, tb["dyn_layout"] / 1e6, tb["dyn_layout_distinct"] / 1e6), flush=True)
    xh = np.concatenate([perturbed_inputs(s, model, 1, i)[0] for i, s in enumerate(structs)])
    x = torch.from_numpy(xh).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    nbytes = 8 * (x.numel() + g.numel() + j.numel())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = side.cuda_stream
        for _ in range(30):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
        side.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
        side.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(10):
                batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n // 10):
            graph.replay()
        torch.cuda.synchronize()
        usg = (time.perf_counter() - t0) / (n // 10 * 10) * 1e6
    print("B=%4d  %.1f us/step (%.2f M cb/s, %.2f TB/s)   graph of 10: %.1f us/step (%.2f M cb/s, %.2f TB/s)"
          % (B, us, B / us, nbytes / us / 1e6, usg, B / usg, nbytes / usg / 1e6), flush=True)
