"""What makes the 1024-candidate sweep slower per callback than 1024 problems of one structure?  Per-kernel times of
  A  one candidate's structure shared by 1024 problems          (tables L2 resident)
  B  1024 separately built structures of that SAME candidate     (identical content, every problem reads its own copy)
  C  the enumerated sweep                                        (ragged: every candidate its own content)
for a trot and a walk candidate.  Usage: python scripts/sweep_factors.py [n=1024]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = ta.model_preset("anymal", "stairs")
cands = sweep.enumerate_candidates(1024)


def run(structs, order, label):
    batch = ta.Batch(structs, order, device=0)
    xs = {}
    xh = []
    for p, si in enumerate(order):
        if si not in xs:
            xs[si] = perturbed_inputs(structs[si], model, 4, si)
        xh.append(xs[si][p % 4])
    x = torch.from_numpy(np.concatenate(xh)).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    batch.profile_begin(30)
    for _ in range(30):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms, _ = batch.profile_end()
    print("%-58s dyn %.4f  rom %.4f  nodes %.4f ms   (%.1f MB)" % (label, ms["dynamic"], ms["rangeofmotion"], ms["nodes"],
                                                                  batch.algorithmic_bytes / 1e6), flush=True)


for rep in range(2):
    for name, idx in (("trot (combo 1, T 1.8)", 300), ("walk (combo 0, T 1.4)", 30)):
        S = sweep.candidate_structure(model, cands[idx])
        run([S], [0] * n, "A shared structure, %s" % name)
        many = sweep.candidate_structures(model, [cands[idx]] * n)
        run(many, list(range(n)), "B own copy of the same structure, %s" % name)
    run(sweep.candidate_structures(model, cands[:n]), list(range(n)), "C enumerated sweep")
