"""Which allocations of the 1024-candidate sweep's buffers evaluate fast (177 us) and which slow (194 us): a list of ballast
sizes (GB) held while the buffers are allocated and freed again afterwards, in the order given; prints the step time and the
device addresses of the Jacobian buffer and of the ballast.  usage: placement_probe5.py 0 2 2 0 2 4 ..."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
STEPS = int(os.environ.get("PROBE_STEPS", "200"))   # (under rocprofv3 --pmc: a few)
B = 1024
cands = sweep.enumerate_candidates(B)
structs = sweep.candidate_structures(model, cands)
batch = ta.Batch(structs, list(range(B)), device=0)
xh = np.concatenate([perturbed_inputs(s, model, 1, i)[0] for i, s in enumerate(structs)])
st = torch.cuda.current_stream().cuda_stream
dev = torch.device("cuda", 0)
torch.empty(1 << 28, dtype=torch.float64, device=dev).fill_(1.0)   # power state
for gb in [float(a) for a in sys.argv[1:]] or [0, 2, 2, 0, 2, 4, 1, 2]:
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb > 0 else None
    x = torch.from_numpy(xh).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    bp = ballast.data_ptr() if ballast is not None else 0
    del ballast
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    print("ballast %4.1f GB @ %#x: %.1f us/step   jac @ %#x (%.0f MB)  g @ %#x  x @ %#x" % (
        gb, bp, (time.perf_counter() - t0) / STEPS * 1e6, j.data_ptr(), j.numel() * 8 / 1e6, g.data_ptr(), x.data_ptr()), flush=True)
    del x, g, j
    torch.cuda.empty_cache()
