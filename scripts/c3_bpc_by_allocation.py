"""C3, 8192 problems, ONE process, several allocations of the buffers: rom_kernel / dyn_kernel with fewer persistent workgroups
per CU (TUNING build: TWR_ROM_BPC / TWR_DYN_BPC are read on every launch) -- does a slow allocation want fewer concurrent store
streams than a fast one?"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

dev = torch.device("cuda", 0)
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)


def run(x, g, j, rom_bpc, dyn_bpc):
    os.environ["TWR_ROM_BPC"], os.environ["TWR_DYN_BPC"] = str(rom_bpc), str(dyn_bpc)
    for _ in range(3):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    batch.profile_begin(15)
    torch.cuda.synchronize()
    for _ in range(15):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    k, _ = batch.profile_end()
    return k


for gb in [float(a) for a in sys.argv[1:]] or [0, 2, 5, 10, 1, 3, 7, 14]:
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb > 0 else None
    x = torch.from_numpy(xh).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    run(x, g, j, 4, 8)
    rows = []
    for rb, db in ((4, 8), (3, 8), (2, 8), (4, 6), (4, 8)):
        k = run(x, g, j, rb, db)
        rows.append("rom%d/dyn%d: %.3f + %.3f" % (rb, db, k["rangeofmotion"], k["dynamic"]))
    print("ballast %4.1f GB  " % gb + "   ".join(rows), flush=True)
    del x, g, j
    torch.cuda.empty_cache()
