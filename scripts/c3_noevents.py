"""Step time of the C3 workload (8192 problems by default) WITHOUT per-kernel events, i.e. with whatever launch structure
launch_eval picks (fused launch or three kernels): for A/B of launch-structure knobs on a tuning build
(TWR_FUSED_MAX_ROM / TWR_FUSED_SPLIT).  Usage: python scripts/c3_noevents.py [problems]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
x = torch.from_numpy(np.tile(base, (B // 256 + 1, 1))[:B].reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, torch.device("cuda", 0), 0.5)
for _ in range(10):
    batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print("B=%d  %.4f ms/step  %.3f M cb/s  %.2f TB/s" % (B, ms, B / ms / 1e3, batch.algorithmic_bytes / ms / 1e9), flush=True)
