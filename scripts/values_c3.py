"""The values-only leg of bench.py alone (C3, 8192 problems: per-kernel events and event-free), for A/B work on the
values-only kernels.  usage: values_c3.py [repeats]"""
import sys
sys.path.insert(0, ".")
import torch
import towr_amd as ta
import bench

model = ta.model_preset("anymal", "flat")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    v = bench.values_c3(ta, torch, model, dev, 0, stream)
    print("values_c3 %.1f M cb/s  %.4f ms with events, %.4f without  %s" % (
        v["value"] / 1e6, v["ms_per_step"], v["ms_per_step_without_events"], {k.split("::")[1]: round(t, 4) for k, t in v["kernel_ms"].items()}), flush=True)
