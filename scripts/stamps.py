import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import towr_amd as ta
from tests.common import baseline_cases
case = baseline_cases()["C3_anymal_trot_K200"](); S = case.S
B=4096
batch = ta.Batch([S],[0]*B)
base = np.stack([case.x_perturbed(i) for i in range(32)])
x = torch.from_numpy(np.tile(base,(B//32,1)).reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device='cuda'); j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
L = ta.lib()
for _ in range(3): batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), 3, st)
torch.cuda.synchronize()
buf = np.zeros(2048*8, dtype=np.uint64)
L.twr_debug_stamps(buf.ctypes.data_as(C.c_void_p), 1)
batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), 3, st); torch.cuda.synchronize()
L.twr_debug_stamps(buf.ctypes.data_as(C.c_void_p), 0)
b = buf.reshape(2048,8).astype(np.float64); b = b[b[:,4]>0]
n = b[:,4].sum()
names=["x-load wait","front","copy-out(prev)","back"]
tot = b[:,:4].sum()
for k in range(4): print("%-16s %8.0f cycles/slice  %5.1f%%" % (names[k], b[:,k].sum()/n, 100*b[:,k].sum()/tot))
clk = b[:,7].mean()/2**20*100e6
print("in-kernel clock %.3f GHz" % (clk/1e9))
print("sum %.0f cycles/slice over %d waves, %.1f slices/wave" % (tot/n, len(b), n/len(b)))
