import sys; sys.path.insert(0,'.')
import numpy as np, torch
import towr_amd as ta
from tests.common import baseline_cases
case = baseline_cases()["C3_anymal_trot_K200"](); S = case.S
B=4096
batch = ta.Batch([S],[0]*B)
base = np.stack([case.x_perturbed(i) for i in range(32)])
x = torch.from_numpy(np.tile(base,(B//32,1)).reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device='cuda'); j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for fl in (3,2,1):
    for _ in range(3): batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), fl, st)
    batch.profile_begin(20)
    for _ in range(20): batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), fl, st)
    ms,n = batch.profile_end()
    print("flags",fl, {k:"%.1f us"%(v*1e3) for k,v in ms.items()})
