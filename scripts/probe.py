import numpy as np, time, sys
sys.path.insert(0,'.')
import towr_amd as ta
from tests.common import baseline_cases, row_scale
for name,mk in baseline_cases().items():
    case=mk(); goal = 2.0 if case.terrain!="flat" else 1.0
    xs=[case.x_perturbed(0,goal), case.x_wild(0)]
    b=ta.Batch([case.S],[0]*len(xs)); g,j=b.eval_host(np.concatenate(xs))
    for p,x in enumerate(xs):
        rg,_,_,rj=case.P.eval(x)
        gd=g[b.g_off[p]:b.g_off[p+1]]; jd=j[b.jac_off[p]:b.jac_off[p+1]]
        rs=row_scale(case.S.row_ptr,rj)
        rel=np.abs(jd-rj)/np.maximum(np.abs(rj),1e-300)
        mask=np.abs(rj)>1e-6*rs
        print(name,p,"g maxerr/maxg %.2e"%(np.abs(gd-rg).max()/np.abs(rg).max()),"J max err/rowscale %.2e"%(np.abs(jd-rj)/np.maximum(rs,1e-300)).max(),"J max pure rel (|ref|>1e-6 rowscale) %.2e"%rel[mask].max(), "nonzero J", np.count_nonzero(jd), "of", jd.size)
import torch
case=baseline_cases()["C3_anymal_trot_K200"]()
B=4096
b=ta.Batch([case.S],[0]*B)
x=torch.tensor(np.concatenate([case.x_perturbed(i) for i in range(64)]*(B//64)),device='cuda')
g=torch.empty(int(b.g_off[-1]),dtype=torch.float64,device='cuda'); j=torch.empty(int(b.jac_off[-1]),dtype=torch.float64,device='cuda')
st=torch.cuda.current_stream().cuda_stream
for fl in (3,1,2):
    for _ in range(3): b.eval_device(x.data_ptr(),g.data_ptr(),j.data_ptr(),fl,st)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): b.eval_device(x.data_ptr(),g.data_ptr(),j.data_ptr(),fl,st)
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print("flags",fl,"B",B,"ms",dt*1e3,"callbacks/s %.3e"%(B/dt),"alg GB/s %.1f"%(b.algorithmic_bytes/dt/1e9))
