import sys; sys.path.insert(0,'.')
import numpy as np, torch
import towr_amd as ta
from tests.common import baseline_cases
case = baseline_cases()["C3_anymal_trot_K200"](); S = case.S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = ta.Batch([S],[0]*B)
base = np.stack([case.x_perturbed(i) for i in range(32)])
xh = np.tile(base,(B//32,1)); x = torch.from_numpy(xh.reshape(-1)).cuda()
g = torch.full((int(batch.g_off[-1]),), float('nan'), dtype=torch.float64, device='cuda')
j = torch.full((int(batch.jac_off[-1]),), float('nan'), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    g.fill_(float('nan')); j.fill_(float('nan'))
    batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), 3, st); torch.cuda.synchronize()
    G = g.view(B,S.m); J = j.view(B,S.nnz)
    dG = (G.view(B//32,32,S.m) != G[:32]); dJ = (J.view(B//32,32,S.nnz) != J[:32])
    print("rep",rep,"nan g",int(torch.isnan(g).sum()),"nan j",int(torch.isnan(j).sum()),"G mismatches",int(dG.sum()),"J mismatches",int(dJ.sum()))
    if dG.any():
        idx = dG.nonzero()[:10].cpu().numpy(); print(" G idx (rep,p,row):", idx.tolist())
        rows = dG.any(dim=0).any(dim=0).nonzero().flatten().cpu().numpy(); print(" rows affected", rows[:40], len(rows))
        for s in S.con_sets: print("  ", s["name"], s["offset"], s["size"])
        r,p,row = idx[0]; print(" values", G[r*32+p,row].item(), G[p,row].item())
    if dJ.any():
        idx = dJ.nonzero()[:10].cpu().numpy(); print(" J idx:", idx.tolist())
        cols = dJ.any(dim=0).any(dim=0).nonzero().flatten().cpu().numpy(); print(" nz affected", cols[:40], len(cols))
        r,p,c = idx[0]; print(" values", J[r*32+p,c].item(), J[p,c].item())
