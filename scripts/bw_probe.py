import torch, time
n = 3_400_000_000 // 8
a = torch.empty(n, dtype=torch.float64, device='cuda')
b = torch.empty(n, dtype=torch.float64, device='cuda')
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms = t(lambda: a.fill_(1.5)); print("fill  %.3f ms  write %.2f TB/s" % (ms, n*8/ms/1e9))
ms = t(lambda: a.zero_()); print("zero  %.3f ms  write %.2f TB/s" % (ms, n*8/ms/1e9))
ms = t(lambda: b.copy_(a)); print("copy  %.3f ms  r+w %.2f TB/s (write %.2f)" % (ms, 2*n*8/ms/1e9, n*8/ms/1e9))
ms = t(lambda: torch.mul(a, 2.0, out=b)); print("scale %.3f ms  r+w %.2f TB/s" % (ms, 2*n*8/ms/1e9))
