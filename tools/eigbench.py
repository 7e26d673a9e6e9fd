import time, numpy as np, scipy.linalg as sla, torch, os
from threadpoolctl import threadpool_info, threadpool_limits
print([ (p['internal_api'], p['num_threads']) for p in threadpool_info()])
def t(f, n=5):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
for N in (240, 510, 1250):
    rng=np.random.default_rng(0); A=rng.standard_normal((N,N)); A=A+A.T
    res={}
    for nt in (1, 8, 16, 32, 128):
        with threadpool_limits(limits=nt):
            res['np.eigh t%d'%nt]=t(lambda: np.linalg.eigh(A), 3)
            res['scipy evr t%d'%nt]=t(lambda: sla.eigh(A, driver='evr'), 3)
            res['scipy evd t%d'%nt]=t(lambda: sla.eigh(A, driver='evd'), 3)
    At=torch.from_numpy(A)
    res['torch cpu eigh']=t(lambda: torch.linalg.eigh(At), 3)
    Ag=At.cuda()
    def g(): torch.linalg.eigh(Ag); torch.cuda.synchronize()
    res['torch gpu eigh']=t(g, 5)
    print(N, {k: round(v,2) for k,v in res.items()})
