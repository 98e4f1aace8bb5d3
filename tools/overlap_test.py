"""Do a compute-bound column pass and a memory-bound row pass overlap when issued on two streams?"""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
A = pt.PtychoCuFFT(4096,256,256,1,768,768); B = pt.PtychoCuFFT(4096,256,256,1,768,768)
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
scanB = scan.clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
y = A.fwd(psi,scan,prb)
def both(kind):
    with torch.cuda.stream(s1):
        r1 = A.fwd(psi,scan,prb) if kind=="fwd" else A.adj(y,scan,prb)
    with torch.cuda.stream(s2):
        r2 = B.fwd(psi,scanB,prb) if kind=="fwd" else B.adj(y,scanB,prb)
    return r1, r2
def seq(kind):
    r1 = A.fwd(psi,scan,prb) if kind=="fwd" else A.adj(y,scan,prb)
    r2 = B.fwd(psi,scanB,prb) if kind=="fwd" else B.adj(y,scanB,prb)
    return r1, r2
def T(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for kind in ("fwd","adj"):
    print(kind, "sequential 2x", T(lambda: seq(kind)), "two streams 2x", T(lambda: both(kind)))
