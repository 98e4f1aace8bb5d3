import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch, ctypes
import libtike.hipfft as pt
from libtike.hipfft import _native as nat
from libtike.hipfft import synthetic as syn
from libtike.hipfft.ptycho import _ptr, _stream
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
slv._cg_fwd_cols(0, psi, scan, prb); slv._cg_fwd_cols(1, psi*0.9, scan, prb)
costs = torch.zeros(17, dtype=torch.float64, device='cuda'); sums = torch.zeros(2, dtype=torch.float64, device='cuda'); cost=torch.zeros(1,dtype=torch.float64,device='cuda')
def T(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for nc in (1,4,8,16):
    print("linesearch ncand",nc, T(lambda: nat.check(nat.cg_linesearch(slv._h,0,1,_ptr(data),None,1.0,nc,_ptr(costs),_stream()))))
print("stats", T(lambda: nat.check(nat.cg_stats(slv._h,0,_ptr(data),_ptr(sums),_stream()))))
sums[0]=1.0; sums[1]=1.0
print("project", T(lambda: nat.check(nat.cg_project(slv._h,0,1,_ptr(data),_ptr(sums),_ptr(cost),_stream()))))
print("fwd_cols", T(lambda: slv._cg_fwd_cols(1, psi, scan, prb)))
