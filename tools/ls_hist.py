"""Histogram of the accepted line-search candidate index over a 50-iteration run (config 2)."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch, collections
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
hist=[]
orig = slv._fused_line_search
def wrapped(*a, **k):
    r = orig(*a, **k); hist.append(r); return r
slv._fused_line_search = wrapped
rp = len(sys.argv) > 1
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=50, recover_prb=rp); torch.cuda.synchronize()
idx=[(-np.log2(h) if h>0 else -1) for h in hist]
print(collections.Counter(int(round(i)) for i in idx))
print([int(round(i)) for i in idx])
