import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False; slv.log_every=1
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
H = {}
for native in (True, False):
    slv.native = native; slv.history = []
    r = slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=12)
    H[native] = list(slv.history)
for a, b in zip(H[True], H[False]):
    print(a, b)
