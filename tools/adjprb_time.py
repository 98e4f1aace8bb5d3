"""Time of the probe adjoint (adj_probe) at 4096 x 256^2, per kernel."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
g = slv.fwd(psi,scan,prb)
for _ in range(3): slv.adj_probe(g, scan, psi)
torch.cuda.synchronize(); slv.profile(True)
for _ in range(10): slv.adj_probe(g, scan, psi)
torch.cuda.synchronize()
print({k: round(v[0]/v[1], 4) for k, v in slv.profile_read().items()})
