import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
R1, R2 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
p = syn.make_problem(R1, R2, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=20); torch.cuda.synchronize()
