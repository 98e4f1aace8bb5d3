import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(16, 16, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False; slv.log_every=1
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
sc = scan.clone(); pr = prb[:,None].clone(); x = torch.ones_like(psi)
for it in range(6):
    r = slv.run(data, x, sc, pr, piter=1); x = r['psi']
    st = slv._cg_state.cpu().numpy()
    print(it, "gamma_psi", st[12], "gamma0", st[14], "ncand", st[15], "ngroups", st[16], "tried", st[17], "resolved", st[18], "failed", st[19], "hints", st[20:22], slv.history[-1])
