"""Per-kernel time of the CG loop (in-library HIP events), native stage loop vs host-driven fused loop."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = syn.make_problem(R, 64 if R == 64 else 32, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
for native in (True, False):
    slv.native = native
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=3); torch.cuda.synchronize()
    slv.profile(True)
    n = 20
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=n); torch.cuda.synchronize()
    prof = slv.profile_read(); slv.profile(False)
    tot = sum(v[0] for v in prof.values())
    print("native" if native else "host-driven", p['nscan'], "positions: sum of kernels %.3f ms/iter" % (tot / n))
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("   %-28s %7.3f ms/iter  %5.1f launches/iter  %.4f ms each" % (k, v[0] / n, v[1] / n, v[0] / v[1]))
