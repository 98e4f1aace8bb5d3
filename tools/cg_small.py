"""CG iteration time at 4096 / 512 / 64 positions of 256^2 (512 = one GPU's share of configs[1] under 8-way strong
scaling, 64 = the host-side floor): native stage loop and the host-driven fused loop."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
for (R1, R2) in ((64, 64), (16, 32), (8, 8)):
    p = syn.make_problem(R1, R2, 8, 256, 256, seed=1234)
    D=lambda x: torch.as_tensor(x,device='cuda')
    slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
    psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
    data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
    for name, native in (("native", True), ("host-driven", False)):
        for rec in (False, True):
            slv.native = native
            slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=6, recover_prb=rec); torch.cuda.synchronize()
            t=time.perf_counter()
            slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=50, recover_prb=rec); torch.cuda.synchronize()
            dt=(time.perf_counter()-t)/50
            print(p['nscan'], "positions %-13s recover_prb=%-5s: %.3f ms/iter, %.1f it/s" % (name, rec, dt*1e3, 1/dt), flush=True)
    slv.free()
