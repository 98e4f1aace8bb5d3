"""8-column strips of the CG column stages (option narrow_strips, default for <= 1024 positions at ndet 256) against the 16-column
strips: native CG loop at 512 / 1024 / 256 positions of 256^2, same box."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn, _native as nat
for (R1, R2) in ((16, 32), (32, 32), (16, 16)):
    p = syn.make_problem(R1, R2, 8, 256, 256, seed=1234)
    D = lambda x: torch.as_tensor(x, device='cuda')
    rng = np.random.default_rng(3)
    probe = (p['probe'] * np.exp(2j*np.pi*rng.random((256,256)))).astype(np.complex64)
    for rep in range(2):
        for narrow in (0, 1):
            slv = pt.CGPtychoSolver(p['nscan'], 256, 256, 1, p['nz'], p['n']); slv.verbose = False
            nat.check(nat.set_option(slv._h, b"narrow_strips", narrow))
            psi, scan, prb = D(p['psi']), D(p['scan']), D(probe)
            data = (torch.abs(slv.fwd(psi, scan, prb))**2).contiguous()
            for rec in (False, True):
                slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=100, recover_prb=rec); torch.cuda.synchronize()
                t = time.perf_counter()
                res = slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=200, recover_prb=rec); torch.cuda.synchronize()
                dt = (time.perf_counter()-t)/200
                print("%5d positions narrow=%d recover_prb=%-5s: %.3f ms/iter  (|psi| sum %.6f)" % (p['nscan'], narrow, rec, dt*1e3, float(res['psi'].abs().sum())), flush=True)
            slv.free()
