"""Check the XCD-team forward kernel against the two-pass path and time both."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
p["scan"][0,5] = [-3.0, 2.0]
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.PtychoCuFFT(4096,256,256,1,768,768)
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
ref = slv.fwd(psi,scan,prb)
slv.set_team(True)
got = slv.fwd(psi,scan,prb)
print("aborted:", slv.team_aborted())
d = (got-ref).abs().max().item(); print("max abs diff", d, "ref max", ref.abs().max().item(), "zeros ok", bool((got[0,5]==0).all()))
def T(n=10):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): slv.fwd(psi,scan,prb)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("team fwd ms", T()); print("aborted:", slv.team_aborted())
slv.set_team(False); print("two-pass fwd ms", T())
