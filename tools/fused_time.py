import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn, _native as nat
D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device='cuda')
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(p["probe"])
with pt.PtychoCuFFT(4096, 256, 256, 1, 768, 768) as s:
    for mode in [int(m) for m in sys.argv[1:]] or (1, 2):
        nat.check(nat.set_option(s._h, b"fused", mode))
        g = s.fwd(psi, scan, prb); torch.cuda.synchronize()
        s.profile(True)
        for _ in range(10): g = s.fwd(psi, scan, prb)
        prof = s.profile_read(); s.profile(False)
        print("fused", mode, {k: round(v[0] / v[1], 4) for k, v in prof.items()}, flush=True)
