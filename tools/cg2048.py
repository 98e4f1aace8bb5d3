"""ndet = 2048: fused CG loop against the statement-by-statement loop (4 positions, 3 iterations)."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(2, 2, 23, 2048, 2048, seed=5)
rng = np.random.default_rng(6)
probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((2048, 2048)))).astype(np.complex64)
out = []
for fused in (True, False):
    with pt.CGPtychoSolver(p["nscan"], 2048, 2048, 1, p["nz"], p["n"]) as slv:
        slv.verbose, slv.log_every, slv.fused = False, 1, fused
        data = np.abs(slv.fwd_ptycho_batch(p["psi"], p["scan"], probe[:, 0])) ** 2
        res = slv.run_batch(data.astype(np.float32), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy().swapaxes(2, 3), piter=3, recover_prb=True)
        out.append((res, list(slv.history)))
(rf, hf), (ru, hu) = out
print(hf); print(hu)
print("psi diff", np.abs(rf["psi"] - ru["psi"]).max(), "probe diff", np.abs(rf["probe"] - ru["probe"]).max() / np.abs(ru["probe"]).max())
