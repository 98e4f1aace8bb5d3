"""Where does the fused N = 256 CG loop leave the oracle's trajectory? (probe recovery on)"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'libtike-cufft_amd')
import numpy as np, torch
from oracle import cg_oracle as cg
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
nprb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = syn.make_problem(8, 8, 8, nprb, 256, seed=11)
rng = np.random.default_rng(111)
probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((nprb, nprb)))).astype(np.complex64)
ora = cg.OracleSolver(p["nscan"], nprb, 256, 1, p["nz"], p["n"])
data = (np.abs(ora.fwd(p["psi"], p["scan"], probe[:, 0])) ** 2).astype(np.float32)
start = probe.copy().swapaxes(2, 3)
for piter in (1, 2, 3):
    ora = cg.OracleSolver(p["nscan"], nprb, 256, 1, p["nz"], p["n"])
    so = p["scan"].copy(); po = start.copy()
    want = ora.run(data.copy(), np.ones_like(p["psi"]), so, po, piter=piter, recover_prb=True)
    for fused in (True, False):
        with pt.CGPtychoSolver(p["nscan"], nprb, 256, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every, slv.fused = False, 1, fused
            sg = torch.as_tensor(p["scan"].copy(), device="cuda"); pg = torch.as_tensor(start.copy(), device="cuda")
            got = slv.run(torch.as_tensor(data, device="cuda"), torch.ones(p["psi"].shape, dtype=torch.complex64, device="cuda"), sg, pg, piter=piter, recover_prb=True)
            ds = np.abs(sg.cpu().numpy() - so)
            print("piter", piter, "fused", fused, "scan diff max %.4f, positions moved differently: %d of %d" % (ds.max(), int((ds.max(-1) > 1e-4).sum()), so.shape[1]),
                  "psi %.2e" % (np.abs(got["psi"].cpu().numpy() - want["psi"]).max() / np.abs(want["psi"]).max()),
                  "prb %.2e" % (np.abs(got["probe"].cpu().numpy() - want["probe"]).max() / np.abs(want["probe"]).max()),
                  "cost", slv.history[-1][3], ora.history[-1][3])
