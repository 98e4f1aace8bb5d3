import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_hip_cg import setup
import libtike.hipfft as pt
for nm, rec in ((1,False),(1,True),(2,True)):
    p, probe, ora, data = setup(nm)
    start = probe.copy().swapaxes(2, 3) if rec else probe.copy()
    ora.history.clear()
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(), piter=6, recover_prb=rec)
    slv = pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]); slv.verbose=False; slv.log_every=1
    got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(), piter=6, recover_prb=rec)
    for a,b in zip(slv.history, ora.history): print(a,b)
    print(np.abs(got["psi"]-want["psi"]).max(), np.abs(got["probe"]-want["probe"]).max()/np.abs(want["probe"]).max())
