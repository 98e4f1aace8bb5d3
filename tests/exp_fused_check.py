"""Run by tests/test_hip_experiments.py in a process of its own with PTYCHO_HIP_LIB = the experiments build
(tools/build/libptychohip_exp.so): the single-launch forward k_fwd_fused256 -- a measured-slower experiment that the
shipped library does not contain -- against the oracle and against the two-pass paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np
import torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
from oracle import ptycho_oracle as op

REL_MAX = 2e-5
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
host = lambda t: t.cpu().numpy()

# (1) against the oracle: padded probes, two angles, a skipped position, positions that overhang the object edges
for nprb, ntheta in ((256, 1), (128, 2), (200, 1)):
    p = syn.make_problem(3, 4, 13, nprb, 256, ntheta=ntheta, seed=3)
    scan = p["scan"].copy()
    scan[0, 0] = (-1.5, 3.0)                                # skipped (kernels.cu:39)
    scan[0, 1] = (p["nz"] - nprb + 0.5, 2.25)               # overhangs the bottom edge
    scan[0, 2] = (0.75, p["n"] - nprb - 0.5)                # touches the right edge
    rng = np.random.default_rng(5)
    prb = (p["probe"] * np.exp(2j * np.pi * rng.random((nprb, nprb)))).astype(np.complex64)
    want = op.fwd(p["psi"], scan, prb, 256, "double")
    with pt.PtychoCuFFT(p["nscan"], nprb, 256, ntheta, p["nz"], p["n"]) as slv:
        for tiles in (1, 2):
            slv.set_fused(tiles)
            g = host(slv.fwd(dev(p["psi"]), dev(scan), dev(prb)))
            assert np.abs(g - want).max() <= REL_MAX * np.abs(want).max(), (nprb, ntheta, tiles)
            assert not g[0, 0].any()

# (2) against the shipped two-pass path at 2304 positions
p = syn.make_problem(48, 48, 4, 256, 256, seed=2, nz=512, n=512)
with pt.PtychoCuFFT(2304, 256, 256, 1, 512, 512) as slv:
    psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
    slv.set_fused(0)
    ref = host(slv.fwd(psi, scan, prb))
    for tiles in (1, 2):
        slv.set_fused(tiles)
        one = host(slv.fwd(psi, scan, prb))
        assert np.abs(one - ref).max() <= 1e-5 * np.abs(ref).max(), tiles
print("experiments build: fused forward ok")
