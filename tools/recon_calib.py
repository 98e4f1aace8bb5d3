"""The reference's two demo reconstructions (tests/test.py:18-64, tests/test_modes.py:18-60) on its own fixtures, at
full length, with the figures of merit of tests/recon_metrics.py.  --backend oracle runs the NumPy restatement on the
CPU (minutes to hours: the thresholds of tests/test_hip_reconstruction.py were set from such a run), --backend gpu
the HIP path.

    python tools/recon_calib.py --backend gpu --scenario single --nscan 1000 --piter 128
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd"), os.path.join(ROOT, "tests")]
import numpy as np

import recon_metrics as rm

N, NZ, NPRB, NDET = 600, 276, 128, 128


def scenario(model, which, nscan):
    psi0 = np.ones([1, NZ, N], dtype="complex64")
    psi0[0] = model["initpsiamp"] * np.exp(1j * model["initpsiang"])
    if which == "single":
        prb_true = np.zeros([1, 1, NPRB, NPRB], dtype="complex64")
        prb_true[0, 0] = model["prbamp"] * np.exp(1j * model["prbang"])
        prb_init = prb_true.copy().swapaxes(2, 3)                       # tests/test.py:58
        temp = np.moveaxis(model["coords"], 0, 1)[:nscan]               # tests/test.py:38
    else:
        nmodes = 3
        prb_true = np.zeros([1, nmodes, NPRB, NPRB], dtype="complex64")
        prb_true[0] = (model["probes_amp"] * np.exp(1j * model["probes_ang"]))[:nmodes]
        prb_init = prb_true.copy()
        for k in range(nmodes):                                          # tests/test_modes.py:42-43
            prb_init[:, k] /= np.max(np.abs(prb_init[:, k]))
        temp = np.moveaxis(model["coords"], 0, 1)[:nscan * 5:5]         # tests/test_modes.py:46
    scan = np.ones([1, temp.shape[0], 2], dtype="float32")
    scan[0, :, 0] = temp[:, 1]
    scan[0, :, 1] = temp[:, 0]
    return psi0, prb_true, prb_init, scan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gpu")
    ap.add_argument("--scenario", default="single")
    ap.add_argument("--nscan", type=int, default=1000)
    ap.add_argument("--piter", type=int, default=128)
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    model = dict(np.load(os.path.join(ROOT, "tests", "golden", "model_fixtures.npz")))
    psi0, prb_true, prb_init, scan = scenario(model, a.scenario, a.nscan)
    nscan = scan.shape[1]
    if a.backend == "gpu":
        import libtike.hipfft as pt
        slv = pt.CGPtychoSolver(nscan, NPRB, NDET, 1, NZ, N)
        slv.verbose, slv.log_every = False, 1
    else:
        from oracle import cg_oracle as cg
        slv = cg.OracleSolver(nscan, NPRB, NDET, 1, NZ, N)
    data = np.zeros([1, nscan, NDET, NDET], dtype="float32")
    for k in range(prb_true.shape[1]):
        data += np.abs(slv.fwd_ptycho_batch(psi0, scan, prb_true[:, k:k + 1])) ** 2
    psi = np.ones([1, NZ, N], dtype="complex64")
    t0 = time.perf_counter()
    res = slv.run_batch(data, psi, scan, prb_init.copy(), piter=a.piter, model="gaussian", recover_prb=True)
    dt = time.perf_counter() - t0
    hist = list(slv.history)
    out = rm.report(res["psi"], res["probe"], psi0, prb_true, scan)
    out.update(backend=a.backend, scenario=a.scenario, nscan=nscan, piter=a.piter, seconds=dt,
               cost_first=hist[0][3], cost_last=hist[-1][3], cost_ratio=hist[-1][3] / hist[0][3],
               start=rm.report(psi, prb_init, psi0, prb_true, scan))
    print(json.dumps(out))
    if a.save:
        np.savez_compressed(a.save, psi=res["psi"], probe=res["probe"])
    print("costs:", " ".join("%.4g" % h[3] for h in hist[::8]))


if __name__ == "__main__":
    main()
