"""The reference's demo scenarios on the reference's own fixtures, shortened so the NumPy
oracle finishes in seconds, GPU loop vs oracle loop:

* ``/root/reference/tests/test.py:18-64``       -- single mode, probe recovered from the
  transposed true probe, measured scan coordinates, 600 x 276 object;
* ``/root/reference/tests/test_modes.py:18-60`` -- three incoherent modes from
  ``probes_*.tiff`` normalised to max 1, every 5th coordinate.

The reference stores no golden output for them (it writes TIFFs for a human), so the
checker is the oracle loop on the same inputs.
"""
import numpy as np
import pytest

from oracle import cg_oracle as cg

pytestmark = pytest.mark.gpu

N, NZ, NPRB, NDET = 600, 276, 128, 128


@pytest.fixture(scope="module")
def pt():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    return pt


def scan_from(model, sel):
    temp = np.moveaxis(model["coords"], 0, 1)[sel]
    scan = np.ones([1, temp.shape[0], 2], dtype="float32")
    scan[0, :, 0] = temp[:, 1]
    scan[0, :, 1] = temp[:, 0]
    return scan


def true_object(model):
    psi0 = np.ones([1, NZ, N], dtype="complex64")
    psi0[0] = model["initpsiamp"] * np.exp(1j * model["initpsiang"])
    return psi0


def compare(pt, data, psi, scan, prb, piter, nmodes):
    ora = cg.OracleSolver(scan.shape[1], NPRB, NDET, 1, NZ, N)
    want = ora.run_batch(data, psi, scan, prb, piter=piter, model="gaussian", recover_prb=True)
    with pt.CGPtychoSolver(scan.shape[1], NPRB, NDET, 1, NZ, N) as slv:
        slv.verbose, slv.log_every = False, 1
        got = slv.run_batch(data, psi, scan, prb, piter=piter, model="gaussian", recover_prb=True)
        hist = list(slv.history)
    for (i, gp, gq, c), (io, gpo, gqo, co) in zip(hist, ora.history):
        assert (i, gp, gq) == (io, gpo, gqo), (hist, ora.history)
        assert abs(c - co) <= 2e-4 * abs(co)
    assert hist[-1][3] < hist[0][3]
    assert np.abs(got["psi"] - want["psi"]).max() < 5e-4 * np.abs(want["psi"]).max()
    assert np.abs(got["probe"] - want["probe"]).max() < 5e-4 * np.abs(want["probe"]).max()


def test_single_mode_demo_scenario(pt, model):
    """tests/test.py: data = |fwd(true object, true probe)|^2, start psi = 1, probe = transposed."""
    nscan, piter = 160, 6
    prb0 = np.zeros([1, 1, NPRB, NPRB], dtype="complex64")
    prb0[0, 0] = model["prbamp"] * np.exp(1j * model["prbang"])
    scan = scan_from(model, slice(0, nscan))
    psi0 = true_object(model)
    ora = cg.OracleSolver(nscan, NPRB, NDET, 1, NZ, N)
    data = (np.abs(ora.fwd_ptycho_batch(psi0, scan, prb0)) ** 2).astype(np.float32)
    with pt.PtychoCuFFT(nscan, NPRB, NDET, 1, NZ, N) as slv:
        data_gpu = np.abs(slv.fwd_ptycho_batch(psi0, scan, prb0)) ** 2
    assert np.abs(data_gpu - data).max() < 1e-4 * data.max()
    psi = np.ones([1, NZ, N], dtype="complex64")
    prb = prb0.copy().swapaxes(2, 3)
    compare(pt, data, psi, scan, prb, piter, 1)


def test_three_mode_demo_scenario(pt, model):
    """tests/test_modes.py: 3 modes normalised to max 1, coords[:5500:5] (here every 25th)."""
    nmodes, piter = 3, 4
    prb0 = np.zeros([1, nmodes, NPRB, NPRB], dtype="complex64")
    prb0[0] = (model["probes_amp"] * np.exp(1j * model["probes_ang"]))[:nmodes]
    prb0 /= np.abs(prb0).max()                      # tests/test_modes.py:38
    scan = scan_from(model, slice(0, 5500, 25))
    nscan = scan.shape[1]
    psi0 = true_object(model)
    ora = cg.OracleSolver(nscan, NPRB, NDET, 1, NZ, N)
    data = np.zeros([1, nscan, NDET, NDET], dtype="float32")
    for k in range(nmodes):                         # tests/test_modes.py:51-53
        data += np.abs(ora.fwd_ptycho_batch(psi0, scan, prb0[:, k:k + 1])) ** 2
    psi = np.ones([1, NZ, N], dtype="complex64")
    prb = prb0.copy().swapaxes(2, 3)
    compare(pt, data, psi, scan, prb, piter, nmodes)
