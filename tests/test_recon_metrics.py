"""The figures of merit of tests/test_hip_reconstruction.py checked on constructed cases (CPU): a reconstruction that differs
from the truth only by what ptychography leaves open -- one complex factor traded between object and probe, one common
sub-pixel translation -- must score ~0, and a real difference must show."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import recon_metrics as rm  # noqa: E402


def smooth_field(shape, seed):
    rng = np.random.default_rng(seed)
    spec = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
    ky = np.fft.fftfreq(shape[0])[:, None]
    kx = np.fft.fftfreq(shape[1])[None, :]
    spec *= np.exp(-(ky ** 2 + kx ** 2) / (2 * 0.04 ** 2))          # band limited: a Fourier shift is exact
    f = np.fft.ifft2(spec)
    return (f / np.abs(f).max()).astype(np.complex64)


def problem():
    nz, n, nprb = 96, 128, 32
    psi = (0.3 + smooth_field((nz, n), 1))[None]
    yy, xx = np.mgrid[:nprb, :nprb] - nprb / 2
    prb = (np.exp(-(yy ** 2 + xx ** 2) / (2 * 5.0 ** 2)) * np.exp(0.05j * (yy ** 2 + xx ** 2))).astype(np.complex64)[None, None]
    gy, gx = np.meshgrid(np.arange(4, 60, 6), np.arange(4, 92, 6), indexing="ij")
    scan = np.stack([gy.ravel(), gx.ravel()], -1).astype(np.float32)[None]
    return psi, prb, scan


def test_factor_and_translation_are_not_errors():
    psi, prb, scan = problem()
    c = 0.7 * np.exp(0.9j)
    d = (0.625, -0.75)
    psi_rec = (rm.fourier_shift(psi[0], -d[0], -d[1]) * c)[None]
    prb_rec = (rm.fourier_shift(prb[0, 0], -d[0], -d[1]) / c)[None, None]
    rep = rm.report(psi_rec, prb_rec, psi, prb, scan)
    assert abs(rep["shift"][0] - d[0]) <= 1 / 16 + 1e-9 and abs(rep["shift"][1] - d[1]) <= 1 / 16 + 1e-9
    assert rep["obj_err"] < 2e-3 and rep["prb_err"][0] < 2e-3 and rep["obj_phase_rms"] < 2e-3
    assert abs(rep["obj_scale"] - 1 / 0.7) < 1e-2
    # without the alignment the same pair would look badly wrong
    assert rm.scaled_error(prb_rec[0, 0], prb[0, 0])[0] > 0.05


def test_a_real_difference_shows():
    psi, prb, scan = problem()
    wrong = psi * np.exp(0.3j * np.abs(smooth_field(psi.shape[1:], 9)))[None]
    rep = rm.report(wrong, prb, psi, prb, scan)
    assert rep["shift"] == [0.0, 0.0]
    assert 0.02 < rep["obj_err"] < 0.3 and rep["prb_err"][0] < 1e-6
    flat = rm.report(np.ones_like(psi), prb, psi, prb, scan)
    assert flat["obj_err"] > rep["obj_err"]


def test_lit_mask_is_the_scanned_region():
    psi, prb, scan = problem()
    mask = rm.lit_mask(scan[0], prb[0], psi.shape[1], psi.shape[2], 0.1)
    assert mask[40, 50] and not mask[90, 120] and not mask[0, 0]
    assert 0.2 < mask.mean() < 0.8
