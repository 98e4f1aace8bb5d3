"""Figures of merit for a ptychographic reconstruction against the known truth.

A ptychographic solution is only defined up to (1) one complex factor traded between
object and probe (``psi c``, ``probe / c``), (2) a common translation of probe and object
(``probe(r - d)``, ``psi(r - d)`` give the same exit waves up to a shift, hence the same
intensities; the single-mode demo starts from the TRANSPOSED probe and settles 0.75 px
off), and (3) in the poorly lit border of the scanned area the data say little.  The
errors below are therefore taken after the best complex scale (least squares) and the
best sub-pixel translation (Fourier shift, searched within +-``max_shift`` px), inside the
well-lit region: pixels whose summed probe intensity over all scan positions is at least
``floor`` of the maximum.

Used by ``tests/test_hip_reconstruction.py`` (GPU) and ``tools/recon_calib.py`` (CPU
oracle run that produced the thresholds written in that test).
"""
import numpy as np


def illumination(scan, probe, nz, n):
    """sum over positions and modes of |probe|^2 placed at the whole-pixel position."""
    ill = np.zeros((nz, n), dtype=np.float64)
    amp2 = (np.abs(probe.reshape(-1, probe.shape[-2], probe.shape[-1])) ** 2).sum(0)
    nprb = amp2.shape[0]
    for py, px in np.asarray(scan, dtype=np.float64).reshape(-1, 2):
        if py < 0 or px < 0:
            continue
        sy, sx = int(py), int(px)
        if sy + nprb > nz or sx + nprb > n:
            continue
        ill[sy:sy + nprb, sx:sx + nprb] += amp2
    return ill


def lit_mask(scan, probe, nz, n, floor=0.1):
    ill = illumination(scan, probe, nz, n)
    return ill >= floor * ill.max()


def scaled_error(rec, true, mask=None):
    """min over complex c of ||c rec - true|| / ||true|| (on mask); returns (err, c)."""
    r = rec[mask] if mask is not None else rec.ravel()
    t = true[mask] if mask is not None else true.ravel()
    r = r.astype(np.complex128)
    t = t.astype(np.complex128)
    c = np.vdot(r, t) / np.vdot(r, r)
    return float(np.linalg.norm(c * r - t) / np.linalg.norm(t)), c


def phase_rms(rec, true, mask, c):
    """|true|^2-weighted RMS of angle(c rec conj(true)) on mask, radians (the fixture's object
    has pixels of zero amplitude, whose phase means nothing)."""
    t = true[mask]
    d = np.angle((c * rec[mask]) * np.conj(t))
    w = np.abs(t) ** 2
    return float(np.sqrt(np.sum(w * d ** 2) / np.sum(w)))


def fourier_shift(a, dy, dx):
    ky = np.fft.fftfreq(a.shape[0])[:, None]
    kx = np.fft.fftfreq(a.shape[1])[None, :]
    return np.fft.ifft2(np.fft.fft2(a) * np.exp(-2j * np.pi * (ky * dy + kx * dx)))


def aligned_error(rec, true, mask=None, max_shift=2.0):
    """min over translations d (|d_i| <= max_shift, to 1/16 px) and complex c of
    ||c shift(rec, d) - true|| / ||true||; returns (err, c, (dy, dx), shifted rec)."""
    best = (np.inf, 1.0, (0.0, 0.0))
    centre, step = (0.0, 0.0), 0.5
    span = max_shift
    while step >= 1.0 / 16:
        n = int(round(span / step))
        for iy in range(-n, n + 1):
            for ix in range(-n, n + 1):
                d = (centre[0] + iy * step, centre[1] + ix * step)
                e, c = scaled_error(fourier_shift(rec, *d), true, mask)
                if e < best[0]:
                    best = (e, c, d)
        centre, span, step = best[2], step, step / 4
    return best[0], best[1], best[2], fourier_shift(rec, *best[2])


def mode_errors(rec_modes, true_modes, shift):
    """per-mode scaled error at the common translation, each mode with its own complex factor"""
    return [scaled_error(fourier_shift(rec_modes[k], *shift), true_modes[k])[0] for k in range(true_modes.shape[0])]


def report(psi, probe, psi_true, probe_true, scan, floor=0.1):
    """dict of the figures asserted by the reconstruction tests (angle 0)."""
    nz, n = psi_true.shape[-2:]
    mask = lit_mask(scan[0], probe_true[0], nz, n, floor)
    # the translation is read off the strongest probe mode (128^2: cheap to search) and shared by all modes and the object
    _, _, shift, _ = aligned_error(probe[0, 0], probe_true[0, 0])
    psi_s = fourier_shift(psi[0], *shift)
    err, c = scaled_error(psi_s, psi_true[0], mask)
    out = {
        "obj_err": err,
        "obj_phase_rms": phase_rms(psi_s, psi_true[0], mask, c),
        "obj_scale": abs(c),
        "shift": [float(shift[0]), float(shift[1])],
        "lit_fraction": float(mask.mean()),
        "prb_err": mode_errors(probe[0], probe_true[0], shift),
    }
    return out
