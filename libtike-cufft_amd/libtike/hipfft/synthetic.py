"""Synthetic problem generators (NumPy, host side) for tests and ``bench.py``.

Geometry follows SURVEY.md section 8(d): raster scan with sub-pixel jitter,
``scan[..., 0]`` = row (y) and ``scan[..., 1]`` = column (x) offset of the
patch's top-left corner (``/root/reference/tests/test_adjoint.py:30-33``).
"""
import numpy as np

__all__ = ["raster_scan", "gaussian_probe", "hermite_modes", "random_object",
           "make_problem", "object_size_for"]


def raster_scan(ny, nx, step, rng, ntheta=1, y0=0.0, x0=0.0):
    """``ny x nx`` raster, ``step`` px apart, plus U[0,1) jitter per axis."""
    yy, xx = np.meshgrid(np.arange(ny) * step, np.arange(nx) * step,
                         indexing="ij")
    scan = np.empty((ntheta, ny * nx, 2), dtype=np.float32)
    for t in range(ntheta):
        scan[t, :, 0] = (yy.ravel() + y0 + rng.random(ny * nx)).astype(np.float32)
        scan[t, :, 1] = (xx.ravel() + x0 + rng.random(ny * nx)).astype(np.float32)
    return scan


def object_size_for(ny, nx, step, nprb, align=64):
    """Smallest ``(nz, n)`` (multiples of ``align``) that keeps
    ``trunc(pos) + nprb + 1 <= size`` for every raster position."""
    def up(v):
        return int(-(-v // align) * align)
    return up((ny - 1) * step + 1 + nprb + 1), up((nx - 1) * step + 1 + nprb + 1)


def gaussian_probe(nprb, ntheta=1):
    """2-D Gaussian (sigma = nprb/6) times a quadratic phase, max |.| = 1."""
    r = np.arange(nprb) - (nprb - 1) / 2.0
    y, x = np.meshgrid(r, r, indexing="ij")
    r2 = x * x + y * y
    p = np.exp(-r2 / (2 * (nprb / 6.0) ** 2)) * np.exp(1j * 0.1 * r2 / nprb)
    p = (p / np.abs(p).max()).astype(np.complex64)
    return np.repeat(p[None], ntheta, 0)


def hermite_modes(nprb, nmodes, ntheta=1):
    """Gaussian x Hermite orders (0,0),(1,0),(0,1),(1,1) with amplitudes
    1, .5, .5, .25 -- the multi-mode probe of SURVEY.md cfg 3."""
    orders = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2)][:nmodes]
    amps = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25][:nmodes]
    base = gaussian_probe(nprb)[0]
    r = (np.arange(nprb) - (nprb - 1) / 2.0) / (nprb / 6.0)
    y, x = np.meshgrid(r, r, indexing="ij")
    herm = [np.ones_like(x), 2 * x, 4 * x * x - 2]
    hermy = [np.ones_like(y), 2 * y, 4 * y * y - 2]
    out = np.empty((ntheta, nmodes, nprb, nprb), dtype=np.complex64)
    for k, ((ox, oy), a) in enumerate(zip(orders, amps)):
        m = base * herm[ox] * hermy[oy]
        out[:, k] = (a * m / np.abs(m).max()).astype(np.complex64)
    return out


def random_object(nz, n, rng, ntheta=1):
    """``amp ~ U[0.8,1]``, ``phase ~ U[-0.5,0.5]``."""
    amp = 0.8 + 0.2 * rng.random((ntheta, nz, n))
    ph = rng.random((ntheta, nz, n)) - 0.5
    return (amp * np.exp(1j * ph)).astype(np.complex64)


def make_problem(ny, nx, step, nprb, ndet=None, ntheta=1, seed=1234,
                 nz=None, n=None):
    """Return dict(psi, scan, probe[ntheta,nprb,nprb], sizes...)."""
    rng = np.random.default_rng(seed)
    ndet = ndet or nprb
    anz, an = object_size_for(ny, nx, step, nprb)
    nz, n = nz or anz, n or an
    return {
        "psi": random_object(nz, n, rng, ntheta),
        "scan": raster_scan(ny, nx, step, rng, ntheta),
        "probe": gaussian_probe(nprb, ntheta),
        "nscan": ny * nx, "nprb": nprb, "ndet": ndet, "ntheta": ntheta,
        "nz": nz, "n": n, "rng": rng,
    }
