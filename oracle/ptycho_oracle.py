"""CPU oracle for the ptychography forward / adjoint operators.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The shipped path (``libtike.hipfft``) never imports this module and raises if
its HIP library is missing.

What it restates (reference = nikitinvv/libtike-cufft v0.4.0, paths relative to
``/root/reference``):

* ``src/cuda/kernels.cu:8-108``   -- ``muloperator``: index math, bilinear
  weights and their multiplication order, conj placement, ``c = 1/ndet``,
  negative-position skip rule;
* ``src/cuda/ptychofft.cu:60-88`` -- operator sequencing: zeroed scratch,
  centred zero-pad when ``nprb < ndet``, unnormalised forward / inverse DFT,
  accumulate-into-output semantics of ``adj``;
* ``src/libtike/cufft/ptycho.py:80-123`` -- shapes / dtypes of
  ``fwd / adj / adj_probe``.

Third-party arithmetic absent from the reference tree: cuFFT
(``cufftExecC2C``, CUDA toolkit, version unpinned in
``src/cuda/CMakeLists.txt:10``).  Its published definition -- the
unnormalised DFT ``X[k] = sum_n x[n] exp(-/+ 2 pi i k n / N)`` -- is restated
here with ``scipy.fft`` (pocketfft).

Pinning status.  The reference cannot be built or imported in the build
container (CUDA + CuPy + cuFFT only, no ``nvcc``, no GPU driver for it), so no
output of the reference binary exists to compare against.  The oracle is pinned
by (i) the reference's only numerical check, the adjoint identities of
``tests/test_adjoint.py:42-56`` evaluated on the reference's own fixtures
(``tests/model/*``; committed as ``tests/golden/model_fixtures.npz``), and
(ii) analytic known-answer tests that do not depend on this restatement
(``tests/test_oracle.py``).  Absolute values beyond those checks are
**parity unpinned** against the reference binary.

Behaviour outside the reference's defined domain: the reference performs no
upper bounds check (``kernels.cu:39`` rejects negative positions only) and
reads/writes out of bounds when ``trunc(pos) + nprb + 1`` exceeds the object.
Here -- and in the HIP path -- taps that fall outside the object read as zero
(``fwd``, ``adj_probe``) or are dropped (``adj``).
"""

import numpy as np
import scipy.fft as _fft

__all__ = [
    "split_positions", "patches", "fwd", "adj", "adj_probe", "nearplane",
    "fft2_unnorm", "ifft2_unnorm",
]


def _ctype(precision):
    return (np.complex64, np.float32) if precision == "single" else (
        np.complex128, np.float64)


def split_positions(scan, precision="single"):
    """``modff`` split of the scan positions -- ``kernels.cu:27-28,39``.

    ``scan[..., 0]`` is the row (y) offset and ``scan[..., 1]`` the column (x)
    offset.  Returns integer parts (int64), fractional parts and the ``valid``
    mask (``False`` where the reference returns early: integer part < 0; note
    ``trunc(-0.3) = -0.0`` is *not* < 0, so such positions stay valid with a
    negative fraction, as in the reference).
    """
    _, ft = _ctype(precision)
    # the reference splits the float32 value; keep that even in double mode so
    # both modes interpolate at the same points.
    s32 = np.asarray(scan, dtype=np.float32)
    ipart = np.trunc(s32)
    frac = (s32 - ipart).astype(ft)
    valid = ~((ipart[..., 0] < 0) | (ipart[..., 1] < 0))
    return ipart.astype(np.int64), frac, valid


def patches(psi, scan, nprb, precision="single"):
    """Bilinearly interpolated object patches -- ``kernels.cu:97-104`` (and
    ``:84-91``).  Returns ``[ptheta, nscan, nprb, nprb]``.

    ``tmp = f00*(1-fx)*(1-fy) + f01*fx*(1-fy) + f10*(1-fx)*fy + f11*fx*fy``
    evaluated left to right in the working precision, taps
    ``f00=psi[y,x], f01=psi[y,x+1], f10=psi[y+1,x], f11=psi[y+1,x+1]``.
    """
    ct, ft = _ctype(precision)
    psi = np.asarray(psi)
    ptheta, nz, n = psi.shape
    ip, fr, valid = split_positions(scan, precision)
    nscan = ip.shape[1]
    out = np.zeros((ptheta, nscan, nprb, nprb), dtype=ct)
    ar = np.arange(nprb + 1)
    for t in range(ptheta):
        f = psi[t].astype(ct)
        sy = ip[t, :, 0]
        sx = ip[t, :, 1]
        rows = sy[:, None] + ar[None, :]          # [nscan, nprb+1]
        cols = sx[:, None] + ar[None, :]
        rok = (rows >= 0) & (rows < nz)
        cok = (cols >= 0) & (cols < n)
        rc = np.clip(rows, 0, nz - 1)
        cc = np.clip(cols, 0, n - 1)
        big = f[rc[:, :, None], cc[:, None, :]]   # [nscan, nprb+1, nprb+1]
        big = np.where(rok[:, :, None] & cok[:, None, :], big, ct(0))
        fy = fr[t, :, 0][:, None, None]
        fx = fr[t, :, 1][:, None, None]
        one = ft(1)
        p = big[:, :-1, :-1] * (one - fx) * (one - fy)
        p = p + big[:, :-1, 1:] * fx * (one - fy)
        p = p + big[:, 1:, :-1] * (one - fx) * fy
        p = p + big[:, 1:, 1:] * fx * fy
        p[~valid[t]] = 0
        out[t] = p
    return out


def fft2_unnorm(x):
    """cuFFT ``CUFFT_FORWARD`` on the last two axes (``ptychofft.cu:72``)."""
    return _fft.fft2(x, axes=(-2, -1), norm="backward")


def ifft2_unnorm(x):
    """cuFFT ``CUFFT_INVERSE``: sign +, **no** 1/N^2 (``ptychofft.cu:85``)."""
    return _fft.ifft2(x, axes=(-2, -1), norm="forward")


def nearplane(psi, scan, probe, ndet, precision="single"):
    """Scratch buffer after ``muloperator(flg=2)`` -- ``kernels.cu:95-107``
    with the centred zero pad of ``kernels.cu:48-57``; ``c = 1/ndet``
    (``kernels.cu:65``)."""
    ct, ft = _ctype(precision)
    probe = np.asarray(probe)
    ptheta, nprb = probe.shape[0], probe.shape[-1]
    assert probe.shape == (ptheta, nprb, nprb)
    p = patches(psi, scan, nprb, precision)
    c = ft(1.0) / ft(ndet)
    near = np.zeros((ptheta, p.shape[1], ndet, ndet), dtype=ct)
    pad = (ndet - nprb) // 2
    q = probe.astype(ct)[:, None] * p
    near[:, :, pad:pad + nprb, pad:pad + nprb] = c * q
    # skipped positions keep the memset zeros (ptychofft.cu:69)
    return near


def fwd(psi, scan, probe, ndet, precision="single"):
    """``ptychofft::fwd`` (``ptychofft.cu:60-73``) / ``PtychoCuFFT.fwd``
    (``ptycho.py:80-89``).  psi ``[ptheta,nz,n]``, scan ``[ptheta,nscan,2]``,
    probe ``[ptheta,nprb,nprb]`` -> farplane ``[ptheta,nscan,ndet,ndet]``."""
    ct, _ = _ctype(precision)
    return fft2_unnorm(nearplane(psi, scan, probe, ndet, precision)).astype(ct)


def _adj_nearplane(farplane, nprb, precision):
    ct, ft = _ctype(precision)
    g = np.asarray(farplane).astype(ct)
    ndet = g.shape[-1]
    pad = (ndet - nprb) // 2
    near = ifft2_unnorm(g).astype(ct)
    return near[:, :, pad:pad + nprb, pad:pad + nprb], ft(1.0) / ft(ndet)


def adj(farplane, scan, probe, nz, n, precision="single"):
    """``ptychofft::adj`` with ``flg=0`` (``ptychofft.cu:76-88``,
    ``kernels.cu:69-81``) / ``PtychoCuFFT.adj`` (``ptycho.py:97-106``):
    ``psi += w * c * conj(prb) * IDFT2(g)`` scattered with the four bilinear
    weights.  Output starts from zeros (``ptycho.py:102``)."""
    ct, ft = _ctype(precision)
    probe = np.asarray(probe)
    ptheta, nprb = probe.shape[0], probe.shape[-1]
    near, c = _adj_nearplane(farplane, nprb, precision)
    ip, fr, valid = split_positions(scan, precision)
    nscan = ip.shape[1]
    # accumulate on a canvas padded so that out-of-object taps are dropped
    out = np.zeros((ptheta, nz, n), dtype=ct)
    one = ft(1)
    for t in range(ptheta):
        tmp = c * (np.conj(probe[t].astype(ct))[None] * near[t])  # [nscan,nprb,nprb]
        big = np.zeros((nz + nprb + 2, n + nprb + 2), dtype=ct)
        for s in range(nscan):
            if not valid[t, s]:
                continue
            sy, sx = int(ip[t, s, 0]), int(ip[t, s, 1])
            if sy >= nz or sx >= n:
                continue
            fy, fx = fr[t, s, 0], fr[t, s, 1]
            v = tmp[s]
            big[sy:sy + nprb, sx:sx + nprb] += v * (one - fx) * (one - fy)
            big[sy:sy + nprb, sx + 1:sx + nprb + 1] += v * fx * (one - fy)
            big[sy + 1:sy + nprb + 1, sx:sx + nprb] += v * (one - fx) * fy
            big[sy + 1:sy + nprb + 1, sx + 1:sx + nprb + 1] += v * fx * fy
        out[t] = big[:nz, :n]
    return out


def adj_probe(farplane, scan, psi, nprb, precision="single"):
    """``ptychofft::adj`` with ``flg=1`` (``kernels.cu:82-94``) /
    ``PtychoCuFFT.adj_probe`` (``ptycho.py:113-123``):
    ``prb += c * IDFT2(g) * conj(bilerp(psi))`` summed over scan positions."""
    ct, _ = _ctype(precision)
    near, c = _adj_nearplane(farplane, nprb, precision)
    p = patches(psi, scan, nprb, precision)     # zero where skipped
    return (c * (near * np.conj(p))).sum(axis=1).astype(ct)
