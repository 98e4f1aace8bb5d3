"""CPU oracle for the conjugate-gradient reconstruction loop and the batched
sub-pixel registration used by its position correction.

TEST INFRASTRUCTURE ONLY (same rules as ``ptycho_oracle.py``: imported by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg,
never by the product).

Restates, bug-compatibly, ``src/libtike/cufft/ptycho.py`` of the reference:

* ``:163-188``  ``_upsampled_dft_batch``
* ``:190-248``  ``register_translation_batch``
* ``:253-281``  ``CGPtychoSolver.line_search_sqr``
* ``:283-488``  ``CGPtychoSolver.run``
* ``:135-162``  ``PtychoCuFFT.run_batch``

Reference quirks that are reproduced on purpose:
the ``(b/a)`` un-scaling of ``fpsi`` while the intensity keeps the rescale
(``:351``); a complex Dai-Yuan beta (``:369-372``); ``probe`` and ``scan`` are
mutated in place (``:344,403,465``); only angle 0 gets position updates
(``:403``); shifts of a one-pattern batch are zeroed (``:243-245``); the
``model='poisson'`` branch reads ``fpsi`` before assignment at ``i == 0``
(``:358-363``) -- here it raises ``UnboundLocalError`` exactly like the
reference.  CuPy (unpinned) supplies the elementwise / reduction semantics in the
reference; NumPy's are identical up to summation order.

Pinning: the reference stores no golden reconstruction (its demos write TIFFs for
a human), so CG absolute values are **parity unpinned** against the reference
binary; the loop is pinned structurally (monotone cost, zero gradient at the true
object, recovery of known shifts by the registration).
"""

import warnings

import numpy as np
import scipy.fft as _fft

from . import ptycho_oracle as op

__all__ = ["upsampled_dft_batch", "register_translation_batch",
           "line_search_sqr", "OracleSolver"]


def upsampled_dft_batch(data, ups, upsample_factor, axis_offsets):
    """``ptycho.py:163-188``: two matrix-multiply DFTs on an ``ups x ups``
    window; first contraction is over the column axis with the *column* offset
    (``axis_offsets[:, 1]``), second over the row axis with the row offset."""
    ups = int(ups)
    nb, _, ncol = data.shape
    freq = np.fft.fftfreq(ncol, upsample_factor)
    grid = np.arange(ups)[None, :]

    def dft_matrix(off):
        ph = (grid - off[:, None])[:, :, None] * freq
        return np.exp(-2j * np.pi * ph)

    # tmp[i, j, p] = sum_k K1[i, j, k] * data[i, p, k]
    tmp = np.einsum("ijk,ipk->ijp", dft_matrix(axis_offsets[:, 1]), data)
    # rec[i, j, p] = sum_k K2[i, j, k] * tmp[i, p, k]
    return np.einsum("ijk,ipk->ijp", dft_matrix(axis_offsets[:, 0]), tmp)


def _argmax2d(a):
    flat = a.reshape(a.shape[0], -1).argmax(1)
    return np.column_stack(np.unravel_index(flat, a.shape[1:]))


def register_translation_batch(src, target, upsample_factor=1, space="real"):
    """``ptycho.py:190-248``: phase cross-correlation, batched over axis 0."""
    if space.lower() == "fourier":
        src_freq, target_freq = src, target
    elif space.lower() == "real":
        src_freq = _fft.fft2(src)
        target_freq = _fft.fft2(target)
    shape = src_freq.shape
    image_product = src_freq * target_freq.conj()
    cross = _fft.ifft2(image_product)
    maxima = _argmax2d(np.abs(cross))
    mid = np.array([np.fix(s / 2) for s in shape[1:]])
    shifts = np.array(maxima, dtype=np.float64)
    shifts[shifts[:, 0] > mid[0], 0] -= shape[1]
    shifts[shifts[:, 1] > mid[1], 1] -= shape[2]
    if upsample_factor > 1:
        shifts = np.round(shifts * upsample_factor) / upsample_factor
        region = np.ceil(upsample_factor * 1.5)
        dftshift = np.fix(region / 2.0)
        normalization = src_freq[0].size * upsample_factor ** 2
        offset = dftshift - shifts * upsample_factor
        cross = upsampled_dft_batch(image_product.conj(), region,
                                    upsample_factor, offset).conj()
        cross /= normalization
        maxima = np.array(_argmax2d(np.abs(cross)), dtype=np.float64) - dftshift
        shifts = shifts + maxima / upsample_factor
    for dim in range(src_freq.ndim):          # reference quirk, :243-245
        if shape[dim] == 1:
            shifts[dim] = 0
    return shifts


def line_search_sqr(f, p1, p2, p3, step_length=1, step_shrink=0.5):
    """``ptycho.py:253-281``: backtracking on the closed-form quadratic."""
    assert 0 < step_shrink < 1
    m = 0
    fp1 = f(p1)
    while f(p1 + step_length ** 2 * p2 + step_length * p3) > fp1 + step_shrink * m:
        if step_length < 1e-32:
            warnings.warn("Line search failed for conjugate gradient.")
            return 0
        step_length *= step_shrink
    return step_length


class OracleSolver:
    """NumPy mirror of ``PtychoCuFFT`` + ``CGPtychoSolver`` (``ptycho.py:34-162,
    250-488``), same constructor argument order (``ptycho.py:58-60``)."""

    def __init__(self, nscan, probe_shape, detector_shape, ntheta, nz, n,
                 precision="single"):
        self.nscan, self.nprb, self.ndet = nscan, probe_shape, detector_shape
        self.ptheta, self.nz, self.n = ntheta, nz, n
        self.precision = precision
        self.history = []   # (i, gammapsi, gammaprb, cost) every iteration

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # -- operators (ptycho.py:80-123) -----------------------------------------
    def fwd(self, psi, scan, probe):
        return op.fwd(psi, scan, probe, self.ndet, self.precision)

    def adj(self, farplane, scan, probe):
        return op.adj(farplane, scan, probe, self.nz, self.n, self.precision)

    def adj_probe(self, farplane, scan, psi):
        return op.adj_probe(farplane, scan, psi, self.nprb, self.precision)

    # -- host batching (ptycho.py:70-78, 91-95, 108-111, 125-129) ------------
    def _batch(self, function, output, *inputs):
        for ids in range(inputs[0].shape[0]):
            output[ids] = function(*[x[ids:ids + 1] for x in inputs])[0]
        return output

    def fwd_ptycho_batch(self, psi, scan, probe):
        out = np.zeros([scan.shape[0], self.nscan, self.ndet, self.ndet],
                       dtype="complex64")
        return self._batch(self.fwd, out, psi, scan, _single_mode(probe))

    def adj_ptycho_batch(self, farplane, scan, probe):
        out = np.zeros([scan.shape[0], self.nz, self.n], dtype="complex64")
        return self._batch(self.adj, out, farplane, scan, _single_mode(probe))

    def adj_ptycho_batch_prb(self, farplane, scan, psi):
        out = np.zeros([scan.shape[0], self.nprb, self.nprb], dtype="complex64")
        return self._batch(self.adj_probe, out, farplane, scan, psi)

    def run_batch(self, data, psi, scan, probe, **kwargs):
        """``ptycho.py:135-162``; scan updates are not returned, remainder
        angles are dropped, like the reference."""
        assert probe.ndim == 4, "probe needs 4 dimensions, not %d" % probe.ndim
        psi = psi.copy()
        probe = probe.copy()
        for k in range(scan.shape[0] // self.ptheta):
            ids = np.arange(k * self.ptheta, (k + 1) * self.ptheta)
            res = self.run(np.array(data[ids]), np.array(psi[ids]),
                           np.array(scan[ids]), np.array(probe[ids]), **kwargs)
            psi[ids], probe[ids] = res["psi"], res["probe"]
        return {"psi": psi, "probe": probe}

    # -- CG (ptycho.py:283-488) ----------------------------------------------
    def run(self, data, psi, scan, probe, piter, model="gaussian",
            recover_prb=False, ortho_prb=False, verbose=False):
        assert probe.ndim == 4, "probe needs 4 dimensions, not %d" % probe.ndim
        nmodes = probe.shape[1]

        def minf(x):
            if model == "gaussian":
                return np.linalg.norm(np.sqrt(np.abs(x)) - np.sqrt(data)) ** 2
            elif model == "poisson":
                return np.sum(np.abs(x) - data * np.log(np.abs(x) + 1e-32))

        def intensity(obj):
            acc = data * 0
            for k in range(nmodes):
                acc += np.abs(self.fwd(obj, scan, probe[:, k])) ** 2
            return acc

        dpsi = gradpsi0 = 0
        dprb = gradprb0 = 0
        gammaprb = 0
        for i in range(piter):
            # object step -- :325-405
            absfpsi = intensity(psi)
            a = np.sum(np.sqrt(absfpsi * data))
            b = np.sum(absfpsi)
            probe *= (a / b)
            absfpsi *= (a / b) ** 2
            gradpsi = np.zeros([self.ptheta, self.nz, self.n], dtype="complex64")
            if model == "gaussian":
                for k in range(nmodes):
                    fpsi = self.fwd(psi, scan, probe[:, k]) * (b / a)
                    gradpsi += self.adj(
                        fpsi - np.sqrt(data) * fpsi / (np.sqrt(absfpsi) + 1e-32),
                        scan, probe[:, k]) / (np.max(np.abs(probe[:, k])) ** 2)
            elif model == "poisson":
                for k in range(nmodes):
                    gradpsi += self.adj(
                        fpsi - data * fpsi / (absfpsi + 1e-32),   # noqa: F821
                        scan, probe[:, k]) / (np.max(np.abs(probe[:, k])) ** 2)
            if i == 0:
                dpsi = -gradpsi
            else:
                dpsi = -gradpsi + (
                    np.linalg.norm(gradpsi) ** 2
                    / (np.sum(np.conj(dpsi) * (gradpsi - gradpsi0))) * dpsi)
            gradpsi0 = gradpsi
            p1, p2, p3 = data * 0, data * 0, data * 0
            for k in range(nmodes):
                t1 = self.fwd(psi, scan, probe[:, k])
                t2 = self.fwd(dpsi, scan, probe[:, k])
                p1 += np.abs(t1) ** 2
                p2 += np.abs(t2) ** 2
                p3 += 2 * (t1.real * t2.real + t1.imag * t2.imag)
            gammapsi = 0.5 * line_search_sqr(minf, p1, p2, p3)
            if i > 0:                                   # :398-403
                ones = probe[:, 0] * 0 + 1
                t1 = self.fwd(psi, scan, ones)[0]
                t2 = self.fwd(psi + gammapsi * dpsi, scan, ones)[0]
                shifts = register_translation_batch(
                    t1, t2, upsample_factor=100, space="fourier")
                scan[0, :] += shifts
            psi = psi + gammapsi * dpsi

            if recover_prb:                             # :409-465
                if i == 0:
                    gradprb = probe * 0
                    gradprb0 = probe * 0
                    dprb = probe * 0
                for m in range(nmodes):
                    fprb = self.fwd(psi, scan, probe[:, m])
                    absfprb = intensity(psi)
                    if model == "gaussian":
                        gradprb[:, m] = self.adj_probe(
                            fprb - np.sqrt(data) * fprb / (np.sqrt(absfprb) + 1e-32),
                            scan, psi,
                        ) / np.max(np.abs(psi)) ** 2 / self.nscan * nmodes
                    elif model == "poisson":
                        gradprb[:, m] = self.adj_probe(
                            fprb - data * fprb / (absfprb + 1e-32), scan, psi,
                        ) / np.max(np.abs(psi)) ** 2 / self.nscan
                    if i == 0:
                        dprb[:, m] = -gradprb[:, m]
                    else:
                        dprb[:, m] = -gradprb[:, m] + (
                            np.linalg.norm(gradprb[:, m]) ** 2
                            / (np.sum(np.conj(dprb[:, m])
                                      * (gradprb[:, m] - gradprb0[:, m])))
                            * dprb[:, m])
                    gradprb0[:, m] = gradprb[:, m]
                    p1 = intensity(psi)
                    t1 = self.fwd(psi, scan, probe[:, m])
                    t2 = self.fwd(psi, scan, dprb[:, m])
                    p2 = np.abs(t2) ** 2
                    p3 = 2 * (t1.real * t2.real + t1.imag * t2.imag)
                    gammaprb = 0.5 * line_search_sqr(minf, p1, p2, p3,
                                                     step_length=1)
                    probe[:, m] = probe[:, m] + gammaprb * dprb[:, m]

            cost = float(minf(absfpsi))      # start-of-iteration value, :480-482
            self.history.append((i, float(gammapsi), float(gammaprb), cost))
            if verbose and i % 32 == 0:
                print("%4d, %.3e, %.3e, %.7e" % self.history[-1])
        return {"psi": psi, "probe": probe}


def _single_mode(probe):
    """A ``[ntheta,1,nprb,nprb]`` probe handed to the ``*_batch`` wrappers keeps
    its memory layout through ``x[ids:ids+1]`` (``ptycho.py:76``) and is read as
    ``[1,nprb,nprb]`` by the native side (``tests/test_adjoint.py:24,44``)."""
    probe = np.asarray(probe)
    if probe.ndim == 4:
        assert probe.shape[1] == 1, "the *_batch wrappers take one probe mode"
        return probe[:, 0]
    return probe
