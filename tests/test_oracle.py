"""Pin the CPU oracle (``oracle/``) before anything is compared against it.

* the reference's own numerical check -- the adjoint identities of
  ``/root/reference/tests/test_adjoint.py:42-56`` on the reference's fixtures;
* analytic known answers that do not depend on the restatement;
* a scalar, thread-by-thread transcription of ``muloperator``'s arithmetic
  (``/root/reference/src/cuda/kernels.cu:8-108``) on tiny sizes.
"""
import numpy as np
import pytest

from oracle import ptycho_oracle as op
from oracle import cg_oracle as cg
from libtike.hipfft import synthetic as syn


def vdot(a, b):
    return np.vdot(b.astype(np.complex128), a.astype(np.complex128))  # sum a conj(b)


def small_problem(ny=4, nx=5, step=5, nprb=16, ndet=16, seed=3, ntheta=1):
    p = syn.make_problem(ny, nx, step, nprb, ndet, ntheta=ntheta, seed=seed)
    return p


# ---------------------------------------------------------------------------
# scalar transcription of muloperator (pure Python loops, tiny sizes only)
# ---------------------------------------------------------------------------
def muloperator_scalar(f, g, prb, scan, ndet, flg):
    """One 'thread' per (tz, ty, tx) exactly as kernels.cu:13-107, float32."""
    f32 = np.float32
    ntheta, nz, n = f.shape
    nscan = scan.shape[1]
    nprb = prb.shape[-1]
    c = f32(1.0) / f32(ndet)
    pad = (ndet - nprb) // 2
    for tz in range(ntheta):
        for ty in range(nscan):
            sx = np.trunc(scan[tz, ty, 1]); sxf = f32(scan[tz, ty, 1] - sx)
            sy = np.trunc(scan[tz, ty, 0]); syf = f32(scan[tz, ty, 0] - sy)
            if sx < 0 or sy < 0:
                continue
            sx, sy = int(sx), int(sy)
            w = [(f32(1) - sxf, f32(1) - syf), (sxf, f32(1) - syf),
                 (f32(1) - sxf, syf), (sxf, syf)]
            for tx in range(nprb * nprb):
                ix, iy = tx % nprb, tx // nprb
                taps = [(sy + iy, sx + ix), (sy + iy, sx + ix + 1),
                        (sy + iy + 1, sx + ix), (sy + iy + 1, sx + ix + 1)]
                gi = (pad + iy, pad + ix)
                if flg == 2 or flg == 1:
                    tmp = np.complex64(0)
                    for (yy, xx), (wa, wb) in zip(taps, w):
                        tmp = np.complex64(tmp + np.complex64(f[tz, yy, xx] * wa) * wb)
                if flg == 2:
                    g[tz, ty][gi] = c * (prb[tz, iy, ix] * tmp)
                elif flg == 1:
                    prb[tz, iy, ix] += c * (g[tz, ty][gi] * np.conj(tmp))
                else:
                    tmp = c * (np.conj(prb[tz, iy, ix]) * g[tz, ty][gi])
                    for (yy, xx), (wa, wb) in zip(taps, w):
                        f[tz, yy, xx] += np.complex64(tmp * wa) * wb


@pytest.mark.parametrize("nprb,ndet", [(6, 6), (4, 8)])
def test_vectorised_oracle_matches_scalar_muloperator(nprb, ndet):
    rng = np.random.default_rng(0)
    ntheta, nz, n, nscan = 2, 14, 17, 5
    psi = (rng.standard_normal((ntheta, nz, n)) + 1j * rng.standard_normal((ntheta, nz, n))).astype(np.complex64)
    prb = (rng.standard_normal((ntheta, nprb, nprb)) + 1j * rng.standard_normal((ntheta, nprb, nprb))).astype(np.complex64)
    scan = (rng.random((ntheta, nscan, 2)) * [nz - nprb - 1, n - nprb - 1]).astype(np.float32)
    scan[0, 1] = [-1.5, 2.0]          # skipped: negative integer part
    scan[1, 2] = [3.0, 4.0]           # exactly integer
    # forward
    near = np.zeros((ntheta, nscan, ndet, ndet), np.complex64)
    muloperator_scalar(psi, near, prb, scan, ndet, 2)
    np.testing.assert_allclose(op.nearplane(psi, scan, prb, ndet), near, rtol=2e-6, atol=1e-7)
    g = np.fft.fft2(near.astype(np.complex128))
    np.testing.assert_allclose(op.fwd(psi, scan, prb, ndet), g, rtol=0, atol=2e-5 * np.abs(g).max())
    # adjoints: start from a random farplane
    far = (rng.standard_normal(near.shape) + 1j * rng.standard_normal(near.shape)).astype(np.complex64)
    inv = (np.fft.ifft2(far.astype(np.complex128)) * ndet * ndet).astype(np.complex64)
    f_acc = np.zeros_like(psi)
    muloperator_scalar(f_acc, inv.copy(), prb, scan, ndet, 0)
    got = op.adj(far, scan, prb, nz, n)
    np.testing.assert_allclose(got, f_acc, rtol=0, atol=3e-5 * np.abs(f_acc).max())
    p_acc = np.zeros_like(prb)
    muloperator_scalar(psi, inv.copy(), p_acc, scan, ndet, 1)
    got = op.adj_probe(far, scan, psi, nprb)
    np.testing.assert_allclose(got, p_acc, rtol=0, atol=3e-5 * np.abs(p_acc).max())


# ---------------------------------------------------------------------------
# the reference's own check, on the reference's own fixtures
# ---------------------------------------------------------------------------
def reference_adjoint_inputs(model, nscan=100):
    # /root/reference/tests/test_adjoint.py:15-40
    n, nz, nprb = 600, 276, 128
    prb0 = np.zeros([1, 1, nprb, nprb], dtype="complex64")
    prb0[0] = model["prbamp"] * np.exp(1j * model["prbang"])
    scan = np.ones([1, nscan, 2], dtype="float32")
    temp = np.moveaxis(model["coords"], 0, 1)[:nscan]
    scan[0, :, 0] = temp[:, 1]
    scan[0, :, 1] = temp[:, 0]
    psi0 = np.ones([1, nz, n], dtype="complex64")
    psi0[0] = model["initpsiamp"] * np.exp(1j * model["initpsiang"])
    return psi0, scan, prb0


def test_reference_adjoint_script_on_reference_fixtures(model):
    psi0, scan, prb0 = reference_adjoint_inputs(model)
    slv = cg.OracleSolver(100, 128, 128, 1, 276, 600)
    t1 = slv.fwd_ptycho_batch(psi0, scan, prb0)
    t2 = slv.adj_ptycho_batch(t1, scan, prb0)
    t3 = slv.adj_ptycho_batch_prb(t1, scan, psi0)
    a = np.sum(psi0 * np.conj(t2))
    b = np.sum(t1 * np.conj(t1))
    c = np.sum(prb0 * np.conj(t3))
    # the reference's acceptance line, tests/test_adjoint.py:56 ...
    assert ((a - b) / a < 1e-3) & ((a - c) / a < 1e-3)
    # ... and a two-sided version with float64 accumulation
    a, b, c = vdot(psi0, t2), vdot(t1, t1), vdot(prb0[:, 0], t3)
    assert abs(a - b) / abs(a) < 1e-5
    assert abs(a - c) / abs(a) < 1e-5


# ---------------------------------------------------------------------------
# analytic known answers
# ---------------------------------------------------------------------------
def test_flat_object_gives_fft_of_padded_probe():
    p = small_problem(nprb=12, ndet=16)
    psi = np.ones_like(p["psi"])
    got = op.fwd(psi, p["scan"], p["probe"], 16, "double")
    pad = np.zeros((16, 16), complex)
    pad[2:14, 2:14] = p["probe"][0]
    want = np.fft.fft2(pad) / 16
    for s in range(p["nscan"]):
        np.testing.assert_allclose(got[0, s], want, atol=1e-12)


def test_integer_positions_unit_probe():
    p = small_problem()
    scan = np.floor(p["scan"])
    prb = np.ones_like(p["probe"])
    got = op.fwd(p["psi"], scan, prb, 16, "double")
    for s in range(p["nscan"]):
        y, x = int(scan[0, s, 0]), int(scan[0, s, 1])
        want = np.fft.fft2(p["psi"][0, y:y + 16, x:x + 16].astype(complex)) / 16
        np.testing.assert_allclose(got[0, s], want, atol=1e-12)


def test_parseval():
    p = small_problem(nprb=12, ndet=16)
    g = op.fwd(p["psi"], p["scan"], p["probe"], 16, "double")
    q = p["probe"][:, None] * op.patches(p["psi"], p["scan"], 12, "double")
    assert abs(np.sum(np.abs(g) ** 2) - np.sum(np.abs(q) ** 2)) < 1e-9 * np.sum(np.abs(q) ** 2)


@pytest.mark.parametrize("nprb,ndet", [(16, 16), (10, 16)])
@pytest.mark.parametrize("precision,tol", [("double", 1e-12), ("single", 1e-5)])
def test_adjoint_identity_independent_y(nprb, ndet, precision, tol):
    p = small_problem(nprb=nprb, ndet=ndet, ntheta=2)
    rng = np.random.default_rng(5)
    y = (rng.standard_normal((2, p["nscan"], ndet, ndet))
         + 1j * rng.standard_normal((2, p["nscan"], ndet, ndet))).astype(np.complex64)
    Ax = op.fwd(p["psi"], p["scan"], p["probe"], ndet, precision)
    Aty = op.adj(y, p["scan"], p["probe"], p["nz"], p["n"], precision)
    Bty = op.adj_probe(y, p["scan"], p["psi"], nprb, precision)
    lhs = vdot(Ax, y)
    assert abs(lhs - vdot(p["psi"], Aty)) / abs(lhs) < tol
    assert abs(lhs - vdot(p["probe"], Bty)) / abs(lhs) < tol


def test_negative_positions_are_skipped_and_oob_taps_read_zero():
    p = small_problem()
    scan = p["scan"].copy()
    scan[0, 0] = [-2.25, 3.0]
    scan[0, 1] = [-0.25, 3.5]            # trunc -> -0.0: NOT skipped (kernels.cu:39)
    g = op.fwd(p["psi"], scan, p["probe"], 16)
    assert np.all(g[0, 0] == 0)
    assert np.abs(g[0, 1]).max() > 0
    # a patch hanging over the object edge reads zeros there
    scan[0, 2] = [p["nz"] - 10.0, 2.0]
    big = np.zeros((1, p["nz"] + 32, p["n"]), np.complex64)
    big[:, :p["nz"]] = p["psi"]
    np.testing.assert_array_equal(op.fwd(p["psi"], scan, p["probe"], 16)[0, 2],
                                  op.fwd(big, scan, p["probe"], 16)[0, 2])


def test_single_matches_double():
    p = small_problem(nprb=12, ndet=16)
    a = op.fwd(p["psi"], p["scan"], p["probe"], 16, "single")
    b = op.fwd(p["psi"], p["scan"], p["probe"], 16, "double")
    assert a.dtype == np.complex64 and b.dtype == np.complex128
    assert np.abs(a - b).max() < 5e-6 * np.abs(b).max()


# ---------------------------------------------------------------------------
# CG loop + registration
# ---------------------------------------------------------------------------
def test_registration_recovers_known_shift():
    rng = np.random.default_rng(0)
    img = rng.standard_normal((3, 32, 32))
    f = np.fft.fft2(img)
    ky = np.fft.fftfreq(32)[:, None]
    kx = np.fft.fftfreq(32)[None, :]
    true = np.array([[1.25, -2.5], [0.0, 0.37], [-3.0, 4.11]])
    moved = np.stack([f[i] * np.exp(-2j * np.pi * (ky * true[i, 0] + kx * true[i, 1]))
                      for i in range(3)])
    got = cg.register_translation_batch(f, moved, 100, "fourier")
    # shift needed to register `moved` onto `f` is -true
    np.testing.assert_allclose(got, -true, atol=0.011)
    # one-pattern quirk (ptycho.py:243-245)
    assert np.all(cg.register_translation_batch(f[:1], moved[:1], 100, "fourier") == 0)


def test_line_search():
    f = lambda x: float(np.sum((x - 1.0) ** 2))
    # cost along gamma: (p1 + g^2 p2 + g p3 - 1)^2
    assert cg.line_search_sqr(f, np.array([2.0]), np.array([1.0]), np.array([-1.0])) == 1
    assert cg.line_search_sqr(f, np.array([1.5]), np.array([4.0]), np.array([-1.0])) == 0.25
    with pytest.warns(UserWarning):
        assert cg.line_search_sqr(lambda x: float(np.abs(x).sum()), np.array([0.0]),
                                  np.array([1.0]), np.array([1.0])) == 0


def cg_problem(nmodes=1, seed=7):
    p = syn.make_problem(6, 6, 4, 16, 16, seed=seed)
    probe = syn.hermite_modes(16, nmodes) if nmodes > 1 else p["probe"][:, None]
    slv = cg.OracleSolver(p["nscan"], 16, 16, 1, p["nz"], p["n"])
    data = np.zeros((1, p["nscan"], 16, 16), np.float32)
    for k in range(probe.shape[1]):
        data += np.abs(slv.fwd(p["psi"], p["scan"], probe[:, k])) ** 2
    return p, probe.astype(np.complex64), slv, data


def test_cg_gradient_vanishes_at_truth_and_cost_decreases():
    p, probe, slv, data = cg_problem()
    # at the true object: first-iteration gradient ~ 0 -> psi unchanged
    res = slv.run(data.copy(), p["psi"].copy(), p["scan"].copy(), probe.copy(), piter=1)
    assert np.abs(res["psi"] - p["psi"]).max() < 1e-4
    slv.history.clear()
    psi0 = np.ones_like(p["psi"])
    res = slv.run(data.copy(), psi0, p["scan"].copy(), probe.copy(), piter=12)
    cost = [h[3] for h in slv.history]
    assert cost[-1] < 0.2 * cost[0]
    assert all(c1 <= c0 * 1.0001 for c0, c1 in zip(cost, cost[1:]))


def test_cg_multimode_with_probe_recovery_runs_and_descends():
    p, probe, slv, data = cg_problem(nmodes=2)
    start = probe.copy().swapaxes(2, 3)
    res = slv.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start,
                  piter=6, recover_prb=True)
    cost = [h[3] for h in slv.history]
    assert cost[-1] < cost[0]
    assert res["probe"].shape == probe.shape


def test_cg_poisson_is_broken_like_the_reference():
    p, probe, slv, data = cg_problem()
    with pytest.raises(UnboundLocalError):
        slv.run(data, np.ones_like(p["psi"]), p["scan"], probe, piter=1, model="poisson")


def test_run_batch_shapes():
    p, probe, slv, data = cg_problem()
    out = slv.run_batch(data, np.ones_like(p["psi"]), p["scan"], probe, piter=2)
    assert out["psi"].shape == p["psi"].shape and out["probe"].shape == probe.shape
