"""Size-independent properties of the HIP operators on randomly drawn geometries
(hypothesis): linearity in the object and in the probe, the adjoint identity for both
adjoints, and agreement with the oracle -- for random ndet / nprb / object sizes / scan
positions, including padded probes, negative (skipped) and edge-overhanging positions."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import ptycho_oracle as op

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    return pt


def dev(x):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda")


def host(x):
    return x.detach().cpu().numpy()


def crand(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


@st.composite
def geometry(draw):
    ndet = draw(st.sampled_from([16, 32, 64, 128, 12, 24, 48, 100]))   # powers of two: fused kernels; others: Bluestein path
    nprb = draw(st.integers(min_value=max(1, ndet // 4), max_value=ndet))
    ntheta = draw(st.integers(1, 2))
    nscan = draw(st.integers(1, 12))
    nz = nprb + draw(st.integers(2, 40))
    n = nprb + draw(st.integers(2, 40))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return ndet, nprb, ntheta, nscan, nz, n, seed


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(geometry())
def test_linearity_adjointness_and_oracle(pt, g):
    ndet, nprb, ntheta, nscan, nz, n, seed = g
    rng = np.random.default_rng(seed)
    scan = np.empty((ntheta, nscan, 2), np.float32)
    scan[..., 0] = rng.uniform(-2, nz - nprb + 3, (ntheta, nscan))   # some negative, some overhanging
    scan[..., 1] = rng.uniform(-2, n - nprb + 3, (ntheta, nscan))
    psi1, psi2 = crand(rng, (ntheta, nz, n)), crand(rng, (ntheta, nz, n))
    prb1, prb2 = crand(rng, (ntheta, nprb, nprb)), crand(rng, (ntheta, nprb, nprb))
    y = crand(rng, (ntheta, nscan, ndet, ndet))
    a, b = np.complex64(0.7 - 0.2j), np.complex64(-1.3 + 0.5j)
    with pt.PtychoCuFFT(nscan, nprb, ndet, ntheta, nz, n) as slv:
        f = lambda ps, pr: host(slv.fwd(dev(ps), dev(scan), dev(pr)))
        g11 = f(psi1, prb1)
        scale = np.abs(g11).max() + 1e-20
        # linear in the object for a fixed probe, and in the probe for a fixed object
        assert np.abs(f(a * psi1 + b * psi2, prb1) - (a * g11 + b * f(psi2, prb1))).max() < 2e-5 * (scale + np.abs(f(psi2, prb1)).max())
        assert np.abs(f(psi1, a * prb1 + b * prb2) - (a * g11 + b * f(psi1, prb2))).max() < 2e-5 * (scale + np.abs(f(psi1, prb2)).max())
        aty = host(slv.adj(dev(y), dev(scan), dev(prb1)))
        bty = host(slv.adj_probe(dev(y), dev(scan), dev(psi1)))
    lhs = np.vdot(y.astype(np.complex128), g11.astype(np.complex128))
    r1 = np.vdot(aty.astype(np.complex128), psi1.astype(np.complex128))
    r2 = np.vdot(bty.astype(np.complex128), prb1.astype(np.complex128))
    nrm = np.linalg.norm(g11) * np.linalg.norm(y) + 1e-20
    assert abs(lhs - r1) < 2e-6 * nrm and abs(lhs - r2) < 2e-6 * nrm
    want = op.fwd(psi1, scan, prb1, ndet, "double")
    assert np.abs(g11 - want).max() <= 2e-5 * (np.abs(want).max() + 1e-20)


def test_poisson_model_is_broken_like_the_reference(pt):
    """ptycho.py:358-363 reads fpsi before assignment at i == 0; the loop here raises too."""
    import torch
    with pt.CGPtychoSolver(4, 16, 16, 1, 40, 40) as slv:
        slv.verbose = False
        z = lambda *s, dt=torch.complex64: torch.ones(s, dtype=dt, device="cuda")
        with pytest.raises(UnboundLocalError):
            slv.run(z(1, 4, 16, 16, dt=torch.float32), z(1, 40, 40), z(1, 4, 2, dt=torch.float32),
                    z(1, 1, 16, 16), piter=1, model="poisson")


@pytest.mark.gpu
@pytest.mark.parametrize("npos,ndet", [(37, 64), (300, 128), (2048, 256)])
def test_cross_workgroup_sums_are_exact_and_repeatable(npos, ndet):
    """The fixed-order fold across workgroups (a, b of ptycho.py:342-343 through ``ptycho_cg_stats``): equal to a float64
    torch reduction of the same farplane to 1e-6, and the same bits on every one of 100 repeats with other kernels in
    between (a stale partial row or a lost ticket would show as a different sum)."""
    import ctypes
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    from libtike.hipfft import _native as nat, synthetic as syn
    from libtike.hipfft.ptycho import _ptr, _stream
    R = int(np.ceil(np.sqrt(npos)))
    p = syn.make_problem(R, R, 3, ndet, ndet, seed=npos)
    scan = np.ascontiguousarray(p["scan"][:, :npos])
    rng = np.random.default_rng(npos)
    prb = (p["probe"] * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    with pt.CGPtychoSolver(npos, ndet, ndet, 1, p["nz"], p["n"]) as slv:
        psi, scan_d, prb_d = dev(p["psi"]), dev(scan), dev(prb)
        g = slv.fwd(psi, scan_d, prb_d)
        inten = (g.real.double() ** 2 + g.imag.double() ** 2)
        data = (torch.abs(g) ** 2 * (0.5 + torch.rand(g.shape, device="cuda"))).float().contiguous()
        want = torch.stack((torch.sqrt(inten.float().double() * data.double()).sum(), inten.sum())).cpu().numpy()
        slv._note_scan(scan_d)
        nat.check(nat.cg_fwd_cols(slv._h, 0, _ptr(psi), _ptr(scan_d), _ptr(prb_d), _stream()))
        sums = torch.zeros(2, dtype=torch.float64, device="cuda")
        seen = set()
        for rep in range(100):
            sums.zero_()
            nat.check(nat.cg_stats(slv._h, 0, _ptr(data), _ptr(sums), _stream()))
            if rep % 3 == 0:     # other work in between: the row pass of an operator call uses the same CUs
                slv.fwd(psi, scan_d, prb_d, out=g)
            seen.add(tuple(sums.cpu().numpy().tolist()))
        assert len(seen) == 1, seen
        got = np.array(next(iter(seen)))
        assert np.all(np.abs(got - want) <= 2e-6 * np.abs(want)), (got, want)
