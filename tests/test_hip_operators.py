"""Parity of the HIP operators (through the C ABI, via ``libtike.hipfft``) with the
CPU oracle.  Everything here needs a real MI355X: ``pytest -m gpu``.

Tolerances (float32 arithmetic on the device, oracle evaluated in float64):
``REL_MAX`` bounds max|got-want| / max|want|, ``REL_L2`` bounds the relative L2
error.  The adjoint identity is held to 1e-5 (BASELINE.json), accumulated in
float64.
"""
import numpy as np
import pytest

from oracle import ptycho_oracle as op
from libtike.hipfft import synthetic as syn

pytestmark = pytest.mark.gpu

REL_MAX = 2e-5
REL_L2 = 3e-6


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


def err(got, want):
    d = np.abs(got.astype(np.complex128) - want)
    return d.max() / np.abs(want).max(), np.sqrt((d ** 2).sum() / (np.abs(want) ** 2).sum())


def vdot(a, b):
    return np.vdot(b.astype(np.complex128), a.astype(np.complex128))


def problem(ndet, nprb, ny, nx, step, ntheta=1, seed=11, hard=True):
    p = syn.make_problem(ny, nx, step, nprb, ndet, ntheta=ntheta, seed=seed)
    rng = np.random.default_rng(seed + 1)
    p["probe"] = (p["probe"] * np.exp(2j * np.pi * rng.random(p["probe"].shape))).astype(np.complex64)
    if hard and p["nscan"] >= 4:
        p["scan"][0, 1] = [-1.5, 2.25]                       # skipped (kernels.cu:39)
        p["scan"][0, 2] = [-0.25, 1.5]                       # trunc -> -0.0, not skipped
        p["scan"][-1, 3] = [p["nz"] - nprb / 2, 3.75]        # hangs over the bottom edge
        p["scan"][0, 0] = [2.0, 5.0]                         # exactly integer
    return p


CASES = [
    # ndet, nprb, ny, nx, step, ntheta
    (16, 16, 3, 3, 5, 1),
    (16, 12, 3, 3, 5, 2),
    (32, 32, 3, 4, 6, 1),
    (32, 21, 2, 3, 6, 1),
    (64, 64, 4, 4, 9, 2),
    (64, 40, 3, 3, 9, 1),
    (128, 128, 4, 5, 11, 1),
    (128, 100, 3, 3, 11, 1),
    # sizes with an odd prime factor and a Stockham plan of their own (radix 3 / 5 / 7 first, fft_core.hpp)
    (48, 48, 3, 4, 7, 1),
    (48, 30, 3, 3, 7, 2),
    (80, 80, 3, 3, 9, 1),
    (80, 55, 2, 3, 9, 1),
    (96, 96, 3, 4, 9, 2),
    (96, 70, 3, 3, 9, 1),
    (112, 112, 3, 4, 11, 1),
    (112, 75, 3, 3, 11, 2),
    (192, 192, 2, 3, 13, 1),
    (192, 130, 2, 2, 13, 2),
    (256, 256, 3, 4, 13, 1),
    (256, 128, 2, 3, 13, 2),
    (512, 512, 2, 2, 17, 1),
    (512, 300, 2, 2, 17, 1),
    (1024, 1024, 1, 2, 17, 1),
    (2048, 2048, 1, 2, 19, 1),
    (2048, 1500, 1, 2, 19, 1),
]


@pytest.mark.parametrize("ndet,nprb,ny,nx,step,ntheta", CASES)
def test_fwd_adj_adjprobe_match_oracle(pt, ndet, nprb, ny, nx, step, ntheta):
    p = problem(ndet, nprb, ny, nx, step, ntheta)
    rng = np.random.default_rng(2)
    with pt.PtychoCuFFT(p["nscan"], nprb, ndet, ntheta, p["nz"], p["n"]) as slv:
        assert (slv.ptheta, slv.nz, slv.n, slv.nscan, slv.ndet, slv.nprb) == \
            (ntheta, p["nz"], p["n"], p["nscan"], ndet, nprb)
        psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
        want_g = op.fwd(p["psi"], p["scan"], p["probe"], ndet, "double")
        y = (rng.standard_normal(want_g.shape) + 1j * rng.standard_normal(want_g.shape)).astype(np.complex64)
        want_a = op.adj(y, p["scan"], p["probe"], p["nz"], p["n"], "double")
        want_p = op.adj_probe(y, p["scan"], p["psi"], nprb, "double")
        # ndet <= 128 has two paths: one launch with the tile in LDS (default) and the two-pass kernels
        for tile in ((True, False) if ndet <= 128 else (True,)):
            slv.set_tile(tile)
            g = host(slv.fwd(psi, scan, prb))
            e = err(g, want_g)
            assert e[0] < REL_MAX and e[1] < REL_L2, ("fwd", tile, e)
            if p["nscan"] >= 4:
                assert np.all(g[0, 1] == 0)    # skipped position -> exact zeros
            e = err(host(slv.adj(dev(y), scan, prb)), want_a)
            assert e[0] < REL_MAX and e[1] < REL_L2, ("adj", tile, e)
            e = err(host(slv.adj_probe(dev(y), scan, psi)), want_p)
            assert e[0] < REL_MAX and e[1] < REL_L2, ("adj_probe", tile, e)


@pytest.mark.parametrize("ndet,nprb,ny,nx", [(16, 16, 151, 65), (32, 27, 67, 39), (64, 64, 41, 19), (128, 100, 17, 9),
                                               (48, 40, 50, 21), (80, 80, 24, 11), (96, 96, 20, 9), (112, 90, 19, 9)])
def test_tile_kernels_equal_the_two_pass_kernels(pt, ndet, nprb, ny, nx):
    """ndet <= 128, more positions than one trip of the persistent workgroups holds, a count that does not fill the last
    workgroup, two angles (the probe adjoint's accumulators are flushed at the angle change), padded probes, skipped and
    overhanging positions: the one-launch kernels (k_tile.hpp) against the two-pass kernels, which the oracle test above
    checks on small cases."""
    p = problem(ndet, nprb, ny, nx, 3, ntheta=2)
    rng = np.random.default_rng(5)
    ns = p["nscan"]
    y = (rng.standard_normal((2, ns, ndet, ndet)) + 1j * rng.standard_normal((2, ns, ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(ns, nprb, ndet, 2, p["nz"], p["n"]) as slv:
        psi, scan, prb, yd = dev(p["psi"]), dev(p["scan"]), dev(p["probe"]), dev(y)
        slv.set_tile(False)
        ref = [host(slv.fwd(psi, scan, prb)), host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        slv.set_tile(True)
        slv.profile(True)
        got = [host(slv.fwd(psi, scan, prb)), host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        ran = slv.profile_read()
        slv.profile(False)
    assert "k_fwd_tile" in ran and "k_adjprb_tile" in ran, ran      # the paths under test did run
    for name, a, b in zip(("fwd", "adj", "adj_probe"), got, ref):
        e = err(a, b.astype(np.complex128))
        assert e[0] < 2e-5 and e[1] < 2e-6, (name, e)
    assert np.all(got[0][0, 1] == 0)        # skipped position -> exact zeros


def test_chunking_does_not_change_results(pt):
    p = problem(64, 48, 5, 5, 7, 2)
    with pt.PtychoCuFFT(p["nscan"], 48, 64, 2, p["nz"], p["n"]) as slv:
        psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
        g0 = host(slv.fwd(psi, scan, prb))
        a0 = host(slv.adj(dev(g0), scan, prb))
        slv.set_chunk(7)                   # ragged chunks crossing the angle boundary
        g1 = host(slv.fwd(psi, scan, prb))
        a1 = host(slv.adj(dev(g0), scan, prb))
        b1 = host(slv.adj_probe(dev(g0), scan, psi))
        slv.set_chunk(0)
        b0 = host(slv.adj_probe(dev(g0), scan, psi))
    np.testing.assert_array_equal(g0, g1)
    assert err(a1, a0.astype(np.complex128))[0] < 1e-5      # atomics: order differs
    assert err(b1, b0.astype(np.complex128))[0] < 1e-5


@pytest.mark.parametrize("ndet,nprb,ntheta", [(64, 64, 2), (128, 100, 1), (256, 256, 1)])
def test_object_adjoint_any_scan_order(pt, ndet, nprb, ntheta):
    """The LDS overlap-add window must not depend on the scan being a sorted raster:
    random positions in random order (with skipped and overhanging ones), window on/off."""
    rng = np.random.default_rng(9)
    nscan, nz, n = 150, ndet + 90, ndet + 130
    scan = np.empty((ntheta, nscan, 2), np.float32)
    scan[..., 0] = rng.random((ntheta, nscan)) * (nz - nprb - 1)
    scan[..., 1] = rng.random((ntheta, nscan)) * (n - nprb - 1)
    scan[0, 5] = [-3.5, 4.0]
    scan[-1, 7] = [nz - nprb / 3, n - nprb / 2]
    scan[0, 9:12] = scan[0, 8]                      # repeated position
    prb = (rng.standard_normal((ntheta, nprb, nprb)) + 1j * rng.standard_normal((ntheta, nprb, nprb))).astype(np.complex64)
    y = (rng.standard_normal((ntheta, nscan, ndet, ndet)) + 1j * rng.standard_normal((ntheta, nscan, ndet, ndet))).astype(np.complex64)
    psi = (rng.standard_normal((ntheta, nz, n)) + 1j * rng.standard_normal((ntheta, nz, n))).astype(np.complex64)
    want = op.adj(y, scan, prb, nz, n, "double")
    want_f = op.fwd(psi, scan, prb, ndet, "double")
    want_p = op.adj_probe(y, scan, psi, nprb, "double")
    with pt.PtychoCuFFT(nscan, nprb, ndet, ntheta, nz, n) as slv:
        res = []
        for chunk, window in ((0, True), (37, True), (0, False)):
            slv.set_chunk(chunk)
            slv.set_window(window)
            res.append((host(slv.adj(dev(y), dev(scan), dev(prb))),
                        host(slv.fwd(dev(psi), dev(scan), dev(prb))),
                        host(slv.adj_probe(dev(y), dev(scan), dev(psi)))))
    for got, got_f, got_p in res:
        for gg, ww in ((got, want), (got_f, want_f), (got_p, want_p)):
            e = err(gg, ww)
            assert e[0] < REL_MAX and e[1] < REL_L2, e
        assert np.all(got_f[0, 5] == 0)


@pytest.mark.parametrize("case", ["one_position", "all_skipped", "one_pixel_probe", "flush_against_edges", "far_outside"])
def test_edge_cases(pt, case):
    rng = np.random.default_rng(3)
    ndet, nprb, nz, n, nscan = 16, 16, 40, 44, 3
    scan = np.array([[[3.25, 4.5], [10.0, 20.75], [17.5, 1.125]]], np.float32)
    if case == "one_position":
        nscan, scan = 1, scan[:, :1]
    elif case == "all_skipped":
        scan = -scan - 1.0
    elif case == "one_pixel_probe":
        nprb = 1
    elif case == "flush_against_edges":      # trunc(pos) + nprb + 1 == size exactly, and pos = 0
        scan = np.array([[[nz - nprb - 1.0, n - nprb - 1.0], [0.0, 0.0], [nz - nprb - 0.5, 0.25]]], np.float32)
    elif case == "far_outside":              # beyond the object: every tap reads zero / is dropped
        scan = np.array([[[1000.0, 3.0], [3.0, 5000.5], [2.0, 2.0]]], np.float32)
    psi = (rng.standard_normal((1, nz, n)) + 1j * rng.standard_normal((1, nz, n))).astype(np.complex64)
    prb = (rng.standard_normal((1, nprb, nprb)) + 1j * rng.standard_normal((1, nprb, nprb))).astype(np.complex64)
    y = (rng.standard_normal((1, nscan, ndet, ndet)) + 1j * rng.standard_normal((1, nscan, ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(nscan, nprb, ndet, 1, nz, n) as slv:
        g = host(slv.fwd(dev(psi), dev(scan), dev(prb)))
        a = host(slv.adj(dev(y), dev(scan), dev(prb)))
        b = host(slv.adj_probe(dev(y), dev(scan), dev(psi)))
    for got, want in ((g, op.fwd(psi, scan, prb, ndet, "double")),
                      (a, op.adj(y, scan, prb, nz, n, "double")),
                      (b, op.adj_probe(y, scan, psi, nprb, "double"))):
        scale = max(np.abs(want).max(), 1e-30)
        assert np.abs(got - want).max() <= REL_MAX * scale + 1e-30
    if case == "all_skipped":
        assert not g.any() and not a.any() and not b.any()


def test_optional_paths_give_the_same_results(pt):
    """ndet = 256 has two forward and two adjoint paths -- split two-pass (default) and unsplit two-pass; they must
    agree.  (The single-launch forward of the experiments build: tests/test_hip_experiments.py.)"""
    p = syn.make_problem(48, 48, 4, 256, 256, seed=2, nz=512, n=512)          # 2304 positions
    rng = np.random.default_rng(1)
    y = (rng.standard_normal((1, 2304, 256, 256)) + 1j * rng.standard_normal((1, 2304, 256, 256))).astype(np.complex64)
    with pt.PtychoCuFFT(2304, 256, 256, 1, 512, 512) as slv:
        psi, scan, prb, yd = dev(p["psi"]), dev(p["scan"]), dev(p["probe"]), dev(y)
        ref = [host(slv.fwd(psi, scan, prb)), host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        slv.set_split(False)               # unsplit column / row kernels
        uns = [host(slv.fwd(psi, scan, prb)), host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        slv.set_split(True)
        for a, b in zip(uns, ref):
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
        slv.release_scratch()              # the adjoint's intermediate is given back and allocated again on demand
        again = host(slv.adj(yd, scan, prb))
        assert np.abs(again - ref[1]).max() <= 1e-5 * np.abs(ref[1]).max()


@pytest.mark.parametrize("ndet", [100, 112])
def test_deterministic_adjoints_at_a_size_that_is_not_a_power_of_two(pt, ndet):
    """Option deterministic on the Bluestein path (ndet 100) and on a mixed-radix plan (112 = 7 x 4 x 4): both adjoints
    equal the float-atomic ones to rounding and are bitwise equal between two calls."""
    import torch
    p = syn.make_problem(12, 12, 6, ndet, ndet, seed=8)
    rng = np.random.default_rng(2)
    y = (rng.standard_normal((1, 144, ndet, ndet)) + 1j * rng.standard_normal((1, 144, ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(144, ndet, ndet, 1, p["nz"], p["n"]) as slv:
        psi, scan, prb, yd = dev(p["psi"]), dev(p["scan"]), dev(p["probe"]), dev(y)
        ref = [host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        slv.set_deterministic(True)
        a1 = [host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
        a2 = [host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))]
    for r, x1, x2 in zip(ref, a1, a2):
        assert np.array_equal(x1, x2)
        assert np.abs(x1 - r).max() <= 2e-6 * np.abs(r).max()


def test_fft2_matches_numpy(pt):
    rng = np.random.default_rng(0)
    # ndet <= 128: one launch with the tile in LDS (default) and the two-pass kernels; 1100 tiles of 64^2 are more than one
    # trip of the persistent workgroups
    for ndet, nb, tile in [(n, 3, True) for n in (16, 32, 48, 64, 80, 96, 112, 128, 192, 256, 512, 1024, 2048)] + \
                          [(n, 3, False) for n in (16, 32, 48, 64, 80, 96, 112, 128)] + [(64, 1100, True), (16, 37, True), (112, 600, True)]:
        x = (rng.standard_normal((nb, ndet, ndet)) + 1j * rng.standard_normal((nb, ndet, ndet))).astype(np.complex64)
        with pt.PtychoCuFFT(1, ndet, ndet, 1, ndet + 2, ndet + 2) as slv:
            slv.set_tile(tile)
            f = host(slv.fft2(dev(x)))
            b = host(slv.fft2(dev(x), inverse=True))
            xd = dev(x)
            inplace = host(slv.fft2(xd, out=xd))
        wf = np.fft.fft2(x.astype(np.complex128))
        wb = np.fft.ifft2(x.astype(np.complex128)) * ndet * ndet
        assert err(f, wf)[1] < REL_L2 and err(b, wb)[1] < REL_L2, (ndet, nb, tile)
        np.testing.assert_array_equal(inplace, f)


def test_reference_adjoint_script(pt, model):
    """/root/reference/tests/test_adjoint.py:15-59 through the *_batch API."""
    n, nz, nscan, nprb, ndet = 600, 276, 100, 128, 128
    prb0 = np.zeros([1, 1, nprb, nprb], dtype="complex64")
    prb0[0] = model["prbamp"] * np.exp(1j * model["prbang"])
    scan = np.ones([1, nscan, 2], dtype="float32")
    temp = np.moveaxis(model["coords"], 0, 1)[:nscan]
    scan[0, :, 0] = temp[:, 1]
    scan[0, :, 1] = temp[:, 0]
    psi0 = np.ones([1, nz, n], dtype="complex64")
    psi0[0] = model["initpsiamp"] * np.exp(1j * model["initpsiang"])
    with pt.PtychoCuFFT(nscan, nprb, ndet, 1, nz, n) as slv:
        t1 = slv.fwd_ptycho_batch(psi0, scan, prb0)
        t2 = slv.adj_ptycho_batch(t1, scan, prb0)
        t3 = slv.adj_ptycho_batch_prb(t1, scan, psi0)
    a = np.sum(psi0 * np.conj(t2))
    b = np.sum(t1 * np.conj(t1))
    c = np.sum(prb0 * np.conj(t3))
    assert ((a - b) / a < 1e-3) & ((a - c) / a < 1e-3)       # the reference's PASSED line
    a, b, c = vdot(psi0, t2), vdot(t1, t1), vdot(prb0[:, 0], t3)
    assert abs(a - b) / abs(a) < 1e-5 and abs(a - c) / abs(a) < 1e-5
    want = op.fwd(psi0, scan, prb0[:, 0], ndet, "double")
    e = err(t1, want)
    assert e[0] < REL_MAX and e[1] < REL_L2


@pytest.mark.parametrize("nprb", [256, 128])
def test_adjoint_identity_full_size(pt, nprb):
    """BASELINE.json config 2 geometry: 4096 x (256 x 256), <Ax,y> = <x,A*y> with
    an independent random y, both adjoints, relative residual < 1e-5."""
    import torch
    p = syn.make_problem(64, 64, 8, nprb, 256, seed=1234, nz=768, n=768)
    with pt.PtychoCuFFT(4096, nprb, 256, 1, 768, 768) as slv:
        psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
        gen = torch.Generator(device="cuda").manual_seed(5)
        y = torch.view_as_complex(torch.randn((1, 4096, 256, 256, 2), generator=gen,
                                              device="cuda", dtype=torch.float32))
        Ax = slv.fwd(psi, scan, prb)
        lhs = torch.sum(Ax.to(torch.complex128) * y.conj().to(torch.complex128))
        Aty = slv.adj(y, scan, prb)
        rhs1 = torch.sum(psi.to(torch.complex128) * Aty.conj().to(torch.complex128))
        Bty = slv.adj_probe(y, scan, psi)
        rhs2 = torch.sum(prb.to(torch.complex128) * Bty.conj().to(torch.complex128))
        # Parseval: ||Ax||^2 = ||prb * patch||^2 -- checked on a sample of positions
        sel = np.arange(0, 4096, 512)
        q = p["probe"][:, None] * op.patches(p["psi"], p["scan"][:, sel], nprb, "double")
        got = (torch.abs(Ax[0, sel].to(torch.complex128)) ** 2).sum(dim=(1, 2)).cpu().numpy()
        want = (np.abs(q[0]) ** 2).sum(axis=(1, 2))
        lhs, rhs1, rhs2 = complex(lhs), complex(rhs1), complex(rhs2)
    assert abs(lhs - rhs1) / abs(lhs) < 1e-5
    assert abs(lhs - rhs2) / abs(lhs) < 1e-5
    np.testing.assert_allclose(got, want, rtol=1e-5)


def test_error_behaviour(pt):
    import torch
    from libtike.hipfft import _native as nat
    with pytest.raises(nat.PtychoHipError):
        pt.PtychoCuFFT(4, 16, 1100, 1, 1200, 1200)    # ndet neither <= 1024 nor a power of two
    with pytest.raises(nat.PtychoHipError):
        pt.PtychoCuFFT(4, 32, 16, 1, 64, 64)          # nprb > ndet
    slv = pt.PtychoCuFFT(4, 16, 16, 1, 64, 64)
    psi = torch.ones((1, 64, 64), dtype=torch.complex64, device="cuda")
    scan = torch.ones((1, 4, 2), dtype=torch.float32, device="cuda")
    prb = torch.ones((1, 16, 16), dtype=torch.complex64, device="cuda")
    with pytest.raises(AssertionError):
        slv.fwd(psi.to(torch.complex128), scan, prb)  # dtype assert, ptycho.py:82-84
    with pytest.raises(ValueError):
        slv.fwd(psi[:, :32], scan, prb)               # wrong shape never reaches the kernel
    slv.fwd(psi, scan, prb)
    slv.free()
    slv.free()                                        # idempotent, ptychofft.cu:49-57
    with pytest.raises(nat.PtychoHipError):
        slv.fwd(psi, scan, prb)
    with pytest.raises(NotImplementedError):
        pt.PtychoCuFFT(4, 16, 16, 1, 64, 64).run(None, None, None, None)
    # options that a path does not cover fail loudly instead of being ignored
    with pt.PtychoCuFFT(4, 30, 30, 1, 64, 64) as gen:               # Bluestein path without its window: no fixed-point object adjoint
        gen.set_deterministic(True)
        gen.set_window(False)
        g = gen.fwd(psi, scan, torch.ones((1, 30, 30), dtype=torch.complex64, device="cuda"))
        with pytest.raises(nat.PtychoHipError):
            gen.adj(g, scan, torch.ones((1, 30, 30), dtype=torch.complex64, device="cuda"))


# ---- detector sizes that are not a power of two (cuFFT takes any size, ptychofft.cu:13-20) ----------------
@pytest.mark.parametrize("ndet", [12, 30, 48, 80, 96, 100, 112, 192, 200, 1000])   # 48, 80, 96, 112, 192: own plans; the others: Bluestein
def test_fft2_any_size_matches_numpy(pt, ndet):
    rng = np.random.default_rng(ndet)
    nb = 3 if ndet < 500 else 2
    x = (rng.standard_normal((nb, ndet, ndet)) + 1j * rng.standard_normal((nb, ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(nb, ndet, ndet, 1, ndet + 2, ndet + 2) as slv:
        got = host(slv.fft2(dev(x)))
        back = host(slv.fft2(dev(x), inverse=True))
    want = np.fft.fft2(x.astype(np.complex128))
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    wantb = np.fft.ifft2(x.astype(np.complex128)) * ndet * ndet          # unnormalised, like cufftExecC2C INVERSE
    assert np.abs(back - wantb).max() <= 2e-6 * np.abs(wantb).max()


@pytest.mark.parametrize("ndet,nprb,ntheta", [(112, 112, 1), (96, 64, 2), (100, 64, 2), (30, 17, 1), (200, 140, 1)])
def test_operators_any_detector_size_match_oracle(pt, ndet, nprb, ntheta):
    """/root/reference/tests/test_fsc.py:115-120 crops the detector and the probe from 128 to 112:
    fwd / adj / adj_probe at such sizes against the oracle, and the adjoint identity."""
    p = syn.make_problem(4, 5, 7, nprb, ndet, ntheta=ntheta, seed=ndet)
    rng = np.random.default_rng(1)
    scan = p["scan"].copy()
    scan[0, 0] = (-0.5, 1.0) if ndet in (96, 100) else scan[0, 0]   # -0.0 integer part: NOT skipped, negative fraction (kernels.cu:39)
    prb = (p["probe"] * np.exp(2j * np.pi * rng.random((nprb, nprb)))).astype(np.complex64)
    y = (rng.standard_normal((ntheta, p["nscan"], ndet, ndet)) + 1j * rng.standard_normal((ntheta, p["nscan"], ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(p["nscan"], nprb, ndet, ntheta, p["nz"], p["n"]) as slv:
        g = host(slv.fwd(dev(p["psi"]), dev(scan), dev(prb)))
        a = host(slv.adj(dev(y), dev(scan), dev(prb)))
        b = host(slv.adj_probe(dev(y), dev(scan), dev(p["psi"])))
    for got, want in ((g, op.fwd(p["psi"], scan, prb, ndet, "double")),
                      (a, op.adj(y, scan, prb, p["nz"], p["n"], "double")),
                      (b, op.adj_probe(y, scan, p["psi"], nprb, "double"))):
        assert np.abs(got - want).max() <= REL_MAX * np.abs(want).max()
    lhs = np.vdot(y.astype(np.complex128), g.astype(np.complex128))
    r1 = np.vdot(a.astype(np.complex128), p["psi"].astype(np.complex128))
    r2 = np.vdot(b.astype(np.complex128), prb.astype(np.complex128))
    assert abs(lhs - r1) < 1e-5 * abs(lhs) and abs(lhs - r2) < 1e-5 * abs(lhs)


@pytest.mark.parametrize("ndet,nprb", [(256, 256), (128, 96), (512, 512)])
def test_deterministic_adjoints_are_bitwise_reproducible(pt, ndet, nprb):
    """Option "deterministic": the object / probe adjoints accumulate in 64-bit fixed point with integer
    atomics, so repeated calls give identical bits (the reference's float atomicAdd, kernels.cu:73-80,92-93,
    and the default path here do not), and the result agrees with the default path to float32 rounding."""
    p = syn.make_problem(12, 12, 9, nprb, ndet, seed=4)
    rng = np.random.default_rng(2)
    y = (rng.standard_normal((1, 144, ndet, ndet)) + 1j * rng.standard_normal((1, 144, ndet, ndet))).astype(np.complex64)
    with pt.PtychoCuFFT(144, nprb, ndet, 1, p["nz"], p["n"]) as slv:
        psi, scan, prb, yd = dev(p["psi"]), dev(p["scan"]), dev(p["probe"]), dev(y)
        ref_a, ref_b = host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))
        slv.set_deterministic(True)
        runs = [(host(slv.adj(yd, scan, prb)), host(slv.adj_probe(yd, scan, psi))) for _ in range(3)]
        slv.set_deterministic(False)
    for a, b in runs[1:]:
        np.testing.assert_array_equal(a, runs[0][0])
        np.testing.assert_array_equal(b, runs[0][1])
    assert np.abs(runs[0][0] - ref_a).max() <= 2e-6 * np.abs(ref_a).max()
    assert np.abs(runs[0][1] - ref_b).max() <= 2e-6 * np.abs(ref_b).max()
