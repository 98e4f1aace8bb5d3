"""CG loop on the GPU against the NumPy oracle loop (same seeded inputs)."""
import numpy as np
import pytest

from oracle import cg_oracle as cg
from libtike.hipfft import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    return pt


def setup(nmodes=1, ndet=32, seed=7):
    p = syn.make_problem(6, 6, 6, ndet, ndet, seed=seed)
    probe = syn.hermite_modes(ndet, nmodes) if nmodes > 1 else p["probe"][:, None].copy()
    # A smooth Gaussian probe on a flat start object predicts ~zero amplitude on most of the
    # detector, where the projection f/|f| takes the phase of float32 rounding noise: the CG
    # trajectory is then not reproducible across FFT implementations (the reference's cuFFT
    # included).  A random phase screen spreads the model over the whole detector.
    rng = np.random.default_rng(seed + 100)
    probe = probe * np.exp(2j * np.pi * rng.random(probe.shape[-2:]))
    ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    data = np.zeros((1, p["nscan"], ndet, ndet), np.float32)
    for k in range(probe.shape[1]):
        data += np.abs(ora.fwd(p["psi"], p["scan"], probe[:, k])) ** 2
    return p, probe.astype(np.complex64), ora, data


@pytest.mark.parametrize("nmodes,recover", [(1, False), (1, True), (2, True)])
def test_cg_tracks_the_oracle(pt, nmodes, recover):
    import torch
    p, probe, ora, data = setup(nmodes)
    piter = 5
    start = probe.copy().swapaxes(2, 3) if recover else probe.copy()
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                   piter=piter, recover_prb=recover)
    with pt.CGPtychoSolver(p["nscan"], probe.shape[-1], probe.shape[-1], 1, p["nz"], p["n"]) as slv:
        slv.verbose = False
        slv.log_every = 1
        got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                            piter=piter, recover_prb=recover)
        hist = list(slv.history)
    # same cost trajectory (start-of-iteration cost, float32 reductions)
    for (i, gpsi, gprb, cost), (io, gpsi_o, gprb_o, cost_o) in zip(hist, ora.history):
        assert i == io
        assert abs(cost - cost_o) <= 1e-4 * abs(cost_o), (i, cost, cost_o)
        assert gpsi == gpsi_o and gprb == gprb_o, (i, gpsi, gpsi_o, gprb, gprb_o)
    d = np.abs(got["psi"] - want["psi"]).max() / np.abs(want["psi"]).max()
    assert d < 2e-4, d
    dp = np.abs(got["probe"] - want["probe"]).max() / np.abs(want["probe"]).max()
    assert dp < 2e-4, dp


def test_fused_loop_matches_unfused_with_padded_probe(pt):
    """Fused CG-stage kernels (ptycho_cg_*) against the statement-by-statement torch loop,
    with nprb < ndet so the zero-padded columns of the work buffers are exercised."""
    import torch
    p = syn.make_problem(6, 6, 5, 24, 32, seed=5)
    rng = np.random.default_rng(8)
    probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((24, 24)))).astype(np.complex64)
    ora = cg.OracleSolver(p["nscan"], 24, 32, 1, p["nz"], p["n"])
    data = (np.abs(ora.fwd(p["psi"], p["scan"], probe[:, 0])) ** 2).astype(np.float32)
    out = []
    for fused in (True, False):
        with pt.CGPtychoSolver(p["nscan"], 24, 32, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every, slv.fused = False, 1, fused
            res = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(),
                                probe.copy().swapaxes(2, 3), piter=5, recover_prb=True)
            out.append((res, list(slv.history)))
    (rf, hf), (ru, hu) = out
    for a, b in zip(hf, hu):
        assert a[:3] == b[:3] and abs(a[3] - b[3]) <= 1e-4 * abs(b[3]), (a, b)
    assert np.abs(rf["psi"] - ru["psi"]).max() < 2e-4
    assert np.abs(rf["probe"] - ru["probe"]).max() < 2e-4 * np.abs(ru["probe"]).max()
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(),
                   probe.copy().swapaxes(2, 3), piter=5, recover_prb=True)
    assert np.abs(rf["psi"] - want["psi"]).max() < 2e-4


def test_fused_multimode_matches_unfused(pt):
    """Three probe modes (the /root/reference/tests/test_modes.py scenario shape), fused
    multi-mode kernels against the statement-by-statement torch loop and the oracle."""
    p, probe, ora, data = setup(3)
    start = probe.copy().swapaxes(2, 3)
    out = []
    for fused in (True, False):
        with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every, slv.fused = False, 1, fused
            res = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                                piter=4, recover_prb=True)
            out.append((res, list(slv.history)))
    (rf, hf), (ru, hu) = out
    for a, b in zip(hf, hu):
        assert a[:3] == b[:3] and abs(a[3] - b[3]) <= 1e-4 * abs(b[3]), (a, b)
    assert np.abs(rf["psi"] - ru["psi"]).max() < 2e-4
    assert np.abs(rf["probe"] - ru["probe"]).max() < 2e-4 * np.abs(ru["probe"]).max()
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(), piter=4, recover_prb=True)
    assert np.abs(rf["psi"] - want["psi"]).max() < 2e-4


def test_two_modes_on_a_1024_detector(pt):
    """ndet = 1024 has no LDS window (the compact multi-mode layout needs one): two probe modes on 4 positions must run
    -- through the statement-by-statement loop on the un-windowed operators -- and track the oracle."""
    ndet = 1024
    p = syn.make_problem(2, 2, 8, ndet, ndet, seed=9)
    rng = np.random.default_rng(10)
    probe = (syn.hermite_modes(ndet, 2) * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64)
    ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    data = sum(np.abs(ora.fwd(p["psi"], p["scan"], probe[:, k])) ** 2 for k in range(2)).astype(np.float32)
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy(), piter=2)
    with pt.CGPtychoSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"]) as slv:
        slv.verbose, slv.log_every = False, 1
        got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy(), piter=2)
        hist = list(slv.history)
    for a, b in zip(hist, ora.history):
        assert a[1] == b[1] and abs(a[3] - b[3]) <= 1e-4 * abs(b[3]), (a, b)
    assert np.abs(got["psi"] - want["psi"]).max() < 2e-4 * np.abs(want["psi"]).max()


def test_two_angles_per_call_fused_and_unfused(pt):
    """ptheta = 2: two angular views solved in one call (shared scalars a, b and one joint
    line search, as in the reference's run); fused kernels vs the torch loop vs the oracle."""
    p, probe, ora, data = setup()
    rng = np.random.default_rng(4)
    scan2 = np.concatenate([p["scan"], p["scan"] + rng.random(p["scan"].shape).astype(np.float32) * 0.4])
    probe2 = np.concatenate([probe, probe * np.exp(0.3j)]).astype(np.complex64)
    psi_true = np.concatenate([p["psi"], np.conj(p["psi"])])
    ora2 = cg.OracleSolver(p["nscan"], 32, 32, 2, p["nz"], p["n"])
    data2 = (np.abs(ora2.fwd(psi_true, scan2, probe2[:, 0])) ** 2).astype(np.float32)
    want = ora2.run(data2.copy(), np.ones_like(psi_true), scan2.copy(), probe2.copy(), piter=4, recover_prb=True)
    for fused in (True, False):
        with pt.CGPtychoSolver(p["nscan"], 32, 32, 2, p["nz"], p["n"]) as slv:
            slv.verbose, slv.fused = False, fused
            got = slv.run_batch(data2.copy(), np.ones_like(psi_true), scan2.copy(), probe2.copy(),
                                piter=4, recover_prb=True)
        assert np.abs(got["psi"] - want["psi"]).max() < 2e-4, fused
        assert np.abs(got["probe"] - want["probe"]).max() < 2e-4 * np.abs(want["probe"]).max(), fused


def test_run_batch_streams_angle_partitions(pt):
    """run_batch over several independent angles (prefetch on a copy stream) equals solving
    each angle on its own; angle_shard splits the partitions over ranks without a collective."""
    p, probe, ora, data = setup()
    nang = 3
    rng = np.random.default_rng(1)
    psis = np.concatenate([np.ones_like(p["psi"])] * nang)
    scans = np.concatenate([p["scan"] + rng.random(p["scan"].shape).astype(np.float32) * 0.3 for _ in range(nang)])
    probes = np.concatenate([probe * (1 + 0.1 * a) for a in range(nang)])
    datas = np.concatenate([(np.abs(ora.fwd(p["psi"], scans[a:a + 1], probes[a:a + 1, 0])) ** 2).astype(np.float32)
                            for a in range(nang)])
    with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        slv.verbose = False
        full = slv.run_batch(datas, psis, scans, probes, piter=3)
        single = [slv.run_batch(datas[a:a + 1], psis[a:a + 1], scans[a:a + 1], probes[a:a + 1], piter=3)
                  for a in range(nang)]
        shard = slv.run_batch(datas, psis, scans, probes, piter=3, angle_shard=(1, 2))
    for a in range(nang):
        np.testing.assert_allclose(full["psi"][a], single[a]["psi"][0], atol=2e-5)
        np.testing.assert_allclose(full["probe"][a], single[a]["probe"][0], atol=2e-5)
    np.testing.assert_allclose(shard["psi"][1], full["psi"][1], atol=2e-5)       # rank 1 of 2 owns angle 1
    np.testing.assert_array_equal(shard["psi"][0], psis[0])                      # others untouched
    np.testing.assert_array_equal(shard["psi"][2], psis[2])


def test_cg_gradient_vanishes_at_truth(pt):
    p, probe, ora, data = setup()
    with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        slv.verbose = False
        got = slv.run_batch(data, p["psi"].copy(), p["scan"].copy(), probe.copy(), piter=1)
    assert np.abs(got["psi"] - p["psi"]).max() < 1e-4


def test_registration_recovers_known_shift(pt):
    import torch
    from libtike.hipfft.ptycho import register_translation_batch
    rng = np.random.default_rng(0)
    img = rng.standard_normal((3, 32, 32))
    f = np.fft.fft2(img)
    ky = np.fft.fftfreq(32)[:, None]
    kx = np.fft.fftfreq(32)[None, :]
    true = np.array([[1.25, -2.5], [0.0, 0.37], [-3.0, 4.11]])
    moved = np.stack([f[i] * np.exp(-2j * np.pi * (ky * true[i, 0] + kx * true[i, 1])) for i in range(3)])
    want = cg.register_translation_batch(f.astype(np.complex64), moved.astype(np.complex64), 100, "fourier")
    with pt.PtychoCuFFT(3, 32, 32, 1, 64, 64) as slv:
        a, b = (torch.as_tensor(z.astype(np.complex64), device="cuda") for z in (f, moved))
        got = register_translation_batch(a, b, 100, "fourier", op=slv).cpu().numpy()
        # the reference's positional signature (ptycho.py:190), without an operator: a temporary handle
        np.testing.assert_array_equal(register_translation_batch(a, b, 100, "fourier").cpu().numpy(), got)
    np.testing.assert_allclose(got, -true, atol=0.011)
    np.testing.assert_allclose(got, want, atol=0.011)


@pytest.mark.gpu
@pytest.mark.parametrize("ndet", [16, 32, 64, 128, 256, 512, 1024])
def test_zoom_kernel_matches_torch_contraction(ndet):
    """Fused sub-pixel registration stage (``ptycho_cg_zoom``: peak wrap, phases, real
    low-rank zoomed DFT on the float64 matrix cores, arg-max) against the torch GEMM
    restatement of ``ptycho.py:163-188,209-235`` and against the oracle's einsum."""
    import torch
    from libtike.hipfft import ptycho as P
    from oracle import cg_oracle as co
    nscan = 37 if ndet <= 256 else 9      # ndet % 64 != 0: scalar-operand kernel; else MFMA <256>, <512>, <1024>
    rng = np.random.default_rng(5)
    with P.CGPtychoSolver(nscan, ndet, ndet, 1, ndet + 8, ndet + 8) as slv:
        # smooth correlation peak near a random sub-pixel shift + noise
        ky = np.fft.fftfreq(ndet)[None, :, None]
        kx = np.fft.fftfreq(ndet)[None, None, :]
        true = rng.uniform(-3, 3, (nscan, 2))
        base = rng.standard_normal((nscan, ndet, ndet)) ** 2 + 0.1
        ip = base * np.exp(-2j * np.pi * (ky * true[:, 0, None, None] + kx * true[:, 1, None, None]))
        ip = (ip + 0.05 * (rng.standard_normal(ip.shape) + 1j * rng.standard_normal(ip.shape))).astype(np.complex64)
        coarse = np.round(true)                        # whole-pixel stage result
        maxima = np.where(coarse < 0, coarse + ndet, coarse).astype(np.int64)   # unwrapped peak indices
        off = 75.0 - coarse * 100
        ref = np.conj(co.upsampled_dft_batch(np.conj(ip), 150, 100, off))
        peak_o = np.stack(np.unravel_index(np.abs(ref).reshape(nscan, -1).argmax(1), (150, 150)), axis=1)
        want = coarse + (peak_o - 75.0) / 100
        dip = torch.as_tensor(ip, device="cuda")
        dmax = torch.as_tensor(maxima, device="cuda")
        packed = (0xffffffff - (dmax[:, 0] * ndet + dmax[:, 1])).contiguous()
        got = P._zoom_shifts_native(slv, dip, packed, 100)
        assert got is not None, "native zoom kernels declined a case they should cover"
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        # the torch GEMM fallback (torch divides by a scalar through its reciprocal: 1 ulp) and
        # the public wrapper agree as well
        np.testing.assert_allclose(P._finish_registration(dip, dmax, 100).cpu().numpy(), want, rtol=0, atol=1e-13)
        np.testing.assert_array_equal(P._finish_registration(dip, dmax, 100, op=slv).cpu().numpy(), want)
        assert np.abs(got.cpu().numpy() - true).max() < 0.02


@pytest.mark.parametrize("ndet", [112, 96, 192, 100])
def test_cg_at_a_cropped_detector_size(pt, ndet):
    """ndet = nprb = 112 (tests/test_fsc.py:115-120), 96 and 192: the device-resident fused loop on the mixed-radix plans
    (7 x 4 x 4, 3 x 4 x 8, 3 x 8 x 8); 100: the statement-by-statement loop on the Bluestein operators.  All track the oracle."""
    p = syn.make_problem(4, 4, 9, ndet, ndet, seed=21)
    rng = np.random.default_rng(9)
    probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64)
    ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    data = (np.abs(ora.fwd(p["psi"], p["scan"], probe[:, 0])) ** 2).astype(np.float32)
    want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy(), piter=4)
    with pt.CGPtychoSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"]) as slv:
        slv.verbose, slv.log_every = False, 1
        got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy(), piter=4)
        hist = list(slv.history)
    for (i, gp, gq, c), (io, gpo, gqo, co) in zip(hist, ora.history):
        assert (i, gp, gq) == (io, gpo, gqo) and abs(c - co) <= 1e-4 * abs(co)
    assert np.abs(got["psi"] - want["psi"]).max() < 2e-4 * np.abs(want["psi"]).max()


def test_line_search_decision_kernel_replays_the_reference_rule(pt):
    """``k_cg_ls_decide`` (C ABI ``ptycho_cg_ls_next(pass = 4)``: decide only) on hand-made cost tables, against a
    transcription of ``line_search_sqr`` (ptycho.py:253-281): first step length whose float32 cost is not above
    f(p1) wins, 0.5 * step lands in the gamma word, the accepted index becomes the hint; below 1e-32 the search fails
    (gamma 0, failure counter + 1).  Several groups of 16 per pass, continuation over passes (gamma0 / tried)."""
    import torch
    from libtike.hipfft import _native as nat
    from libtike.hipfft.ptycho import _ptr, _stream
    LS_GAMMA0, LS_NCAND, LS_NGROUPS, LS_TRIED, LS_RESOLVED = 14, 15, 16, 17, 18     # enum PTYCHO_ST_* (ptycho_hip.h)
    rng = np.random.default_rng(12)
    with pt.CGPtychoSolver(4, 16, 16, 1, 48, 48) as slv:
        dummy = torch.zeros(4, device="cuda")
        for case in range(40):
            ngroups = int(rng.integers(1, 8))
            ncand = 16 if ngroups > 1 else int(rng.choice([4, 8, 12, 16]))
            tried0 = int(rng.choice([0, 4, 16, 48]))
            gamma0 = 0.5 ** tried0
            fp1 = float(rng.uniform(1.0, 2.0)) * 1e6
            table = np.zeros((7, 17))
            accept_at = int(rng.integers(0, ngroups * ncand + 6))      # beyond the pass: unresolved
            for g in range(ngroups):
                for j in range(ncand):
                    k = g * ncand + j
                    table[g, j] = fp1 * (1.0 + 1e-3) if k < accept_at else fp1 * (1.0 - 1e-3 * rng.random())
                    if k == accept_at and case % 3 == 0:
                        table[g, j] = np.float64(np.float32(fp1)) * (1.0 + 1e-9)   # equal in float32: "not above" accepts
                table[g, ncand] = fp1
            st = torch.zeros(nat.ST_WORDS, dtype=torch.float64)
            st[nat.ST_HINT:nat.ST_HINT + 2] = 14.0
            st[LS_GAMMA0], st[LS_NCAND], st[LS_NGROUPS], st[LS_TRIED] = gamma0, ncand, ngroups, tried0
            st[nat.ST_COSTS:nat.ST_COSTS + 7 * 17] = torch.as_tensor(table.ravel())
            st = st.cuda()
            nat.check(nat.cg_ls_next(slv._h, _ptr(st), 0, 4, _ptr(dummy), 0, _stream()))
            got = st.cpu().numpy()
            # transcription of the rule over this pass's step lengths
            step, want_gamma, want_hint, failed, resolved = gamma0, None, 14.0, 0, False
            for k in range(ngroups * ncand):
                g, j = divmod(k, ncand)
                if not (np.float32(table[g, j]) > np.float32(table[g, ncand])):
                    want_gamma, want_hint, resolved = 0.5 * step, float(tried0 + k), True
                    break
                if step < 1e-32:
                    want_gamma, failed, resolved = 0.0, 1, True
                    break
                step *= 0.5
            if not resolved:            # pass 4 is the last one: the kernel's fail-safe closes the search
                want_gamma, failed = 0.0, 1
            assert got[nat.ST_GAMMA_PSI] == want_gamma, (case, got[nat.ST_GAMMA_PSI], want_gamma)
            assert got[nat.ST_HINT] == want_hint and got[nat.ST_LS_FAILED] == failed and got[LS_RESOLVED] == 1.0, case


def test_cg_runs_are_bitwise_reproducible(pt):
    """The fused loops use the deterministic adjoints (``solver.reproducible``, default on): two runs of the same
    problem give the same bits -- object, probe, positions and logged costs; with the reference's float ``atomicAdd``
    (kernels.cu:73-80) they do not.  One mode (native stage loop) and three modes (compact slot layout)."""
    ndet = 64
    p = syn.make_problem(6, 6, 8, ndet, ndet, seed=5)
    rng = np.random.default_rng(6)
    for M in (1, 3):
        probe = np.stack([p["probe"] * np.exp(2j * np.pi * rng.random((ndet, ndet))) / (k + 1) for k in range(M)], axis=1).astype(np.complex64)
        ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
        data = sum(np.abs(ora.fwd(p["psi"], p["scan"], probe[:, k])) ** 2 for k in range(M)).astype(np.float32)
        runs = []
        for rep in range(2):
            with pt.CGPtychoSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"]) as slv:
                slv.verbose, slv.log_every = False, 1
                assert slv.reproducible
                import torch
                scan = torch.as_tensor(p["scan"].copy(), device="cuda")
                got = slv.run(torch.as_tensor(data, device="cuda"), torch.ones((1, p["nz"], p["n"]), dtype=torch.complex64, device="cuda"),
                              scan, torch.as_tensor(probe.copy(), device="cuda"), piter=5, recover_prb=True)
                runs.append((got["psi"].cpu().numpy(), got["probe"].cpu().numpy(), scan.cpu().numpy(), list(slv.history)))
        a, b = runs
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
        assert a[3] == b[3]


def test_fused_registration_right_on_its_first_launch():
    """NOTEBOOK.md (round 1) recorded an occupancy-capped variant of the registration row pass that was wrong on its
    first launch in a process and right afterwards.  That variant is gone; this pins the shipped path: in a FRESH
    process, the very first solver calls are the fused position-correction kernels (two ones-probe column passes,
    CROSS, arg-max, zoom) on never-used work slots, checked against the un-fused registration (HIP operators +
    register_translation_batch through torch)."""
    import subprocess
    import sys
    import os
    code = r'''
import sys, os
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "libtike-cufft_amd"))
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
from libtike.hipfft.ptycho import register_translation_batch
for ndet in (256, 64):
    p = syn.make_problem(5, 6, 9, ndet, ndet, seed=3)
    rng = np.random.default_rng(1)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi, scan = dev(p["psi"]), dev(p["scan"])
    dpsi = dev((rng.standard_normal(p["psi"].shape) + 1j * rng.standard_normal(p["psi"].shape)).astype(np.complex64) * 0.05)
    probe = dev(p["probe"][:, None])
    slv = pt.CGPtychoSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    got = slv._position_shifts(psi, dpsi, 0.25, scan, probe).cpu().numpy()     # first launches of this handle
    ones = torch.ones_like(probe[:, 0])
    t1 = slv.fwd(psi, scan, ones)[0]
    t2 = slv.fwd(psi + 0.25 * dpsi, scan, ones)[0]
    want = register_translation_batch(t1, t2, 100, "fourier").cpu().numpy()
    again = slv._position_shifts(psi, dpsi, 0.25, scan, probe).cpu().numpy()
    assert np.array_equal(got, again), (ndet, np.abs(got - again).max())
    assert np.abs(got - want).max() <= 0.0100001, (ndet, np.abs(got - want).max())   # one step of the 1/100 px grid at most
    assert (np.abs(got - want) > 1e-9).mean() < 0.1, ndet
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_work_slots_follow_the_shared_gathers(pt):
    """The native loop with the position correction's shared patch gathers holds FOUR farplane-sized work slots (0-3);
    without sharing two, and the two extra ones are given back (the footprint doubled silently).  The result
    does not depend on it, and a released slot is allocated again on demand."""
    import torch
    p, probe, ora, data = setup(1)
    D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    res = {}
    with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        slv.verbose = False
        for share in (True, False, True):
            slv.share_ones = share
            out = slv.run(D(data), torch.ones_like(D(p["psi"])), D(p["scan"]).clone(), D(probe).clone(), piter=4)
            slots = slv.work_slots_allocated()
            assert slots == ([0, 1, 2, 3] if share else [0, 1]), (share, slots)
            res.setdefault(share, []).append(out["psi"].cpu().numpy())
        slv.release_work(3)
        assert slv.work_slots_allocated() == [0, 1, 2]
        again = slv.run(D(data), torch.ones_like(D(p["psi"])), D(p["scan"]).clone(), D(probe).clone(), piter=4)["psi"].cpu().numpy()
    assert np.array_equal(res[True][0], res[True][1]) and np.array_equal(res[True][0], again)
    assert np.abs(res[True][0] - res[False][0]).max() <= 1e-5 * np.abs(res[True][0]).max()


def test_profile_read_accepts_the_16_entry_arrays_of_earlier_headers(pt):
    import ctypes
    import torch
    from libtike.hipfft import _native as nat
    p, probe, ora, data = setup(1)
    D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    with pt.PtychoCuFFT(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        slv.profile(True)
        slv.fwd(D(p["psi"]), D(p["scan"]), D(probe[:, 0]))
        ms, cnt = (ctypes.c_double * 16)(), (ctypes.c_longlong * 16)()
        nat.check(nat.profile_read(slv._h, ms, cnt, 16))          # ids 16 / 17 (the tile kernels of ndet <= 128) are dropped
        assert sum(cnt) == 0 or all(c >= 0 for c in cnt)
        with pytest.raises(nat.PtychoHipError):
            nat.check(nat.profile_read(slv._h, ms, cnt, 15))


def test_pending_deferred_gradient_is_guarded(pt):
    """Option defer_finish leaves the object gradient in the adjoint's fixed-point image until ptycho_cg_obj_dir folds it in.
    A deterministic adjoint or a projection issued in that window used to corrupt both results silently; they
    now fail with PTYCHO_ERR_ARG, and work again once the gradient has been folded in."""
    import torch
    from libtike.hipfft import _native as nat
    from libtike.hipfft.ptycho import _ptr, _stream
    p, probe, ora, data = setup(1)
    D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        h = slv._h
        nat.check(nat.set_option(h, b"deterministic", 1))
        nat.check(nat.set_option(h, b"defer_finish", 1))
        st = torch.zeros(nat.ST_WORDS, dtype=torch.float64, device="cuda")
        st[nat.ST_HINT:nat.ST_HINT + 2] = 14.0
        psi, scan, prb, dat = torch.ones_like(D(p["psi"])), D(p["scan"]), D(probe[:, 0]).contiguous(), D(data)
        grad, grad0, dpsi = torch.empty_like(psi), torch.zeros_like(psi), torch.zeros_like(psi)
        S = _stream()
        nat.check(nat.cg_obj_begin(h, _ptr(st), _ptr(psi), _ptr(scan), _ptr(prb), _ptr(dat), S))
        nat.check(nat.cg_obj_grad(h, _ptr(st), _ptr(scan), _ptr(prb), _ptr(dat), _ptr(grad), S))
        g = torch.zeros((1, p["nscan"], 32, 32), dtype=torch.complex64, device="cuda")
        out, cost = torch.zeros_like(psi), torch.zeros(1, dtype=torch.float64, device="cuda")
        assert nat.adj(h, _ptr(out), _ptr(g), _ptr(scan), _ptr(prb), 0, S) != 0 and b"pending" in nat.last_error()
        assert nat.cg_project(h, 0, 1, _ptr(dat), None, _ptr(cost), S) != 0 and b"pending" in nat.last_error()
        nat.check(nat.cg_obj_dir(h, _ptr(st), 1, _ptr(scan), _ptr(prb), _ptr(dat), _ptr(grad), _ptr(grad0), _ptr(dpsi), S))
        nat.check(nat.adj(h, _ptr(out), _ptr(g), _ptr(scan), _ptr(prb), 0, S))      # folded in: allowed again
        torch.cuda.synchronize()
        assert torch.isfinite(torch.view_as_real(dpsi)).all() and float(dpsi.abs().max()) > 0
        nat.check(nat.set_option(h, b"defer_finish", 0))
        nat.check(nat.set_option(h, b"deterministic", 0))
