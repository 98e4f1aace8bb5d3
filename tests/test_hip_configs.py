"""Parity on the kernel instantiations and sizes that ``bench.py`` times.

BASELINE.json configs, and what checks each of them here:

* configs[1]  4096 x 256^2, 1 mode, 50 CG iterations
    - ``test_cg256_tracks_the_oracle``: the fused N = 256 kernels (16 x 16 plan, unsplit
      column passes, ``k_rows_fused<256, EP>``, ``k_cols_adjwin<256>``, ``k_cols_argmax<256>``,
      ``k_zoom_mfma<256>``) against ``oracle.cg_oracle`` on the configs[1] geometry cut to
      8 x 8 positions: identical step sizes, cost within 1e-4, psi / probe within 2e-4;
    - ``test_cg_full_size_properties``: the full 4096 x 256^2 problem: 50 iterations with a
      non-increasing logged cost, fused == statement-by-statement loop over the first 3
      iterations, zero gradient at the truth.
* configs[2]  4096 x 512^2, 4 probe modes
    - ``test_cg512_four_modes_tracks_the_oracle``: 8 x 8 x 8 plan, ``STATS_M`` /
      ``LINESEARCH_M`` with the four Hermite modes of SURVEY.md 8(d) on 16 positions;
    - ``test_cfg3_full_size_operator_properties``: 4096 x 512^2 adjoint identity < 1e-5.
    - ``test_cfg3_full_size_cg_properties``: the multi-mode CG loop at full size (cost descends,
      fused == statement-by-statement).
* configs[3]  262144 positions over 8 GPUs = 32768 positions x 256^2 per GPU
    - ``test_cfg4_shard_adjoint_identity_and_cg``: one rank's shard (16 GiB farplane).
* configs[4]  180 angles streamed
    - ``test_cfg5_angle_stream_full_size_equals_per_angle_solves``: 4 angles of 4096 x 256^2 through
      ``run_batch`` with two ``angle_shard``s == per-angle solves, bit for bit;
      ``tests/test_hip_cg.py::test_run_batch_streams_angle_partitions`` covers the angle loop's edge cases.
* bench.py's own CG problem (smooth probe): ``test_bench_problem_tracks_the_oracle_while_it_can`` (8 x 8
  positions vs the oracle), ``test_bench_problem_full_size_is_reproducible_and_descends``.
"""
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


from cg_cases import phase_screen   # noqa: E402  (shared with the CPU oracle tests)
import cg_cases as cc                # noqa: E402


def check_history(hist, ohist, rtol=1e-4, free_prb_from=None):
    """Identical step sizes, cost within ``rtol``.  ``free_prb_from``: from that iteration on the probe
    step is only required to be 'deep' (< 2^-20, or a failed search) on both sides -- see
    tests/test_oracle_divergence.py for why no two implementations can share it."""
    assert len(hist) == len(ohist) and len(hist) > 0
    for (i, gpsi, gprb, cost), (io, gpsi_o, gprb_o, cost_o) in zip(hist, ohist):
        assert i == io
        assert gpsi == gpsi_o, (i, gpsi, gpsi_o)
        if free_prb_from is not None and i >= free_prb_from:
            assert gprb < 2.0 ** -20 and gprb_o < 2.0 ** -20, (i, gprb, gprb_o)
        else:
            assert gprb == gprb_o, (i, gprb, gprb_o)
        assert abs(cost - cost_o) <= rtol * abs(cost_o), (i, cost, cost_o)


@pytest.mark.parametrize("nprb,recover", [(256, False), (256, True), (128, False), (128, True)])
def test_cg256_tracks_the_oracle(pt, nprb, recover):
    """configs[1] geometry (raster step 8 px + jitter, Gaussian probe) cut to 8 x 8 positions.

    nprb = 128 with probe recovery (the padded variant of SURVEY.md 8d): the probe line search of
    iteration 2 backtracks to ~2^-30 -- the complex64 oracle accepts 2^-29, the complex128 oracle fails
    the search (``tests/test_oracle_divergence.py`` runs both).  Everything before that search must equal
    the oracle; from it on the probe step only has to be that deep too (the probe then moves by < 1e-9 of
    itself either way, so object, probe and the later object steps still have to agree)."""
    ndet, piter = 256, 4
    p, data, start = cc.cfg256_case(nprb, recover)
    ora = cg.OracleSolver(p["nscan"], nprb, ndet, 1, p["nz"], p["n"])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                       piter=piter, recover_prb=recover)
    split = cc.CFG256_NPRB128_SPLIT[0] if (nprb == 128 and recover) else None
    with pt.CGPtychoSolver(p["nscan"], nprb, ndet, 1, p["nz"], p["n"]) as slv:
        slv.verbose, slv.log_every = False, 1
        assert slv.fused
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                                piter=piter, recover_prb=recover)
        hist = list(slv.history)
    # Cost tolerance.  Step sizes and position shifts must be identical.  The cost is a sum of
    # squared differences of nearly equal numbers, and with probe recovery the probe feeds every
    # pattern, so the float32 rounding pattern of the FFT implementation is amplified: measured
    # here (tools/dbg_cfg256.py, r02) oracle complex64 vs complex128 differ by 3e-5, the fused and
    # the statement-by-statement GPU loops agree to 2e-6 with each other and sit 2.8e-4 from the
    # oracle at iteration 2.  Object and probe stay within 2e-4 of the oracle.
    check_history(hist, ora.history, rtol=5e-4 if recover else 1e-4, free_prb_from=split)
    assert np.abs(got["psi"] - want["psi"]).max() < 2e-4 * np.abs(want["psi"]).max()
    assert np.abs(got["probe"] - want["probe"]).max() < 2e-4 * np.abs(want["probe"]).max()
    if recover:   # fused kernels == HIP operators + torch elementwise loop, far below that amplification
        with pt.CGPtychoSolver(p["nscan"], nprb, ndet, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every, slv.fused = False, 1, False
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                                    piter=piter, recover_prb=recover)
            check_history(hist, list(slv.history), rtol=2e-5, free_prb_from=split)
        assert np.abs(got["psi"] - ref["psi"]).max() < 2e-5 * np.abs(ref["psi"]).max()
        assert np.abs(got["probe"] - ref["probe"]).max() < 2e-5 * np.abs(ref["probe"]).max()


def test_bench_problem_tracks_the_oracle_while_it_can(pt):
    """bench.py's own CG problem (smooth Gaussian probe, flat start, no probe recovery) cut to 8 x 8
    positions.  99 % of this problem's gradient is the phase of float32 rounding noise on detector
    pixels the model leaves dark (``tests/test_oracle_divergence.py``: five FFT orderings of the oracle
    itself give three trajectories), and the fused loop applies the probe rescale a/b after the
    transform instead of before it (linear, but a different rounding pattern -- measured in round 3: the
    statement-by-statement GPU loop accepts the oracle's first step 2^-2, the fused loop 2^-1).  So no
    trajectory is shared; what is noise free must match the oracle: the cost at the start of iteration
    0 (a, b and the rescaled intensity enter it) to 1e-5, for the fused and the statement-by-statement
    loop.  Beyond that: the logged cost never increases and no line search fails over 12 iterations.
    The loop arithmetic itself is oracle-checked on phase-screened probes (the tests above)."""
    import warnings
    p, data, probe = cc.bench_case(8)
    hs, _ = cc.oracle_history(p, data, probe, 1, False, "single")
    for fused in (True, False):
        with pt.CGPtychoSolver(p["nscan"], 256, 256, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every, slv.fused = False, 1, fused
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), probe.copy(),
                              piter=12 if fused else 2)
            hist = list(slv.history)
        assert abs(hist[0][3] - hs[0][3]) <= 1e-5 * hs[0][3], (fused, hist[0], hs[0])
        costs = np.array([h[3] for h in hist])
        assert np.all(costs[1:] <= costs[:-1] * (1 + 1e-6)), (fused, costs)
        assert all(h[1] > 0 for h in hist), (fused, hist)


def test_cg512_four_modes_tracks_the_oracle(pt):
    """configs[2]: ndet = nprb = 512, the four Gaussian x Hermite modes, 4 x 4 positions."""
    ndet, piter, M = 512, 3, 4
    p = syn.make_problem(4, 4, 8, ndet, ndet, seed=12)
    probe = phase_screen(syn.hermite_modes(ndet, M), 112)
    ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    data = np.zeros((1, p["nscan"], ndet, ndet), np.float32)
    for k in range(M):
        data += np.abs(ora.fwd(p["psi"], p["scan"], probe[:, k])) ** 2
    for recover in (False, True):
        ora = cg.OracleSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
        start = probe.copy().swapaxes(2, 3) if recover else probe.copy()
        want = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                       piter=piter, recover_prb=recover)
        with pt.CGPtychoSolver(p["nscan"], ndet, ndet, 1, p["nz"], p["n"]) as slv:
            slv.verbose, slv.log_every = False, 1
            got = slv.run_batch(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                                piter=piter, recover_prb=recover)
            hist = list(slv.history)
        check_history(hist, ora.history, rtol=5e-4 if recover else 1e-4)
        assert np.abs(got["psi"] - want["psi"]).max() < 2e-4 * np.abs(want["psi"]).max(), recover
        assert np.abs(got["probe"] - want["probe"]).max() < 2e-4 * np.abs(want["probe"]).max(), recover


def _cfg2_device_problem(torch, seed=1234):
    """configs[1] at full size, on the device: 64 x 64 raster, 768^2 object, phase-screened probe."""
    p = syn.make_problem(64, 64, 8, 256, 256, seed=seed, nz=768, n=768)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    probe = phase_screen(p["probe"][:, None], seed + 1)
    return p, dev(p["psi"]), dev(p["scan"]), dev(probe)


def test_cg_full_size_properties(pt):
    """configs[1], full size (4096 x 256^2, 50 iterations): properties that need no oracle run.
    (1) the logged cost (start-of-iteration cost, ptycho.py:475-482) never increases;
    (2) the fused loop equals the statement-by-statement loop (HIP operators + torch
        elementwise, i.e. the reference's expressions) over the first 3 iterations:
        identical step sizes, cost within 1e-4, psi within 2e-4;
    (3) started at the true object the gradient vanishes: psi does not move;
    (4) the one-GPU and the multi-GPU line-search schedules accept the same step lengths."""
    import torch
    p, psi_true, scan, probe = _cfg2_device_problem(torch)
    with pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768) as slv:
        slv.verbose, slv.log_every = False, 1
        data = (torch.abs(slv.fwd(psi_true, scan, probe[:, 0])) ** 2).contiguous()
        out = slv.run(data, torch.ones_like(psi_true), scan.clone(), probe.clone(), piter=50)
        hist = list(slv.history)
        assert len(hist) == 50
        costs = np.array([h[3] for h in hist])
        assert np.all(np.isfinite(costs))
        assert np.all(costs[1:] <= costs[:-1] * (1 + 1e-6)), costs
        assert costs[-1] < 0.05 * costs[0], costs
        assert all(h[1] > 0 for h in hist), "a line search failed"
        assert bool(torch.isfinite(torch.view_as_real(out["psi"])).all())

        res = []
        for fused in (True, False):
            slv.fused = fused
            slv.history = []
            r = slv.run(data, torch.ones_like(psi_true), scan.clone(), probe.clone(), piter=3)
            res.append((r["psi"].clone(), list(slv.history)))
            del r
            torch.cuda.empty_cache()
        (pf, hf), (pu, hu) = res
        check_history(hf, hu)
        d = float(torch.abs(pf - pu).max() / torch.abs(pu).max())
        assert d < 2e-4, d

        slv.fused = True
        got = slv.run(data, psi_true.clone(), scan.clone(), probe.clone(), piter=1)
        assert float(torch.abs(got["psi"] - psi_true).max()) < 1e-4

        # (4) the line-search schedules of the native loop (passes of <=16, 16, 32, 64 step lengths on one GPU;
        # <=16, 32, 80 with a process group, one collective per pass; <=16, 112 on request) accept the same steps.
        # From the flat start the first searches of this problem go far beyond the first 16 step lengths, where the
        # schedules differ.
        smooth = torch.as_tensor(np.ascontiguousarray(p["probe"][:, None]), device="cuda")   # bench.py's probe
        data = (torch.abs(slv.fwd(psi_true, scan, smooth[:, 0])) ** 2).contiguous()
        # deep searches decide between float32 costs that differ in the last digits, so the float-atomic adjoint's
        # run-to-run rounding noise can move an accepted index; the fixed-point adjoints take that out
        slv.set_deterministic(True)
        res = []
        for two in (False, True, "all"):
            slv.ls_two_pass = two
            slv.history = []
            r = slv.run(data, torch.ones_like(psi_true), scan.clone(), smooth.clone(), piter=4, recover_prb=True)
            res.append((r["psi"].clone(), list(slv.history)))
            del r
        (pa, ha), (pb, hb), (pc, hc) = res
        assert min(h[1] for h in ha) < 2.0 ** -18, ha
        check_history(ha, hb)
        check_history(ha, hc)
        assert float(torch.abs(pa - pb).max() / torch.abs(pb).max()) < 2e-4
        assert float(torch.abs(pa - pc).max() / torch.abs(pc).max()) < 2e-4
        slv.ls_two_pass = None


def _dot(a, b, step=1024):
    import torch
    s = 0
    for i in range(0, a.shape[1], step):
        s = s + torch.sum(a[:, i:i + step].to(torch.complex128) * b[:, i:i + step].conj().to(torch.complex128))
    return complex(s)


def test_cfg3_full_size_operator_properties(pt):
    """configs[2] operators at full size: 4096 x 512^2 (8 GiB farplane), nprb = 512: adjoint
    identity with an independent y for both adjoints < 1e-5, linearity in the probe over the
    four Hermite modes."""
    import torch
    p = syn.make_problem(64, 64, 8, 512, 512, seed=21, nz=1024, n=1024)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi, scan = dev(p["psi"]), dev(p["scan"])
    modes = dev(syn.hermite_modes(512, 4))
    gen = torch.Generator(device="cuda").manual_seed(5)
    with pt.PtychoCuFFT(4096, 512, 512, 1, 1024, 1024) as slv:
        y = torch.view_as_complex(torch.randn((1, 4096, 512, 512, 2), generator=gen, device="cuda"))
        ax = slv.fwd(psi, scan, modes[:, 0])
        lhs = _dot(ax, y)
        aty = slv.adj(y, scan, modes[:, 0])
        r1 = complex(torch.sum(psi.to(torch.complex128) * aty.conj().to(torch.complex128)))
        bty = slv.adj_probe(y, scan, psi)
        r2 = complex(torch.sum(modes[:, 0].to(torch.complex128) * bty.conj().to(torch.complex128)))
        assert abs(lhs - r1) / abs(lhs) < 1e-5 and abs(lhs - r2) / abs(lhs) < 1e-5, (lhs, r1, r2)
        del y, aty, bty
        # fwd is linear in the probe: fwd(psi, m0 + i m3) = fwd(psi, m0) + i fwd(psi, m3)
        mix = (modes[:, 0] + 1j * modes[:, 3]).contiguous()
        gm = slv.fwd(psi, scan, mix)
        gm -= ax
        del ax
        g3 = slv.fwd(psi, scan, modes[:, 3])
        gm -= 1j * g3
        assert float(torch.abs(gm).max() / torch.abs(g3).max()) < 2e-5


def test_cfg4_shard_adjoint_identity_and_cg(pt):
    """configs[3]: one GPU's share of the 262144-position job = 32768 positions x 256^2 (a
    64 x 512 band of the 512 x 512 raster, 16 GiB farplane): adjoint identity < 1e-5 for both
    adjoints and a short CG run with a decreasing cost."""
    import torch
    R1, R2 = 64, 512
    nz, n = syn.object_size_for(R1, R2, 8, 256)
    rng = np.random.default_rng(1)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi = dev(syn.random_object(nz, n, rng))
    scan = dev(syn.raster_scan(R1, R2, 8, rng))
    prb = dev(phase_screen(syn.gaussian_probe(256), 3))
    npos = R1 * R2
    with pt.CGPtychoSolver(npos, 256, 256, 1, nz, n) as slv:
        slv.verbose, slv.log_every = False, 1
        gen = torch.Generator(device="cuda").manual_seed(5)
        y = torch.view_as_complex(torch.randn((1, npos, 256, 256, 2), generator=gen, device="cuda"))
        ax = slv.fwd(psi, scan, prb)
        lhs = _dot(ax, y, 4096)
        aty = slv.adj(y, scan, prb)
        r1 = complex(torch.sum(psi.to(torch.complex128) * aty.conj().to(torch.complex128)))
        bty = slv.adj_probe(y, scan, psi)
        r2 = complex(torch.sum(prb.to(torch.complex128) * bty.conj().to(torch.complex128)))
        assert abs(lhs - r1) / abs(lhs) < 1e-5 and abs(lhs - r2) / abs(lhs) < 1e-5, (lhs, r1, r2)
        del y, aty, bty
        data = torch.empty((1, npos, 256, 256), dtype=torch.float32, device="cuda")
        for i in range(0, npos, 4096):
            data[:, i:i + 4096] = torch.abs(ax[:, i:i + 4096]) ** 2
        del ax
        torch.cuda.empty_cache()
        out = slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=6)
        costs = np.array([h[3] for h in slv.history])
        assert len(costs) == 6 and np.all(costs[1:] <= costs[:-1] * (1 + 1e-6)), costs
        assert bool(torch.isfinite(torch.view_as_real(out["psi"])).all())


def test_cfg3_full_size_cg_properties(pt):
    """configs[2] CG at full size (4096 x 512^2, the four Hermite modes, phase-screened so that the
    trajectory is well conditioned): (1) the logged cost never increases over 6 iterations and every
    line search succeeds; (2) the fused multi-mode loop equals the statement-by-statement loop (HIP
    operators + the reference's expressions in torch) over 2 iterations: identical step sizes, cost
    within 1e-4, psi within 2e-4."""
    import torch
    M, ndet = 4, 512
    p = syn.make_problem(64, 64, 8, ndet, ndet, seed=31, nz=1024, n=1024)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi_true, scan = dev(p["psi"]), dev(p["scan"])
    modes = dev(phase_screen(syn.hermite_modes(ndet, M), 32))
    with pt.CGPtychoSolver(4096, ndet, ndet, 1, 1024, 1024) as slv:
        slv.verbose, slv.log_every = False, 1
        data = torch.zeros((1, 4096, ndet, ndet), dtype=torch.float32, device="cuda")
        for k in range(M):
            g = slv.fwd(psi_true, scan, modes[:, k].contiguous())
            data += torch.abs(g) ** 2
        del g
        torch.cuda.empty_cache()
        out = slv.run(data, torch.ones_like(psi_true), scan.clone(), modes.clone(), piter=6)
        hist = list(slv.history)
        costs = np.array([h[3] for h in hist])
        assert len(hist) == 6 and np.all(np.isfinite(costs))
        assert np.all(costs[1:] <= costs[:-1] * (1 + 1e-6)), costs
        assert all(h[1] > 0 for h in hist), hist
        assert bool(torch.isfinite(torch.view_as_real(out["psi"])).all())
        del out
        res = []
        for fused in (True, False):
            slv.fused = fused
            slv.history = []
            r = slv.run(data, torch.ones_like(psi_true), scan.clone(), modes.clone(), piter=2)
            res.append((r["psi"].clone(), list(slv.history)))
            del r
            torch.cuda.empty_cache()
        (pf, hf), (pu, hu) = res
        check_history(hf, hu)
        d = float(torch.abs(pf - pu).max() / torch.abs(pu).max())
        assert d < 2e-4, d


def test_cfg5_angle_stream_full_size_equals_per_angle_solves(pt):
    """configs[4] (angle streaming), per-angle problem at full size: ``run_batch`` over 4 angles of
    4096 x 256^2 (1 GiB of data each, staged through the pinned double buffer), split over two
    ``angle_shard``s as two GPUs would take them, equals four separate one-angle solves bit for bit
    (the fused loops run on the order-free adjoints and reductions)."""
    import torch
    nang, piter = 4, 3
    p = syn.make_problem(64, 64, 8, 256, 256, seed=41, nz=768, n=768, ntheta=nang)
    probe = phase_screen(np.repeat(p["probe"][:1, None], nang, 0), 42)        # [nang, 1, 256, 256]
    probe = np.ascontiguousarray(probe * np.exp(0.05j * np.arange(nang))[:, None, None, None]).astype(np.complex64)
    data = np.empty((nang, 4096, 256, 256), np.float32)
    with pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768) as slv:
        slv.verbose = False
        for a in range(nang):
            g = slv.fwd(torch.as_tensor(p["psi"][a:a + 1], device="cuda"), torch.as_tensor(p["scan"][a:a + 1], device="cuda"),
                        torch.as_tensor(probe[a:a + 1, 0], device="cuda"))
            data[a] = (torch.abs(g) ** 2)[0].cpu().numpy()
        del g
        torch.cuda.empty_cache()
        psi0 = np.ones_like(p["psi"])
        parts = [slv.run_batch(data, psi0, p["scan"].copy(), probe.copy(), piter=piter, angle_shard=(r, 2))
                 for r in (0, 1)]
        got = parts[0]["psi"].copy()
        got[1::2] = parts[1]["psi"][1::2]
        # a shard leaves the other shard's angles at their input values
        assert np.array_equal(parts[0]["psi"][1::2], psi0[1::2]) and np.array_equal(parts[1]["psi"][0::2], psi0[0::2])
        for a in range(nang):
            one = slv.run_batch(data[a:a + 1], psi0[a:a + 1], p["scan"][a:a + 1].copy(), probe[a:a + 1].copy(), piter=piter)
            assert np.array_equal(one["psi"][0], got[a]), (a, np.abs(one["psi"][0] - got[a]).max())
            assert np.abs(got[a] - 1).max() > 1e-3          # the solve did move the object


def test_bench_problem_full_size_is_reproducible_and_descends(pt):
    """The problem ``bench.py`` times (4096 x 256^2, smooth Gaussian probe, flat start, 50 iterations):
    two runs are bitwise equal (object, step sizes, logged costs -- the adjoints and every scalar
    reduction are order free), the logged cost never increases and no line search fails."""
    import torch
    p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi_true, scan, probe = dev(p["psi"]), dev(p["scan"]), dev(p["probe"][:, None])
    with pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768) as slv:
        slv.verbose, slv.log_every = False, 1
        data = (torch.abs(slv.fwd(psi_true, scan, probe[:, 0])) ** 2).contiguous()
        runs = []
        for _ in range(2):
            slv.history = []
            r = slv.run(data, torch.ones_like(psi_true), scan.clone(), probe.clone(), piter=50)
            runs.append((r["psi"].clone(), list(slv.history)))
        (pa, ha), (pb, hb) = runs
        assert ha == hb
        assert bool(torch.equal(pa, pb))
        costs = np.array([h[3] for h in ha])
        assert np.all(costs[1:] <= costs[:-1] * (1 + 1e-6)), costs
        assert all(h[1] > 0 for h in ha), ha
