"""CG test problems shared by the CPU oracle tests (``test_oracle_divergence.py``) and the
GPU parity tests (``test_hip_configs.py``): the same seeded inputs on both sides.

Two of them sit where float32 and float64 arithmetic take different line-search
decisions.  The CPU tests show that with the oracle alone (``OracleSolver(precision=
"single")`` against ``("double")``); the GPU tests then demand equality with the oracle
up to -- not beyond -- the step the two oracles themselves part ways at.
"""
import numpy as np

from oracle import cg_oracle as cg
from libtike.hipfft import synthetic as syn


def phase_screen(probe, seed):
    """Random phase screen on the probe: keeps the model amplitude away from zero over the
    whole detector, so the CG trajectory is reproducible across FFT implementations."""
    rng = np.random.default_rng(seed)
    return (probe * np.exp(2j * np.pi * rng.random(probe.shape[-2:]))).astype(np.complex64)


def cfg256_case(nprb, recover):
    """configs[1] geometry (raster step 8 px + jitter, Gaussian probe with a phase screen) cut to
    8 x 8 positions; with probe recovery the start probe is the transposed true one, as in the
    reference's tests/test.py:58."""
    ndet = 256
    p = syn.make_problem(8, 8, 8, nprb, ndet, seed=11)
    probe = phase_screen(p["probe"][:, None], 111)
    data = (np.abs(cg.OracleSolver(p["nscan"], nprb, ndet, 1, p["nz"], p["n"]).fwd(
        p["psi"], p["scan"], probe[:, 0])) ** 2).astype(np.float32)
    start = probe.copy().swapaxes(2, 3) if recover else probe.copy()
    return p, data, start


def bench_case(R=8):
    """bench.py's own CG problem (SURVEY.md 8d cfg 2: smooth Gaussian probe, no phase screen,
    initial object = 1, no probe recovery) cut to R x R positions."""
    nprb = ndet = 256
    p = syn.make_problem(R, R, 8, nprb, ndet, seed=1234)
    probe = p["probe"][:, None].copy()
    data = (np.abs(cg.OracleSolver(p["nscan"], nprb, ndet, 1, p["nz"], p["n"]).fwd(
        p["psi"], p["scan"], probe[:, 0])) ** 2).astype(np.float32)
    return p, data, probe


def oracle_history(p, data, start, piter, recover, precision):
    import warnings
    ora = cg.OracleSolver(p["nscan"], start.shape[-1], data.shape[-1], 1, p["nz"], p["n"], precision=precision)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")        # a failed line search warns, like the reference
        res = ora.run(data.copy(), np.ones_like(p["psi"]), p["scan"].copy(), start.copy(),
                      piter=piter, recover_prb=recover)
    return ora.history, res


#: cfg256_case(128, True): first (iteration, search) at which single and double oracles differ
CFG256_NPRB128_SPLIT = (2, "prb")
#: bench_case(8): first iteration whose object step differs between the two oracles
BENCH8_SPLIT = 1
