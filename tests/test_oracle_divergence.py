"""Where float32 and float64 restatements of the reference loop stop agreeing (CPU, oracle only).

VERDICT r02 asked for committed evidence behind two statements the GPU tests rely on:

* configs[1] geometry, nprb = 128 < ndet = 256, probe recovery: the probe line search of
  iteration 2 backtracks below float32 resolution of the cost -- the complex64 oracle accepts
  2^-29, the complex128 oracle fails the search (step 0).  ``test_cg256_tracks_the_oracle
  [128-True]`` therefore compares everything up to that search and only bounds what follows.
* bench.py's own problem (smooth Gaussian probe, flat start): the two oracles accept different
  object steps from iteration 1 on (the projection f/|f| takes the phase of rounding noise
  where the model predicts no amplitude), so only iteration 0 can be shared by two
  implementations; ``test_bench_problem_tracks_the_oracle_while_it_can`` checks exactly that.
"""
import numpy as np

import cg_cases as cc


def test_single_and_double_oracles_split_at_the_third_probe_search_nprb128():
    p, data, start = cc.cfg256_case(128, True)
    hs, _ = cc.oracle_history(p, data, start, 3, True, "single")
    hd, _ = cc.oracle_history(p, data, start, 3, True, "double")
    it, which = cc.CFG256_NPRB128_SPLIT
    for i in range(it):                      # identical decisions before the split
        assert hs[i][1] == hd[i][1] and hs[i][2] == hd[i][2], (i, hs[i], hd[i])
        assert abs(hs[i][3] - hd[i][3]) <= 1e-5 * abs(hd[i][3])
    assert hs[it][1] == hd[it][1]            # the object step of iteration 2 still agrees
    assert which == "prb" and hs[it][2] != hd[it][2], (hs[it], hd[it])
    # both are far below anything a float32 cost can resolve: 2^-29 against a failed search
    assert hs[it][2] < 2.0 ** -20 and hd[it][2] < 2.0 ** -20, (hs[it], hd[it])


def test_single_and_double_oracles_split_at_iteration_1_on_the_bench_problem():
    p, data, probe = cc.bench_case(8)
    hs, _ = cc.oracle_history(p, data, probe, 2, False, "single")
    hd, _ = cc.oracle_history(p, data, probe, 2, False, "double")
    assert hs[0][1] == hd[0][1] and abs(hs[0][3] - hd[0][3]) <= 1e-5 * abs(hd[0][3]), (hs[0], hd[0])
    k = cc.BENCH8_SPLIT
    assert hs[k][1] != hd[k][1], (hs[k], hd[k])
    # ... and already the cost at the start of iteration 1 differs in the second digit: the gradient of
    # iteration 0 itself depends on rounding noise
    assert abs(hs[k][3] - hd[k][3]) > 1e-3 * abs(hd[k][3]), (hs[k], hd[k])
