"""Randomised comparison of the exact solve with HiGHS (scipy.optimize.milp, gap 0) on the reference's restated
-q0 program (oracle/solve_oracle.py build_milp): graphs of tens of vertices and up to a dozen walks, where the
relaxation sets, the certificate and branch and bound all get exercised and brute force is out of reach.
Usage (GPU box): python tests/fuzz/fuzz_vs_highs.py SEED SECONDS     -- not collected by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from graphgen import random_graph, mosaic_reads
from oracle import oracle as O
import test_gpu_parity as T

seed0 = int(sys.argv[1]); t_end = time.time() + float(sys.argv[2])
s = seed0 * 100000; n = 0; ncap = 0; ncap_opt = 0; nskip = 0; runs = []
while time.time() < t_end:
    s += 1
    rng = np.random.default_rng(s)
    k, w = int(rng.integers(3, 9)), int(rng.integers(1, 5))
    rep = bytes(rng.choice(list(b"ACGT"), size=k + int(rng.integers(1, 6))).tolist()) if rng.random() < 0.6 else None
    g = random_graph(rng, n_sites=int(rng.integers(6, 25)), n_walks=int(rng.integers(3, 12)), seg_len=(2, int(rng.integers(5, 14))),
                     alt_len=(1, int(rng.integers(2, 8))), p_del=float(rng.choice([0, 0.2, 0.4])), repeat=rep)
    reads = mosaic_reads(rng, g, n_reads=int(rng.integers(10, 80)), read_len=int(rng.integers(k + w + 3, 60)), n_seg=int(rng.integers(1, 5)), err=float(rng.choice([0, 0.02])))
    R = int(rng.choice([0, 1, 2, 3, 5, 100])); Tt = float(rng.choice([1.0, 0.5, 0.7, 2.0]))
    ctx = phi_amd.Context(0); ctx.set_params(k=k, w=w, threshold=Tt, recombination=R)
    T._set_graph(ctx, g); ctx.add_reads(reads)
    res0 = ctx.solve()
    if res0["optimal"] == 0:
        assert res0["n_dp_runs"] >= 256 and res0["objective"] <= res0["upper_bound"], (s, res0["n_dp_runs"])
        ncap += 1
        try:                                       # the incumbent and the bound against the true optimum
            from oracle import solve_oracle as S
            st = O.run_stage12(g, reads, k, w, Tt)
            best, _, _ = S.Model(g, st, R).milp_solve(time_limit=20.0)
            assert res0["objective"] <= best <= res0["upper_bound"], ("cap", res0["objective"], best, res0["upper_bound"])
            ncap_opt += res0["objective"] == best
        except RuntimeError:
            pass
        ctx.close(); continue
    try:
        st, res, m = T._check_against_oracle(O, ctx, g, reads, k, w, Tt, R)
        try:
            best, _, _ = m.milp_solve(time_limit=20.0)
        except RuntimeError:                       # HiGHS ran out of its 20 s
            nskip += 1
            ctx.close(); continue
        assert res["objective"] == best, ("highs", res["objective"], best)
    except AssertionError as e:
        # the run budget is 256 runs + 2 s of wall clock: a case that takes about 2 s can prove its optimum in
        # one solve and stop at the budget in the next (the checker solves again) -- then it is a cap case
        r2 = ctx.solve()
        if r2["optimal"] == 0 and r2["n_dp_runs"] >= 256:
            from oracle import solve_oracle as S
            try:
                best, _, _ = S.Model(g, O.run_stage12(g, reads, k, w, Tt), R).milp_solve(time_limit=20.0)
            except RuntimeError:
                best = None
            if best is None or r2["objective"] <= best <= r2["upper_bound"]:
                ncap += 1; ncap_opt += best is not None and r2["objective"] == best
                ctx.close(); continue
        print("FAIL seed", s, "k", k, "w", w, "R", R, "T", Tt, repr(e)[:300]); sys.exit(1)
    runs.append(res["n_dp_runs"])
    ctx.close(); n += 1
print("fuzz ok:", n, "cases against HiGHS,", ncap, "at the run cap (", ncap_opt, "of them with the optimal path all the same),", nskip, "HiGHS time-outs; DP runs per case: median", int(np.median(runs)) if runs else 0, "max", max(runs) if runs else 0)
