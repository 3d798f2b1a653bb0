"""Randomised agreement of the DP's parallel blocks with the whole chain (PHI_DP_NOBLOCKS=1): walk lanes up to 64 walks,
class lanes for 65-256 walks, with block lengths forced to 1 .. 12 steps or left to the solve, dense graphs (few cuts,
many classes: the fallbacks) and sparse ones (k ~ 5, w ~ 20: many cuts), the solve's bookkeeping on the device copy of
the anchors or on the host copy.  Objective, bound, proof flag and counters must agree; the path is evaluated on the
restated model.
Usage (GPU box): python tests/fuzz/fuzz_dp_blocks.py SEED SECONDS [MIN_WALKS MAX_WALKS]  (default 2 257)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from graphgen import random_graph, mosaic_reads
import test_gpu_parity as T
from oracle import oracle as O, solve_oracle as S
seed0 = int(sys.argv[1]); t_end = time.time() + float(sys.argv[2])
lo_w, hi_w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 257)
s = seed0 * 100000; n = 0; modes = {}
while time.time() < t_end:
    s += 1
    rng = np.random.default_rng(s)
    nw = int(rng.integers(lo_w, hi_w))
    sparse = bool(rng.integers(0, 2))
    k, w = (int(rng.integers(4, 8)), int(rng.integers(12, 28))) if sparse else (int(rng.integers(5, 16)), int(rng.integers(1, 9)))
    g = random_graph(rng, n_sites=int(rng.integers(10, 150)), n_walks=nw, seg_len=(int(rng.integers(2, 9)), int(rng.integers(10, 45))),
                     alt_len=(1, int(rng.integers(2, 12))), p_del=float(rng.choice([0, 0.2, 0.4])))
    if rng.integers(0, 4) == 0 and nw > 2:
        g.paths[1] = g.paths[1][: max(2, len(g.paths[1]) - int(rng.integers(1, 6)))]       # a walk that ends inside the graph
    reads = mosaic_reads(rng, g, n_reads=int(rng.integers(20, 300)), read_len=int(rng.integers(k + w + 5, 140)), n_seg=int(rng.integers(1, 6)), err=float(rng.choice([0, 0.01])))
    R = int(rng.choice([0, 1, 2, 5, 20, 100])); Tt = float(rng.choice([1.0, 0.6, 0.9]))
    steps = rng.choice(["1", "2", "3", "5", "8", "12", ""])
    dev = str(int(rng.integers(0, 2)))
    out = {}
    for mode in ("blocks", "whole"):
        os.environ["PHI_SOLVE_DEVICE"] = dev
        if mode == "whole": os.environ["PHI_DP_NOBLOCKS"] = "1"
        elif steps: os.environ["PHI_DP_BLOCK_STEPS"] = str(steps)
        ctx = phi_amd.Context(0); ctx.set_params(k=k, w=w, threshold=Tt, recombination=R)
        ctx.set_solve_budget(24)
        T._set_graph(ctx, g); ctx.add_reads(reads)
        res = ctx.solve()
        info = ctx.solve_stats()
        out[mode] = (res, info)
        ctx.close()
        for name in ("PHI_DP_NOBLOCKS", "PHI_DP_BLOCK_STEPS", "PHI_SOLVE_DEVICE"): os.environ.pop(name, None)
    (a, ia), (b, ib) = out["blocks"], out["whole"]
    modes[ia["dp_mode"]] = modes.get(ia["dp_mode"], 0) + 1
    key = (s, nw, k, w, R, Tt, steps, dev, ia)
    for f in ("spectrum_size", "filtered", "n_in_model"): assert a[f] == b[f], (key, f)
    if a["optimal"] and b["optimal"]: assert a["objective"] == b["objective"], (key, a["objective"], b["objective"])
    else: assert a["objective"] <= b["upper_bound"] and b["objective"] <= a["upper_bound"], key
    if n % 8 == 0:
        st = O.run_stage12(g, reads, k, w, Tt)
        m = S.Model(g, st, R)
        for res in (a, b):
            obj, cov, nsw = m.objective(S.states_from_path(res["path_vtx"], res["path_hap"]))
            assert obj == res["objective"] and obj <= res["upper_bound"], key
    n += 1
print("fuzz_dp_blocks ok:", n, "graphs with", lo_w, "-", hi_w, "walks; DP modes of the block runs:", modes)
