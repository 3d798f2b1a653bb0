"""Randomised agreement of the DP kernels: event-driven (1 consumer wave / 2 waves / 4 waves, by walk count)
against the every-vertex kernel (PHI_DP_DENSE=1) on random graphs.
Usage (GPU box): python tests/fuzz/fuzz_dp_kernels.py SEED SECONDS [MIN_WALKS MAX_WALKS]  (default 2 257: all three kernels)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from graphgen import random_graph, mosaic_reads
import test_gpu_parity as T
seed0 = int(sys.argv[1]); t_end = time.time() + float(sys.argv[2])
lo_w, hi_w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 257)
s = seed0 * 100000; n = 0
while time.time() < t_end:
    s += 1
    rng = np.random.default_rng(s)
    nw = int(rng.integers(lo_w, hi_w))
    g = random_graph(rng, n_sites=int(rng.integers(10, 150)), n_walks=nw, seg_len=(2, int(rng.integers(4, 25))), alt_len=(1, int(rng.integers(2, 12))), p_del=float(rng.choice([0, 0.2, 0.4])))
    k, w = int(rng.integers(5, 16)), int(rng.integers(1, 9))
    reads = mosaic_reads(rng, g, n_reads=int(rng.integers(20, 300)), read_len=int(rng.integers(k + w + 5, 120)), n_seg=int(rng.integers(1, 6)), err=float(rng.choice([0, 0.01])))
    R = int(rng.choice([0, 1, 2, 5, 20, 100])); Tt = float(rng.choice([1.0, 0.6, 0.9]))
    out = {}
    for mode in ("events", "dense"):
        if mode == "dense": os.environ["PHI_DP_DENSE"] = "1"
        else: os.environ.pop("PHI_DP_DENSE", None)
        ctx = phi_amd.Context(0); ctx.set_params(k=k, w=w, threshold=Tt, recombination=R)
        T._set_graph(ctx, g); ctx.add_reads(reads)
        res = ctx.solve()
        out[mode] = (res["objective"], res["upper_bound"], res["optimal"], res["n_in_model"], res["spectrum_size"])
        ctx.close()
    os.environ.pop("PHI_DP_DENSE", None)
    if out["events"][2] == 1 and out["dense"][2] == 1:
        assert out["events"] == out["dense"], (s, nw, k, w, R, Tt, out)
    else:
        assert out["events"][3:] == out["dense"][3:] and out["events"][0] <= out["dense"][1] and out["dense"][0] <= out["events"][1], (s, out)
    n += 1
print("fuzz5 ok:", n, "graphs with", lo_w, "-", hi_w, "walks")
