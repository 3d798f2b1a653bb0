"""Usage (GPU box): python tests/fuzz/fuzz_full_path.py SEED SECONDS   -- not collected by pytest.
Full-path fuzz: random graphs (optionally with N / lower case in segments), random reads (with N), random
k, w, threshold, R: every stage counter against the oracle, objective against brute force when small."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from graphgen import random_graph, mosaic_reads
from oracle import oracle as O
from oracle import solve_oracle as S
import test_gpu_parity as T
seed0 = int(sys.argv[1]); t_end = time.time() + float(sys.argv[2])
n = 0; nbf = 0; ncap = 0
s = seed0 * 100000 + (int(sys.argv[3]) if len(sys.argv) > 3 else 0)      # (a third argument: cases to skip -- to come back to one case)
verbose = os.environ.get("PHI_FUZZ_VERBOSE")
while time.time() < t_end:
    s += 1
    if verbose: print("case", s, file=sys.stderr, flush=True)
    rng = np.random.default_rng(s)
    k, w = int(rng.integers(2, 10)), int(rng.integers(1, 7))
    if rng.random() < 0.2: k, w = int(rng.integers(10, 33)), int(rng.integers(1, 40))
    rep = bytes(rng.choice(list(b"ACGT"), size=k + 3).tolist()) if rng.random() < 0.4 else None
    g = random_graph(rng, n_sites=int(rng.integers(2, 7)), n_walks=int(rng.integers(1, 6)), seg_len=(1, int(rng.integers(3, 25))), alt_len=(1, int(rng.integers(2, 10))), p_del=float(rng.choice([0, 0.2, 0.5])), repeat=rep)
    if rng.random() < 0.3:   # bases outside ACGT / lower case in the graph
        v = int(rng.integers(0, len(g.node_seq))); sq = bytearray(g.node_seq[v]); sq[int(rng.integers(0, len(sq)))] = ord(rng.choice(list("Nnacgt"))); g.node_seq[v] = bytes(sq)
    reads = mosaic_reads(rng, g, n_reads=int(rng.integers(1, 40)), read_len=int(rng.integers(k + w, k + w + 40)), n_seg=int(rng.integers(1, 4)), err=float(rng.choice([0, 0.02])))
    if rng.random() < 0.3:
        reads = [bytes(bytearray(r[:len(r)//2]) + b"N" + bytearray(r[len(r)//2:])) if rng.random() < 0.3 else r for r in reads]
    R = int(rng.choice([0, 1, 2, 3, 7, 100])); Tt = float(rng.choice([1.0, 0.5, 0.6, 2.0]))
    ctx = phi_amd.Context(0); ctx.set_params(k=k, w=w, threshold=Tt, recombination=R)
    try:
        T._set_graph(ctx, g)
    except Exception as e:
        ctx.close(); continue
    ctx.add_reads(reads)
    res0 = ctx.solve()
    if res0["optimal"] == 0:                       # the cap of 256 DP runs: a proven bound instead of a proof
        assert res0["n_dp_runs"] >= 256 and res0["objective"] <= res0["upper_bound"], (s, res0["n_dp_runs"])
        ncap += 1; ctx.close(); continue
    try:
        st, res, m = T._check_against_oracle(O, ctx, g, reads, k, w, Tt, R)
        if g.n_walks <= 4 and len(g.node_seq) <= 22:
            best, arg = m.brute_force()
            assert res["objective"] == best, ("bf", res["objective"], best)
            nbf += 1
    except AssertionError as e:
        print("FAIL seed", s, "k", k, "w", w, "R", R, "T", Tt, repr(e)[:300]); sys.exit(1)
    ctx.close(); n += 1
print("fuzz ok:", n, "cases,", nbf, "with brute force,", ncap, "at the run cap")
