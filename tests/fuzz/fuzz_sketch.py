"""Randomised comparison of phi_sketch (count / ordered write, 2-bit and byte-wise) with the oracle.
Usage (GPU box): python tests/fuzz/fuzz_sketch.py SEED SECONDS"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phi_amd
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1])); t_end = time.time() + float(sys.argv[2])
ctx = phi_amd.Context(0)
n = 0
while time.time() < t_end:
    k = int(rng.choice([31, int(rng.integers(1, 33)), int(rng.integers(1, 65))])); w = int(rng.choice([25, int(rng.integers(1, 257))]))   # (k > 32: the byte-wise routine for every window)
    seqs = []
    for _ in range(int(rng.integers(1, 12))):
        L = int(rng.choice([0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, int(rng.integers(1, 6000))]))
        alpha = [b"ACGT", b"ACGTNacgtn", b"AT", b"ACGTRYKM-*", b"AAAAAC"][int(rng.integers(0, 5))]
        seqs.append(bytes(rng.choice(list(alpha), size=max(L, 0)).tolist()))
    h, p, s = ctx.sketch(seqs, k, w)
    eh, ep, es = [], [], []
    for i, q in enumerate(seqs):
        a, b = O.sketch(q, k, w)
        eh.append(a); ep.append(b); es.append(np.full(len(a), i, np.int32))
    eh, ep, es = np.concatenate(eh), np.concatenate(ep), np.concatenate(es)
    assert len(h) == len(eh) and np.array_equal(s, es) and np.array_equal(p, ep) and np.array_equal(h, eh), (k, w, [q[:80] for q in seqs])
    n += 1
print("fuzz4 ok:", n)
