"""Randomised comparison of the read path (phi_add_reads_device: 2-bit + byte-wise sketch, spectrum set, hit flags)
with the oracle: random k, w, graphs, read sets with N / lower case, several batches and resets per context.
Usage (GPU box): python tests/fuzz/fuzz_probe.py SEED SECONDS      -- not collected by pytest."""
import os, sys, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from phi_amd import dist as pdist
from graphgen import random_graph, mosaic_reads
from oracle import oracle as O
orc = O
def sketch(seq, k, w): return orc.sketch(seq, k, w)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
n_ok = 0
while time.time() < t_end:
    k = int(rng.choice([31, 31, 31, int(rng.integers(1, 33)), int(rng.integers(33, 65))])); w = int(rng.choice([25, 25, int(rng.integers(1, 70))]))
    g = random_graph(rng, n_sites=int(rng.integers(3, 40)), n_walks=int(rng.integers(1, 9)), seg_len=(1, int(rng.integers(5, 120))), alt_len=(1, int(rng.integers(2, 40))))
    ctx = phi_amd.Context(0); ctx.set_params(k=k, w=w, threshold=1.0, recombination=5)
    _st = torch.cuda.Stream(); torch.cuda.set_stream(_st)
    ctx.set_stream(_st.cuda_stream)
    A = g.arrays()
    try:
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    except Exception as e:
        ctx.close(); continue
    walk_h = np.concatenate([sketch(b"".join(g.node_seq[v] for v in path), k, w)[0] for path in g.paths]) if g.paths else np.zeros(0, np.uint64)
    _, first = np.unique(walk_h, return_index=True); uniq = walk_h[np.sort(first)]
    for gen in range(int(rng.integers(1, 4))):
        ctx.reset_reads()
        allreads = []
        for batch in range(int(rng.integers(1, 4))):
            reads = []
            if len(walk_h) and rng.random() < 0.8:
                try: reads += mosaic_reads(rng, g, n_reads=int(rng.integers(1, 60)), read_len=int(rng.integers(20, 400)), n_seg=2, err=float(rng.choice([0, 0.01, 0.05])))
                except Exception: pass
            for _ in range(int(rng.integers(0, 6))):
                L = int(rng.choice([0, 1, k - 1, k, k + w - 1, k + w, int(rng.integers(1, 4000)), int(rng.integers(4000, 60000))]))
                alpha = b"ACGT" if rng.random() < 0.7 else b"ACGTNacgtn"
                r = rng.choice(list(alpha), size=max(L, 0))
                if L > 200 and rng.random() < 0.5:             # long clean stretches with a few bases outside ACGT: chunk seams, reseeds
                    r = rng.choice(list(b"ACGT"), size=L)
                    for q in rng.integers(0, L, size=int(rng.integers(1, 12))):
                        r[q] = ord("N")
                if L > 100 and rng.random() < 0.3:             # low entropy: minimisers come back to earlier values
                    r = rng.choice(list(b"AAC"), size=L)
                    for q in rng.integers(0, L, size=int(rng.integers(0, 8))):
                        r[q] = ord("n")
                reads.append(bytes(r.tolist()))
            if not reads: reads = [b"ACGT" * 30]
            off = np.zeros(len(reads) + 1, np.int64); np.cumsum([len(r) for r in reads], out=off[1:])
            if off[-1] == 0: continue
            d_b = torch.from_numpy(np.frombuffer(b"".join(reads), np.uint8).copy()).cuda(); d_o = torch.from_numpy(off).cuda()
            ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), int(off[-1])); torch.cuda.synchronize()
            allreads += reads
        per = [sketch(r, k, w)[0] for r in allreads]
        rh = np.unique(np.concatenate(per)) if per else np.zeros(0, np.uint64)
        st = ctx.reads_stats()
        assert st["n_emitted"] == sum(len(x) for x in per), ("emitted", k, w, st, sum(len(x) for x in per))
        assert st["n_distinct"] == len(rh), ("distinct", k, w, st, len(rh))
        p, n = ctx.hits_buffer()
        hit = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy() if n else np.zeros(0, np.uint8)
        assert np.array_equal(hit, np.isin(uniq, rh).astype(np.uint8)), ("hits", k, w)
        p, m = ctx.spectrum_export()
        sp = torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64) if m else np.zeros(0, np.uint64)
        assert np.array_equal(np.sort(sp), rh[~np.isin(rh, uniq)]), ("export", k, w)
        n_ok += 1
    ctx.close()
print("fuzz ok:", n_ok, "generations")
