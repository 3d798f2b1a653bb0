"""GPU parity tests: the HIP path, through the C ABI, against the CPU oracle and the golden counters.

Bit-exact bar: hashes, positions, counters, anchors and objective values are integers and must be
equal.  Run on the GPU box with `pytest -m gpu`.
"""
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN
from graphgen import mosaic_reads, random_graph

pytestmark = pytest.mark.gpu


_STREAMS = []


def _bind_explicit_stream(ctx):
    """torch's copies / tensors and the context's kernels on ONE explicit stream (the null stream has handle 0,
    which the C ABI reads as "private stream": Context.set_stream refuses it)."""
    import torch
    st = torch.cuda.Stream()
    _STREAMS.append(st)                      # keep it alive for the test session
    torch.cuda.set_stream(st)
    ctx.set_stream(st.cuda_stream)
    return st


def _set_graph(ctx, g):
    A = g.arrays()
    ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])


# --------------------------------------------------------------------------- sketch kernel

@pytest.mark.parametrize("k,w", [(31, 25), (3, 2), (15, 10), (32, 1), (1, 1), (21, 64), (31, 200), (5, 9), (32, 256), (2, 256), (31, 8), (31, 9),
                                 (33, 25), (40, 7), (47, 1), (48, 16), (63, 30), (64, 256)])
def test_sketch_random_sequences(oracle, ctx_factory, k, w):
    rng = np.random.default_rng(1000 * k + w)
    ctx = ctx_factory()
    seqs = []
    for L in [0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, 150, 151, 2047, 2048, 2049, 5000, 0, 33, 12345]:
        L = max(L, 0)
        seqs.append(bytes(rng.choice(list(b"ACGT"), size=L).tolist()))
    # low-complexity and mixed-case sequences exercise ties and upper-casing
    seqs.append(b"A" * 300)
    seqs.append(b"ACACACACACACACACACACACACACACACACACACACACACACACACACACACACACACACAC" * 5)
    seqs.append(bytes(rng.choice(list(b"acgtACGT"), size=700).tolist()))
    seqs.append((b"ACGTTGCA" * 40 + b"T" * 50) * 3)
    h, p, s = ctx.sketch(seqs, k, w)
    eh, ep, es = [], [], []
    for i, q in enumerate(seqs):
        a, b = oracle.sketch(q, k, w)
        eh.append(a); ep.append(b); es.append(np.full(len(a), i, np.int32))
    eh, ep, es = np.concatenate(eh), np.concatenate(ep), np.concatenate(es)
    assert len(h) == len(eh)
    assert np.array_equal(s, es)
    assert np.array_equal(p, ep)
    assert np.array_equal(h, eh)


def test_sketch_many_short_reads(oracle, ctx_factory):
    """Ragged 150-bp-like reads crossing many chunk seams."""
    rng = np.random.default_rng(7)
    ctx = ctx_factory()
    seqs = [bytes(rng.choice(list(b"ACGT"), size=int(L)).tolist()) for L in rng.integers(40, 260, size=400)]
    h, p, s = ctx.sketch(seqs, 31, 25)
    eh = np.concatenate([oracle.sketch(q, 31, 25)[0] for q in seqs])
    ep = np.concatenate([oracle.sketch(q, 31, 25)[1] for q in seqs])
    assert np.array_equal(h, eh) and np.array_equal(p, ep)


def test_sketch_empty_and_errors(ctx_factory):
    import phi_amd
    ctx = ctx_factory()
    h, p, s = ctx.sketch([], 31, 25)
    assert len(h) == 0
    h, p, s = ctx.sketch([b"", b""], 31, 25)
    assert len(h) == 0
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.sketch([b"ACGT"], 65, 25)                      # (k <= 64: 33 .. 64 through the byte-wise routine)
    assert e.value.status == phi_amd.PHI_ERR_INVALID
    # the context stays usable after a failed call
    h, p, s = ctx.sketch([b"ACGTACGTAGCTAGCTAGCTAGCATCGATCGATCAGCTAGCTAGCATCGAT"], 5, 3)
    assert len(h) > 0


@pytest.mark.parametrize("k,w", [(31, 25), (5, 3), (15, 40), (32, 2)])
def test_sketch_bytes_outside_acgt(oracle, ctx_factory, k, w):
    """N and other bytes are kept as bytes by the reference (N sorts between G and T and is its own
    complement, ILP_index.cpp:350-353): the byte-wise path must agree with the oracle exactly."""
    rng = np.random.default_rng(31 * k + w)
    ctx = ctx_factory()
    seqs = []
    for L, nbad in [(400, 1), (400, 7), (2000, 3), (150, 1), (150, 150), (700, 40), (3000, 1), (60, 2)]:
        a = rng.choice(list(b"ACGT"), size=L)
        idx = rng.choice(L, size=min(nbad, L), replace=False)
        a[idx] = rng.choice(list(b"NnRYKMxX*-.a"), size=len(idx))
        seqs.append(bytes(a.tolist()))
    seqs.append(b"N" * 200)
    seqs.append(b"ACGT" * 30 + b"N" + b"TTGCA" * 30)
    seqs.append(bytes(rng.choice(list(b"ACGT"), size=1500).tolist()))      # a clean one in between
    seqs.append(b"acgtn" * 50)
    h, p, s = ctx.sketch(seqs, k, w)
    eh = np.concatenate([oracle.sketch(q, k, w)[0] for q in seqs])
    ep = np.concatenate([oracle.sketch(q, k, w)[1] for q in seqs])
    es = np.concatenate([np.full(len(oracle.sketch(q, k, w)[0]), i, np.int32) for i, q in enumerate(seqs)])
    assert np.array_equal(s, es) and np.array_equal(p, ep) and np.array_equal(h, eh)


def test_full_path_with_bytes_outside_acgt(oracle, ctx_factory):
    """Reads with N (fused byte-wise windows) and a graph whose segments hold N / lower case."""
    rng = np.random.default_rng(77)
    g = random_graph(rng, n_sites=7, n_walks=4, seg_len=(10, 30))
    reads = mosaic_reads(rng, g, n_reads=60, read_len=36, n_seg=2, err=0.01)
    reads = [bytes(bytearray(r[:7] + b"N" + r[8:])) if i % 5 == 0 else (r.lower() if i % 7 == 0 else r) for i, r in enumerate(reads)]
    ctx = ctx_factory(k=7, w=4, threshold=1.0, recombination=4)
    _set_graph(ctx, g)
    ctx.add_reads(reads[:20])
    ctx.add_reads(reads[20:])
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 7, 4, 1.0, 4)
    best, _ = m.brute_force()
    assert res["objective"] == best
    # now the graph itself carries such bases
    g2 = random_graph(np.random.default_rng(78), n_sites=6, n_walks=3, seg_len=(10, 30))
    seq = bytearray(g2.node_seq[0]); seq[3] = ord("N"); g2.node_seq[0] = bytes(seq)
    g2.node_seq[3] = g2.node_seq[3].lower()
    reads2 = mosaic_reads(np.random.default_rng(79), g2, n_reads=50, read_len=36, n_seg=2)
    ctx = ctx_factory(k=7, w=4, threshold=1.0, recombination=4)
    _set_graph(ctx, g2)
    ctx.add_reads(reads2)
    st, res, m = _check_against_oracle(oracle, ctx, g2, reads2, 7, 4, 1.0, 4)
    assert res["objective"] == m.brute_force()[0]


# --------------------------------------------------------------------------- full path, small

def _check_against_oracle(oracle, ctx, g, reads, k, w, T, R):
    from oracle import solve_oracle as S
    st = oracle.run_stage12(g, reads, k, w, T)
    res = ctx.solve()
    # stage 1: walk minimisers (hash, position)
    for hh in range(g.n_walks):
        gh, gp = ctx.walk_minimizers(hh)
        lo, hi = st.m_off[hh], st.m_off[hh + 1]
        assert np.array_equal(gh, st.m_hash[lo:hi]), f"walk {hh} hashes"
        assert np.array_equal(gp, st.m_pos[lo:hi]), f"walk {hh} positions"
    assert np.array_equal(res["n_minimizers"], st.n_minimizers)
    # stage 2: counters and kept anchors
    assert res["spectrum_size"] == len(st.spectrum)
    assert res["filtered"] == st.filtered
    assert res["retained"] == st.retained
    assert res["n_in_model"] == st.n_in_model
    assert np.array_equal(res["n_anchors"], st.n_anchors)
    kh, kw, k0, k1 = ctx.kept_anchors()
    exp = sorted(zip(st.spectrum[st.a_r].tolist(), st.a_h.tolist(), st.a_t0.tolist(), st.a_t1.tolist()))
    got = sorted(zip(kh.tolist(), kw.tolist(), k0.tolist(), k1.tolist()))
    assert got == exp
    # stage 3: the decoded path is feasible on the restated model and attains the reported objective
    m = S.Model(g, st, R)
    states = S.states_from_path(res["path_vtx"], res["path_hap"])
    obj, cov, nsw = m.objective(states)
    assert obj == res["objective"]
    assert cov == res["n_covered"]
    assert nsw == res["n_switches"]
    assert res["optimal"] == 1 and res["upper_bound"] == res["objective"]
    seq = ctx.path_sequence(res["hap_len"])
    assert seq == b"".join(g.node_seq[v] for v in res["path_vtx"])
    return st, res, m


def test_reference_toy_graph(oracle, ctx_factory):
    """test/test.gfa + test/read.fa at -k3 -w2: the reference's counters."""
    g = oracle.parse_gfa(os.path.join(DATA, "test.gfa"))
    reads = [s for _, s in oracle.read_reads(os.path.join(DATA, "read.fa"))]
    gold = json.load(open(os.path.join(GOLDEN, "counters.json")))["test_gfa_k3_w2"]
    for R in (100, 2, 0):
        ctx = ctx_factory(k=3, w=2, threshold=1.0, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        st, res, m = _check_against_oracle(oracle, ctx, g, reads, 3, 2, 1.0, R)
        assert res["n_minimizers"].tolist() == gold["n_minimizers"]
        assert res["spectrum_size"] == gold["spectrum_size"]
        assert res["n_anchors"].tolist() == gold["n_anchors"]
        assert res["filtered"] == gold["filtered"] and res["n_in_model"] == gold["n_in_model"]
        best, arg = m.brute_force()
        assert res["objective"] == best


@pytest.mark.parametrize("seed", range(40))
def test_random_small_graphs_vs_brute_force(oracle, ctx_factory, seed):
    rng = np.random.default_rng(seed)
    k, w = int(rng.integers(3, 8)), int(rng.integers(1, 5))
    rep = bytes(rng.choice(list(b"ACGT"), size=k + 3).tolist()) if seed % 2 else None
    g = random_graph(rng, n_sites=int(rng.integers(3, 6)), n_walks=int(rng.integers(2, 5)), repeat=rep)
    reads = mosaic_reads(rng, g, n_reads=25, read_len=k + w + 8, n_seg=2)
    R = int(rng.choice([0, 1, 2, 3, 100]))
    T = float(rng.choice([1.0, 0.5, 2.0]))
    ctx = ctx_factory(k=k, w=w, threshold=T, recombination=R)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, k, w, T, R)
    best, arg = m.brute_force()
    assert res["objective"] == best, (seed, k, w, R, T)


@pytest.mark.parametrize("k,w", [(4, 1), (6, 2), (9, 1), (31, 25)])
def test_walks_that_end_on_a_vertex_shorter_than_a_kmer(oracle, ctx_factory, k, w):
    """The class of a walk's LAST entry is [left base][its vertex]: when the vertex has k - 1 bases the class space holds
    exactly one k-mer, the left base's own window, whose record is dropped -- and whose bases reach past everything the
    class's entries own.  phi_class_rec_kernel used to look for the entry under that k-mer's last base anyway and ran off
    the end of the walk entries (a read of whatever lay behind the array: a GPU memory fault in one fuzz case, k = 4 on a
    last vertex of 3 bases, once the order of the allocations had changed).  Last vertices of 1 .. k + 1 bases, one and
    several walks, against the oracle."""
    rng = np.random.default_rng(40 * k + w)
    for last_len in list(range(1, min(k + 2, 12))) + [k - 1]:
        g = random_graph(rng, n_sites=4, n_walks=int(rng.integers(1, 4)), seg_len=(k, k + 20), alt_len=(1, 6))
        # the common last segment cut down to last_len bases
        sink = g.paths[0][-1]
        assert all(p[-1] == sink for p in g.paths)
        g.node_seq[sink] = g.node_seq[sink][:last_len] if len(g.node_seq[sink]) >= last_len else (g.node_seq[sink] * k)[:last_len]
        reads = mosaic_reads(rng, g, n_reads=30, read_len=k + w + 12, n_seg=2)
        ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=3)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 3)
        ctx.close()


@pytest.mark.parametrize("k,w", [(15, 12), (11, 30), (31, 25)])
def test_full_path_wide_windows(oracle, ctx_factory, k, w):
    """w > 8 takes the suffix/core/prefix kernel instances (generic and the (31,25) one) in all
    three modes; enough read bases for interior chunks (the check-free k-mer roll)."""
    rng = np.random.default_rng(100 * k + w)
    g = random_graph(rng, n_sites=40, n_walks=5, seg_len=(20, 60), alt_len=(1, 8), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=250, read_len=k + w + 60, n_seg=3, err=0.005)
    reads[3] = reads[3][:k + w - 2]                       # too short for a window
    reads[4] = b""
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=6)
    _set_graph(ctx, g)
    ctx.add_reads(reads[:100])
    ctx.add_reads(reads[100:])
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 6)
    assert res["spectrum_size"] > 50


@pytest.mark.parametrize("k,w", [(33, 12), (45, 25), (64, 5)])
def test_full_path_with_k_above_32(oracle, ctx_factory, k, w):
    """The reference is string based and takes any k (ILP_index.cpp:388-394).  k-mers of 33 .. 64 bases do not fit the
    2-bit kernels: graph side and read side take the exact byte-wise routine for every window (MurmurHash3 over up to
    four 16-byte blocks), the rest of the path is unchanged -- as long as no k-mer covers 32 vertices or more
    (PHI_ERR_UNSUPPORTED then: the DP's windows hold 31 run lengths).  Reads with lower case and N, batches that
    start mid-set."""
    rng = np.random.default_rng(500 * k + w)
    g = random_graph(rng, n_sites=30, n_walks=5, seg_len=(25, 70), alt_len=(3, 12), p_del=0.1)
    reads = mosaic_reads(rng, g, n_reads=220, read_len=k + w + 70, n_seg=3, err=0.004)
    reads[3] = reads[3][:k + w - 2]                       # too short for a window
    reads[4] = b""
    reads[5] = reads[5][:40] + b"N" + reads[5][41:]
    reads[6] = reads[6].lower()
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=6)
    _set_graph(ctx, g)
    ctx.add_reads(reads[:90])
    ctx.add_reads(reads[90:])
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 6)
    assert res["spectrum_size"] > 50
    # a second generation of reads on the same context: the slots of the first are emptied although they were never logged
    ctx.reset_reads()
    ctx.add_reads(reads[100:])
    st2 = oracle.run_stage12(g, reads[100:], k, w, 1.0)
    assert ctx.solve()["spectrum_size"] == len(st2.spectrum)


def test_k_above_32_on_one_base_vertices_is_refused(oracle, ctx_factory):
    """k = 40 on a graph chopped into 1-bp vertices: a k-mer covers 40 vertices, more than the DP's run-length window."""
    import phi_amd
    rng = np.random.default_rng(77)
    g = random_graph(rng, n_sites=40, n_walks=3, seg_len=(1, 1), alt_len=(1, 1), p_del=0.0)
    reads = mosaic_reads(rng, g, n_reads=60, read_len=70, n_seg=1, err=0.0)
    ctx = ctx_factory(k=40, w=3, threshold=1.0, recombination=6)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.solve()
    assert "spans" in str(e.value)


def test_dense_spectrum_small_window(oracle, ctx_factory):
    """w = 1 emits a minimiser per base: the read-spectrum set is sized for that (it used to be
    sized for w = 25 and overflowed), in one batch and when it grows across batches."""
    rng = np.random.default_rng(2024)
    g = random_graph(rng, n_sites=12, n_walks=3, seg_len=(30, 60), alt_len=(2, 8))
    reads = [bytes(rng.choice(list(b"ACGT"), size=1500).tolist()) for _ in range(200)]   # 300 kbases, ~all 15-mers distinct
    reads += mosaic_reads(rng, g, n_reads=40, read_len=60, n_seg=2)
    for batches in (1, 4):
        ctx = ctx_factory(k=15, w=1, threshold=1.0, recombination=3)
        _set_graph(ctx, g)
        step = (len(reads) + batches - 1) // batches
        for i in range(0, len(reads), step):
            ctx.add_reads(reads[i:i + step])
        st, res, m = _check_against_oracle(oracle, ctx, g, reads, 15, 1, 1.0, 3)
        assert res["spectrum_size"] > 250_000


@pytest.mark.parametrize("seed", range(8))
def test_random_mid_graphs_vs_highs(oracle, ctx_factory, seed):
    """Graphs too large for path enumeration (25-40 sites, 4-9 walks, repeats, small R so that
    recombinations pay): the objective against HiGHS on the restated program."""
    rng = np.random.default_rng(5000 + seed)
    k, w = int(rng.integers(5, 12)), int(rng.integers(1, 7))
    rep = bytes(rng.choice(list(b"ACGT"), size=k + 4).tolist()) if seed % 2 else None
    g = random_graph(rng, n_sites=int(rng.integers(25, 41)), n_walks=int(rng.integers(4, 10)), seg_len=(3, 14),
                     alt_len=(1, 8), p_del=0.25, repeat=rep)
    reads = mosaic_reads(rng, g, n_reads=150, read_len=k + w + 25, n_seg=int(rng.integers(2, 5)), err=0.01)
    R = int(rng.choice([0, 1, 2, 4, 8]))
    T = float(rng.choice([1.0, 0.6]))
    ctx = ctx_factory(k=k, w=w, threshold=T, recombination=R)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, k, w, T, R)
    best, _, _ = m.milp_solve(time_limit=200.0)
    assert res["objective"] == best, (seed, k, w, R, T, res["objective"], best, res["n_dp_runs"])


def test_more_than_255_anchors_in_a_window(oracle, ctx_factory):
    """w = 1 on 10-12 bp vertices: ~11 anchors end per walk entry, so the 30-entry window of the
    event DP holds more than 255 and its byte counters overflow -- the exact CSR count must take
    over.  Checked against HiGHS on the restated program."""
    rng = np.random.default_rng(77)
    g = random_graph(rng, n_sites=24, n_walks=3, seg_len=(10, 12), alt_len=(10, 12), p_del=0.0)
    # reads tile a mosaic of the three walks (a third of each): every k-mer of the mosaic is seen,
    # and following it takes two recombinations right inside anchor-dense stretches
    seqs = [b"".join(g.node_seq[v] for v in g.paths[hh]) for hh in range(g.n_walks)]
    n3 = min(len(q) for q in seqs) // 3
    hap = seqs[0][:n3] + seqs[1][n3:2 * n3] + seqs[2][2 * n3:]
    reads = [hap[a:a + 90] for a in range(0, len(hap) - 20, 15)]
    for R in (1, 6):
        ctx = ctx_factory(k=21, w=1, threshold=100.0, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        st, res, m = _check_against_oracle(oracle, ctx, g, reads, 21, 1, 100.0, R)
        per_entry = res["n_anchors"].max() / (max(len(p) for p in g.paths) / 3)
        assert per_entry * 30 > 255, per_entry                 # the overflow path is really taken
        if R == 1:
            assert res["n_switches"] >= 2
        best, _, _ = m.milp_solve(time_limit=200.0)
        assert res["objective"] == best, (R, res["objective"], best)


def test_degenerate_inputs(oracle, ctx_factory):
    """Single walk; reads too short for any window; thresholds that filter everything / nothing;
    odd R (the reference halves it twice, ILP_index.cpp:1276,1299)."""
    rng = np.random.default_rng(99)
    g1 = random_graph(rng, n_sites=6, n_walks=1, seg_len=(8, 20))
    reads = mosaic_reads(rng, g1, n_reads=30, read_len=30, n_seg=1)
    ctx = ctx_factory(k=7, w=3, threshold=1.0, recombination=5)
    _set_graph(ctx, g1)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g1, reads, 7, 3, 1.0, 5)
    assert res["n_switches"] == 0
    # reads without a single window: empty spectrum, objective 0, the path is still a walk
    g = random_graph(rng, n_sites=6, n_walks=4, seg_len=(8, 20))
    short = [b"ACGTACG", b"", b"AC", b"ACGTACGTA"]
    ctx = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(ctx, g)
    ctx.add_reads(short)
    st, res, m = _check_against_oracle(oracle, ctx, g, short, 7, 4, 1.0, 5)
    assert res["spectrum_size"] == 0 and res["objective"] == 0
    reads = mosaic_reads(rng, g, n_reads=40, read_len=30, n_seg=2)
    for T, R in [(0.0, 5), (0.26, 5), (100.0, 5), (1.0, 1), (1.0, 7)]:
        ctx = ctx_factory(k=7, w=4, threshold=T, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        st, res, m = _check_against_oracle(oracle, ctx, g, reads, 7, 4, T, R)
        assert res["objective"] == m.brute_force()[0], (T, R)
        if T == 0.0:
            assert res["retained"] == 0 and res["objective"] == 0


def test_walks_ending_inside_the_graph(oracle, ctx_factory):
    """Walks that stop at a vertex with successors must stop there (sink row, ILP_index.cpp:1388-1401)."""
    rng = np.random.default_rng(123)
    g = random_graph(rng, n_sites=7, n_walks=4, seg_len=(8, 16))
    g.paths[1] = g.paths[1][: len(g.paths[1]) // 2]       # ends on an interior vertex
    g.paths[3] = g.paths[3][: len(g.paths[3]) - 2]
    reads = mosaic_reads(rng, g, n_reads=40, read_len=28, n_seg=2)
    for R in (0, 4):
        ctx = ctx_factory(k=5, w=2, threshold=1.0, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        st, res, m = _check_against_oracle(oracle, ctx, g, reads, 5, 2, 1.0, R)
        assert res["objective"] == m.brute_force()[0]


@pytest.mark.parametrize("n", [40, 100])
def test_walks_that_share_nothing(oracle, ctx_factory, n):
    """Walks through private 400-bp branches: the walk-minimiser table's first sizing (32x the
    records of an average walk) is too small -- with 40 walks it is re-inserted at a larger size, with
    100 the first build overflows and falls back to the records-sized table."""
    rng = np.random.default_rng(4040)

    def rseq(m):
        return bytes(rng.choice(list(b"ACGT"), size=m).tolist())
    node_seq = [rseq(20)] + [rseq(400) for _ in range(n)] + [rseq(20)]
    adj = [list(range(1, n + 1))] + [[n + 1] for _ in range(n)] + [[]]
    g = oracle.Graph(seg_names=[f"s{i}" for i in range(n + 2)], node_seq=node_seq, adj=adj,
                     paths=[[0, i + 1, n + 1] for i in range(n)], hap_names=[f"h{i}.0" for i in range(n)])
    oracle.kahn(g)
    reads = [g.node_seq[0] + g.node_seq[7][:200], g.node_seq[7][150:] + g.node_seq[n + 1], g.node_seq[23][50:300]]
    ctx = ctx_factory(k=15, w=5, threshold=1.0, recombination=3)
    _set_graph(ctx, g)
    hist, n_distinct = ctx.walk_sharing(n)
    assert hist[1] > 30 * hist[n]                              # almost everything is private to one walk
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 15, 5, 1.0, 3)
    assert res["path_hap"][1] == 6 and res["n_switches"] == 0  # walk 6 carries branch vertex 7


def test_walk_sharing_histogram(oracle, ctx_factory):
    """The reference's -d1 report (ILP_index.cpp:565-604): distinct walk minimisers by the number of
    walks they occur in, against a recount from the oracle's per-walk sketches."""
    rng = np.random.default_rng(8)
    g = random_graph(rng, n_sites=30, n_walks=6, seg_len=(10, 40), alt_len=(1, 10))
    k, w = 11, 5
    ctx = ctx_factory(k=k, w=w)
    _set_graph(ctx, g)
    hist, n_distinct = ctx.walk_sharing(g.n_walks)
    per_walk = [np.unique(oracle.sketch(b"".join(g.node_seq[v] for v in p), k, w)[0]) for p in g.paths]
    allh, counts = np.unique(np.concatenate(per_walk), return_counts=True)
    want = np.bincount(counts, minlength=g.n_walks + 1)
    assert n_distinct == len(allh)
    assert hist.tolist() == want.tolist() and hist[0] == 0 and hist.sum() == n_distinct


def test_streaming_batches_equal_one_batch(oracle, ctx_factory):
    rng = np.random.default_rng(5)
    g = random_graph(rng, n_sites=8, n_walks=4, seg_len=(8, 30))
    reads = mosaic_reads(rng, g, n_reads=60, read_len=30, n_seg=3, err=0.02)
    a = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(a, g)
    a.add_reads(reads)
    ra = a.solve()
    b = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(b, g)
    for i in range(0, len(reads), 7):
        b.add_reads(reads[i:i + 7])
    b.add_reads([])
    rb = b.solve()
    for key in ("objective", "spectrum_size", "filtered", "n_in_model", "n_covered"):
        assert ra[key] == rb[key]
    assert np.array_equal(ra["path_vtx"], rb["path_vtx"]) and np.array_equal(ra["path_hap"], rb["path_hap"])
    # reset forgets the reads but keeps the index
    b.reset_reads()
    b.add_reads(reads)
    rc = b.solve()
    assert rc["objective"] == ra["objective"] and rc["spectrum_size"] == ra["spectrum_size"]


@pytest.mark.parametrize("waves", ["1", "3", "64", "6144"])
def test_pooled_read_kernel_on_small_batches(oracle, ctx_factory, monkeypatch, waves):
    """Read batches of 12 Mbases and more go through phi_sketch_pool_kernel: a wave takes several chunks (g, g + waves, ...)
    and hashes their items in full rounds of 64, what a round leaves over waiting in registers for the next chunk's.
    PHI_SKETCH_POOL_MIN=1 sends every batch that way and PHI_SKETCH_WAVES sets how many waves share a batch (1: one wave
    takes every chunk; more waves than chunks: waves with nothing to do), so that the batches the oracle finishes exercise
    the rounds that span chunks, the last partial round, the per-wave slot log and its use by the next reset.  Reads with
    bases outside ACGT, reads shorter than a window, of one length (no offsets) and of mixed lengths; several batches and
    a reset between them; k and w of the specialised and of the generic instances."""
    monkeypatch.setenv("PHI_SKETCH_POOL_MIN", "1")
    monkeypatch.setenv("PHI_SKETCH_WAVES", waves)
    rng = np.random.default_rng(9100 + int(waves))
    for (k, w) in ((31, 25), (15, 10), (7, 3), (32, 9)):
        g = random_graph(rng, n_sites=12, n_walks=4, seg_len=(30, 80), alt_len=(2, 9))
        reads = mosaic_reads(rng, g, n_reads=160, read_len=150, n_seg=2, err=0.02)
        reads += [bytes(rng.choice(list(b"ACGTNacgtn"), size=int(rng.integers(1, 400))).tolist()) for _ in range(60)]
        reads += [bytes(rng.choice(list(b"ACGT"), size=150).tolist()) for _ in range(700)]        # several chunks of novel sequence: rounds full of new slots
        uniform = [r for r in reads if len(r) == 150]
        ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=5)
        _set_graph(ctx, g)
        for batch in (uniform, reads[:50], reads[50:]):
            ctx.add_reads(batch)
        sk = [oracle.sketch(r, k, w)[0] for r in uniform + reads]
        st = ctx.reads_stats()
        assert st["n_emitted"] == sum(len(x) for x in sk) and st["n_distinct"] == len(np.unique(np.concatenate(sk)))
        # a reset (the next launch empties what this generation logged), then the mixed reads alone, against the oracle
        ctx.reset_reads()
        ctx.add_reads(reads[:90])
        ctx.add_reads(reads[90:])
        _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 5)


@pytest.mark.parametrize("n_walks", [70, 130, 300, 600, 1022])
def test_more_than_64_walks_vs_highs(oracle, ctx_factory, n_walks):
    """lane <-> walk: more than 64 walks take the multi-wave instances of the DP kernel (600 and 1 022: the sixteen-wave
    instance, whose entry words come from HBM and whose LDS rings are shorter).  Brute
    force is out of reach with this many labels; the objective is checked against HiGHS on the
    reference's restated -q0 program (oracle/solve_oracle.py build_milp)."""
    rng = np.random.default_rng(1000 + n_walks)
    g = random_graph(rng, n_sites=8, n_walks=n_walks, seg_len=(6, 12), alt_len=(2, 5), p_del=0.3)
    reads = mosaic_reads(rng, g, n_reads=60, read_len=28, n_seg=4)
    R = 4
    ctx = ctx_factory(k=5, w=2, threshold=0.6, recombination=R)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 5, 2, 0.6, R)
    assert res["objective"] < res["n_in_model"]          # switching costs bite
    best, _, _ = m.milp_solve(time_limit=120.0)
    assert res["objective"] == best, (n_walks, res["objective"], best)


def test_more_walks_than_the_largest_workgroup_has_lanes_are_refused(ctx_factory):
    """1 023 walks: one more than a walk id has room for in the DP's packed tops -- PHI_ERR_UNSUPPORTED with a message, at
    phi_set_graph (the reference has no limit: include/phi_amd.h says so)."""
    import phi_amd
    rng = np.random.default_rng(77)
    g = random_graph(rng, n_sites=3, n_walks=1023, seg_len=(6, 12), alt_len=(2, 5), p_del=0.3)
    ctx = ctx_factory(k=5, w=2, threshold=0.6, recombination=4)
    with pytest.raises(phi_amd.PhiError) as e:
        _set_graph(ctx, g)
    assert "more than 1022 walks" in str(e.value)


def test_dense_and_event_dp_agree(oracle, ctx_factory, monkeypatch):
    """PHI_DP_DENSE=1 (read by phi_set_graph) forces the every-vertex kernel of dp.hip that otherwise
    only serves more than 128 walks; both kernels must give the same objective and a feasible path."""
    rng = np.random.default_rng(4242)
    g = random_graph(rng, n_sites=60, n_walks=9, seg_len=(4, 12), alt_len=(2, 6), p_del=0.25)
    reads = mosaic_reads(rng, g, n_reads=200, read_len=40, n_seg=4, err=0.01)
    out = {}
    for mode in ("events", "dense"):
        if mode == "dense":
            monkeypatch.setenv("PHI_DP_DENSE", "1")
        for R in (0, 3, 100):
            ctx = ctx_factory(k=7, w=3, threshold=0.8, recombination=R)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            st, res, m = _check_against_oracle(oracle, ctx, g, reads, 7, 3, 0.8, R)
            out[(mode, R)] = res["objective"]
    monkeypatch.delenv("PHI_DP_DENSE")
    for R in (0, 3, 100):
        assert out[("events", R)] == out[("dense", R)], (R, out)


@pytest.mark.parametrize("k,w,seg_len", [(9, 2, (8, 16)), (15, 8, (3, 25))])
def test_anchors_from_packed_selected_records_equal_the_generic_expansion(oracle, ctx_factory, monkeypatch, k, w, seg_len):
    """phi_solve expands the model's anchors from the filter's selected class records packed per class (a block's anchors
    staged in LDS when they are at most 1 536, written directly otherwise: dense minimisers, k = 9 / w = 2, take the second
    branch); PHI_EXPAND_GENERIC=1 walks every record of every entry's class as before.  Same anchors in the same order, and
    the oracle's kept anchors."""
    rng = np.random.default_rng(6100 + k)
    g = random_graph(rng, n_sites=1500, n_walks=20, seg_len=seg_len, alt_len=(1, 12), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=1200, read_len=100, n_seg=5, err=0.01)
    got = {}
    for mode in ("packed", "generic"):
        if mode == "generic":
            monkeypatch.setenv("PHI_EXPAND_GENERIC", "1")
        ctx = ctx_factory(k=k, w=w, threshold=0.8, recombination=10)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        res = ctx.solve()
        got[mode] = (ctx.kept_anchors(), res["objective"], res["n_anchors"].tolist())
        monkeypatch.delenv("PHI_EXPAND_GENERIC", raising=False)
        if mode == "packed":
            _check_against_oracle(oracle, ctx, g, reads, k, w, 0.8, 10)
    for a, b in zip(got["packed"][0], got["generic"][0]):
        assert np.array_equal(a, b)
    assert got["packed"][1:] == got["generic"][1:]
    assert len(got["packed"][0][0]) > 20000


@pytest.mark.parametrize("n_walks,k,w", [(20, 15, 8), (100, 15, 8), (30, 9, 2)])
def test_cut_flags_without_counting_place_the_same_blocks(ctx_factory, monkeypatch, n_walks, k, w):
    """Where the DP's chain may be cut: an entry no anchor reaches across.  phi_solve marks what the anchors ending on a tile of
    entries cover (LDS, no atomics); PHI_CUT_COUNTED=1 counts coverage as before (two atomics per anchor, a prefix sum).  The same
    flags give the same blocks (PHI_DP_BLOCK_STEPS=5: a cut wherever one is allowed), objective and path."""
    rng = np.random.default_rng(8200 + n_walks)
    g = random_graph(rng, n_sites=1500, n_walks=n_walks, seg_len=(3, 25), alt_len=(1, 12), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=1200, read_len=100, n_seg=5, err=0.01)
    monkeypatch.setenv("PHI_DP_BLOCK_STEPS", "5")
    got = {}
    for mode in ("direct", "counted"):
        if mode == "counted":
            monkeypatch.setenv("PHI_CUT_COUNTED", "1")
        ctx = ctx_factory(k=k, w=w, threshold=0.8, recombination=10)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        res = ctx.solve()
        st = ctx.solve_stats()
        got[mode] = (st["n_blocks"], st["dp_mode"], res["objective"], res["path_vtx"].tolist(), res["path_hap"].tolist())
        monkeypatch.delenv("PHI_CUT_COUNTED", raising=False)
    assert got["direct"] == got["counted"]
    assert (n_walks, k) != (20, 15) or got["direct"][0] > 5          # (this one is cut into blocks)


def test_sixteen_wave_dense_dp_agrees_with_the_event_dp(oracle, ctx_factory, monkeypatch):
    """The dense kernel's sixteen-wave instance (513..1022 walks: entry words from HBM, step stream in chunks of 32, the last
    256 steps' leaving states in LDS and older ones from HBM) forced onto a graph of thousands of vertices and 100 walks
    (PHI_DP_WAVES=16 lays the step masks out for sixteen waves too): the event DP's objective, a feasible path."""
    rng = np.random.default_rng(9103)
    g = random_graph(rng, n_sites=1500, n_walks=100, seg_len=(3, 25), alt_len=(1, 12), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=1200, read_len=100, n_seg=5, err=0.01)
    k, w = 15, 8
    out = {}
    for R in (2, 40):
        for mode in ("events", "dense16"):
            if mode == "dense16":
                monkeypatch.setenv("PHI_DP_DENSE", "1")
                monkeypatch.setenv("PHI_DP_WAVES", "16")
            ctx = ctx_factory(k=k, w=w, threshold=0.8, recombination=R)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            res = ctx.solve()
            assert res["optimal"] == 1
            out[(mode, R)] = (res["objective"], res["recombination_count"])
            monkeypatch.delenv("PHI_DP_DENSE", raising=False)
            monkeypatch.delenv("PHI_DP_WAVES", raising=False)
        assert out[("events", R)][0] == out[("dense16", R)][0], (R, out)


@pytest.mark.parametrize("seed,n_walks", [(1, 60), (2, 33), (3, 100), (4, 200), (5, 256)])
def test_dense_and_event_dp_agree_at_scale(oracle, ctx_factory, monkeypatch, seed, n_walks):
    """Two independent DP kernels (event-driven with prefix sums / every-vertex with a difference
    ring) on graphs with thousands of vertices and tens of walks: same objective, same proof."""
    rng = np.random.default_rng(9000 + seed)
    g = random_graph(rng, n_sites=1500, n_walks=n_walks, seg_len=(3, 25), alt_len=(1, 12), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=1200, read_len=100, n_seg=5, err=0.01)
    k, w, R = 15, 8, int(rng.choice([2, 10, 40]))
    out = {}
    for mode in ("events", "dense"):
        if mode == "dense":
            monkeypatch.setenv("PHI_DP_DENSE", "1")
        ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        res = ctx.solve()
        out[mode] = (res["objective"], res["upper_bound"], res["optimal"], res["spectrum_size"], res["n_in_model"])
    monkeypatch.delenv("PHI_DP_DENSE")
    assert out["events"] == out["dense"], out
    # and the path is feasible on the restated model with the reported value
    from oracle import solve_oracle as S
    st = oracle.run_stage12(g, reads, k, w, 1.0)
    m = S.Model(g, st, R)
    obj, cov, nsw = m.objective(S.states_from_path(res["path_vtx"], res["path_hap"]))
    assert obj == res["objective"]


def test_event_dp_queue_overflow_falls_back_to_dense(oracle, ctx_factory, monkeypatch):
    """129-256 walks: the four-wave event kernel keeps 16 live runs per lane; a lane that needs more raises
    PHI_KERR_DP_QUEUE and the solve reruns on the every-vertex kernel.  PHI_DP_QLIMIT=1 provokes it."""
    rng = np.random.default_rng(777)
    g = random_graph(rng, n_sites=300, n_walks=150, seg_len=(3, 20), alt_len=(1, 10), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=400, read_len=80, n_seg=4, err=0.01)
    k, w, R = 11, 5, 3
    out = {}
    for mode in ("events", "fallback", "dense"):
        if mode == "fallback":
            monkeypatch.setenv("PHI_DP_QLIMIT", "1")
        if mode == "dense":
            monkeypatch.delenv("PHI_DP_QLIMIT")
            monkeypatch.setenv("PHI_DP_DENSE", "1")
        ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        res = ctx.solve()
        out[mode] = (res["objective"], res["upper_bound"], res["optimal"], res["n_in_model"], res["path_vtx"].tolist(), res["path_hap"].tolist())
    monkeypatch.delenv("PHI_DP_DENSE")
    assert out["events"][:4] == out["dense"][:4] == out["fallback"][:4], [o[:4] for o in out.values()]
    assert out["fallback"] == out["dense"]                   # the fallback IS the dense kernel: same path
    from oracle import solve_oracle as S
    st = oracle.run_stage12(g, reads, k, w, 1.0)
    m = S.Model(g, st, R)
    for mode in ("events", "dense"):
        obj, cov, nsw = m.objective(S.states_from_path(np.array(out[mode][4]), np.array(out[mode][5])))
        assert obj == out[mode][0]


def test_many_switches_backtrack(oracle, ctx_factory):
    """R = 0 on a long bubble chain: the best path switches walks hundreds of times, so the
    backtrack leaves its read-a-few-entries mode for the bulk download."""
    rng = np.random.default_rng(321)
    g = random_graph(rng, n_sites=400, n_walks=4, seg_len=(6, 14), alt_len=(3, 6), p_del=0.0)
    # truth alternates ref / alt allele site by site; the walks chose theirs at random
    succ = {v: sorted(a) for v, a in enumerate(g.adj)}
    v, truth, site = g.paths[0][0], [], 0
    while True:
        truth.append(v)
        nx = succ[v]
        if not nx:
            break
        if len(nx) == 2:
            v = nx[site % 2]
            site += 1
        else:
            v = nx[0]
    hap = b"".join(g.node_seq[x] for x in truth)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(700):
        a = int(rng.integers(0, len(hap) - 60))
        r = hap[a:a + 60]
        reads.append(r.translate(comp)[::-1] if rng.random() < 0.5 else r)
    ctx = ctx_factory(k=9, w=3, threshold=1.0, recombination=0)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 9, 3, 1.0, 0)
    assert res["n_switches"] > 64, res["n_switches"]


def test_run_budget_is_deterministic_and_reports_a_proven_bound(oracle, ctx_factory):
    """A hard instance (R = 0, eight walks, short repeats) needs many DP runs.  The budget of the exact search
    is counted in DP runs (phi_set_solve_budget), never in wall-clock time: with a small budget the result says
    optimal = 0, carries a finite proven bound the feasible path respects, and is the SAME on every run; with
    the default budget the instance is proven, and the proven optimum lies inside every earlier bound."""
    from oracle import solve_oracle as S
    rng = np.random.default_rng(321)
    g = random_graph(rng, n_sites=400, n_walks=8, seg_len=(8, 16), alt_len=(3, 6), p_del=0.0)
    succ = {v: sorted(a) for v, a in enumerate(g.adj)}
    v, truth, site = g.paths[0][0], [], 0
    while True:
        truth.append(v)
        nx = succ[v]
        if not nx:
            break
        v = nx[site % 2] if len(nx) == 2 else nx[0]
        site += len(nx) == 2
    hap = b"".join(g.node_seq[x] for x in truth)
    reads = [hap[a:a + 60] for a in rng.integers(0, len(hap) - 60, size=700)]
    st = oracle.run_stage12(g, reads, 7, 2, 1.0)
    m = S.Model(g, st, 0)

    def run(budget):
        ctx = ctx_factory(k=7, w=2, threshold=1.0, recombination=0)
        if budget is not None:
            ctx.set_solve_budget(budget)
        _set_graph(ctx, g)
        ctx.add_reads(reads)
        res = ctx.solve()
        obj, cov, nsw = m.objective(S.states_from_path(res["path_vtx"], res["path_hap"]))
        assert obj == res["objective"]
        assert res["objective"] <= res["upper_bound"] <= res["n_in_model"]
        assert res["optimal"] == (res["objective"] == res["upper_bound"])
        return res

    full = run(600)
    assert full["n_dp_runs"] <= 600
    small = [run(3) for _ in range(3)]
    for r in small:
        assert r["n_dp_runs"] <= 3
        for key in ("objective", "upper_bound", "optimal", "n_dp_runs", "n_covered", "n_switches"):
            assert r[key] == small[0][key], key
        assert np.array_equal(r["path_vtx"], small[0]["path_vtx"]) and np.array_equal(r["path_hap"], small[0]["path_hap"])
        assert r["objective"] <= full["objective"] <= r["upper_bound"] or not full["optimal"]
    if full["n_dp_runs"] > 3:
        assert small[0]["optimal"] == 0                         # it really ran out of its budget
    # a larger budget never reports a worse path or a looser bound
    assert full["objective"] >= small[0]["objective"] and full["upper_bound"] <= small[0]["upper_bound"]


def test_rccl_communicator_inside_the_library(oracle, ctx_factory):
    """phi_comm_*: librccl loaded by the library, ncclCommInitRank on the context's device, the hit-vector
    all-reduce and the spectrum all-gather on the context's stream.  One GPU here, so a communicator of one
    rank: the exchange must leave the result exactly as it was (the N > 1 arithmetic is the export / import
    pair of test_two_read_shards_merge_to_the_single_context_result and tests/test_cpu_dist.py)."""
    import phi_amd
    rng = np.random.default_rng(2)
    g = random_graph(rng, n_sites=12, n_walks=5, seg_len=(10, 40))
    reads = mosaic_reads(rng, g, n_reads=80, read_len=40, n_seg=3, err=0.02) + [bytes(rng.choice(list(b"ACGT"), size=300).tolist())]
    ctx = ctx_factory(k=9, w=5, threshold=1.0, recombination=4)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    before = ctx.solve()
    uid = phi_amd.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    assert ctx.comm_info() == (0, 1)
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.comm_exchange()                                     # no communicator yet
    assert e.value.status == phi_amd.PHI_ERR_STATE
    ctx.comm_init(uid, 0, 1)
    assert ctx.comm_info() == (0, 1)
    with pytest.raises(phi_amd.PhiError):
        ctx.comm_init(uid, 0, 1)                                # one communicator per context
    ctx.comm_allreduce_hits()
    ctx.comm_exchange()
    after = ctx.solve()
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 9, 5, 1.0, 4)
    for key in ("objective", "spectrum_size", "filtered", "n_in_model", "n_covered"):
        assert before[key] == after[key] == res[key]
    # a second generation of reads through the same communicator
    ctx.reset_reads()
    ctx.add_reads(reads[:40])
    ctx.comm_exchange()
    _check_against_oracle(oracle, ctx, g, reads[:40], 9, 5, 1.0, 4)
    ctx.comm_destroy()
    assert ctx.comm_info() == (0, 1)


def test_resets_empty_only_what_was_filled(oracle, ctx_factory):
    """A reset folded into the next preparation launch empties the logged spectrum slots instead of the
    whole set -- unless a chunk filled more slots than its log holds, the byte-wise path inserted, a list
    was imported or the set was rebuilt, in which case it empties everything.  Generation after
    generation, through each of those cases, the set and the flags must be exactly the batch's."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(99)
    k, w = 15, 6
    g = random_graph(rng, n_sites=40, n_walks=6, seg_len=(20, 60), alt_len=(1, 8))
    walk_h = np.concatenate([oracle.sketch(b"".join(g.node_seq[v] for v in path), k, w)[0] for path in g.paths])
    _, first = np.unique(walk_h, return_index=True)
    uniq = walk_h[np.sort(first)]

    def rseq(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    gens = [
        ("graph reads", mosaic_reads(rng, g, n_reads=80, read_len=90, n_seg=2)),
        ("foreign reads: every hash is inserted, far more than a chunk logs", [rseq(3000) for _ in range(6)]),
        ("graph reads again", mosaic_reads(rng, g, n_reads=60, read_len=90, n_seg=2)),
        ("reads with bytes outside ACGT: the byte-wise path inserts", [rseq(200) + b"N" + rseq(300) + b"nn" + rseq(150) for _ in range(5)]),
        ("graph reads once more", mosaic_reads(rng, g, n_reads=70, read_len=90, n_seg=3)),
        ("three batches in one generation", None),
        ("after an import", mosaic_reads(rng, g, n_reads=30, read_len=90, n_seg=2)),
        ("last", mosaic_reads(rng, g, n_reads=50, read_len=90, n_seg=2)),
    ]
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=5)
    _bind_explicit_stream(ctx)
    _set_graph(ctx, g)
    for name, reads in gens:
        ctx.reset_reads()
        if reads is None:
            a = mosaic_reads(rng, g, n_reads=40, read_len=90, n_seg=2) + [rseq(500)]
            b = mosaic_reads(rng, g, n_reads=40, read_len=90, n_seg=2) + [rseq(700)]
            big = mosaic_reads(rng, g, n_reads=600, read_len=90, n_seg=2)      # outgrows the log buffer mid-generation
            for part in (a, b, big):
                off = np.zeros(len(part) + 1, np.int64)
                np.cumsum([len(r) for r in part], out=off[1:])
                d_b = torch.from_numpy(np.frombuffer(b"".join(part), np.uint8).copy()).cuda()
                d_o = torch.from_numpy(off).cuda()
                ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), len(part), int(off[-1]))
                torch.cuda.synchronize()
            reads = a + b + big
        else:
            off = np.zeros(len(reads) + 1, np.int64)
            np.cumsum([len(r) for r in reads], out=off[1:])
            d_b = torch.from_numpy(np.frombuffer(b"".join(reads), np.uint8).copy()).cuda()
            d_o = torch.from_numpy(off).cuda()
            ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), int(off[-1]))
            torch.cuda.synchronize()
        read_h = np.unique(np.concatenate([oracle.sketch(r, k, w)[0] for r in reads]))
        st = ctx.reads_stats()
        assert st["n_distinct"] == len(read_h), name
        p, n = ctx.hits_buffer()
        hit = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy()
        assert np.array_equal(hit, np.isin(uniq, read_h).astype(np.uint8)), name
        p, m = ctx.spectrum_export()
        sp = torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone()
        assert np.array_equal(np.sort(sp.cpu().numpy().view(np.uint64)), read_h[~np.isin(read_h, uniq)]), name
        if name == "three batches in one generation":
            extra = torch.from_numpy(np.array([111, 222, 333], np.uint64).view(np.int64)).cuda()
            ctx.spectrum_import(extra.data_ptr(), 3)
            torch.cuda.synchronize()
            assert ctx.reads_stats()["n_distinct"] == len(read_h) + 3


def test_reset_is_deferred_but_never_visible(oracle, ctx_factory):
    """phi_reset_reads is folded into the next batch's preparation launch; every observer in between
    (stats, hit vector, spectrum export, solve) must still see the reads forgotten."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(15)
    g = random_graph(rng, n_sites=8, n_walks=4, seg_len=(8, 30))
    reads = mosaic_reads(rng, g, n_reads=40, read_len=30, n_seg=2)
    other = mosaic_reads(np.random.default_rng(16), g, n_reads=9, read_len=30, n_seg=1)
    fresh = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(fresh, g)
    fresh.add_reads(other)
    want = fresh.solve()
    want_stats = fresh.reads_stats()

    c = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(c, g)
    c.add_reads(reads)
    c.reset_reads()
    s0 = c.reads_stats()
    assert s0["n_reads"] == 0 and s0["n_bases"] == 0 and s0["n_emitted"] == 0 and s0["n_distinct"] == 0
    c.add_reads(reads)
    c.reset_reads()
    p, n = c.spectrum_export()
    assert n == 0
    c.add_reads(reads)
    c.reset_reads()
    assert c.solve()["spectrum_size"] == 0
    c.add_reads(reads)
    c.reset_reads()
    p, n = c.hits_buffer()                          # fetching the pointer applies the pending reset
    hit = torch.as_tensor(pdist.DevArray(p, n), device="cuda")
    torch.cuda.synchronize()
    assert int(hit.sum().item()) == 0
    c.add_reads(reads)
    torch.cuda.synchronize()
    assert int(hit.sum().item()) > 0
    c.reset_reads()
    p2, n2 = c.hits_buffer()                        # by contract the pointer is fetched again after a reset (double buffers: it moves)
    assert n2 == n
    hit = torch.as_tensor(pdist.DevArray(p2, n2), device="cuda")
    torch.cuda.synchronize()
    assert int(hit.sum().item()) == 0
    # reset + other reads == a fresh context with the other reads (twice, to exercise both parities)
    for _ in range(2):
        c.reset_reads()
        c.add_reads(other)
        got = c.solve()
        assert c.reads_stats() == want_stats
        for key in ("objective", "spectrum_size", "filtered", "n_in_model", "n_covered"):
            assert got[key] == want[key]
    # deferred reset in front of a batch with bases outside ACGT
    c2 = ctx_factory(k=7, w=4, threshold=1.0, recombination=5)
    _set_graph(c2, g)
    c2.add_reads([b"ACGTNNACGTAGCTAGCTAGGATCGATCGTAGCTAGC"] * 3)
    c2.reset_reads()
    c2.add_reads(other)
    assert c2.reads_stats() == want_stats and c2.solve()["objective"] == want["objective"]


def test_bad_inputs(oracle, ctx_factory):
    import phi_amd
    g = oracle.parse_gfa(os.path.join(DATA, "test.gfa"))
    A = g.arrays()
    ctx = ctx_factory(k=3, w=2)
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.solve()
    assert e.value.status == phi_amd.PHI_ERR_STATE
    # a walk that jumps over a missing edge (the reference exits at ILP_index.cpp:1568-1572)
    wv = A["walk_vtx"].copy()
    wv[1] = 2 if wv[1] == 1 else 1
    wv[2] = 7
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], wv, A["top_rank"])
    assert e.value.status == phi_amd.PHI_ERR_WALK
    # a walk entry that names no vertex (checked by the first kernel that reads the walks), at either end of the range
    for bad in (len(A["seq_off"]) - 1, 10 ** 6, -1):
        wv = A["walk_vtx"].copy()
        wv[3] = bad
        with pytest.raises(phi_amd.PhiError) as e:
            ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], wv, A["top_rank"])
        assert e.value.status == phi_amd.PHI_ERR_WALK and "out of range" in str(e.value)
    # a rank array that is not a topological order
    tr = A["top_rank"].copy()
    tr[[0, 7]] = tr[[7, 0]]
    with pytest.raises(phi_amd.PhiError) as e:
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], tr)
    assert e.value.status == phi_amd.PHI_ERR_INVALID
    # and the context still works afterwards
    _set_graph(ctx, g)
    ctx.add_reads([b"ATCGATCATACTTACCATG"])
    assert ctx.solve()["objective"] == 4


# --------------------------------------------------------------------------- config 1 (reference fixtures)

def test_config1_mhc4(oracle, ctx_factory):
    """test/MHC_4.gfa.gz + test/CHM13_reads.fq.gz, defaults: every counter the reference logs."""
    gold = json.load(open(os.path.join(GOLDEN, "counters.json")))["mhc4_chm13_k31_w25"]
    g = oracle.parse_gfa(os.path.join(DATA, "MHC_4.gfa.gz"))
    reads = [s for _, s in oracle.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))]
    ctx = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, g, reads, 31, 25, 1.0, 100)
    assert g.hap_names == gold["hap_names"]
    assert res["n_minimizers"].tolist() == gold["n_minimizers"]
    assert res["spectrum_size"] == gold["spectrum_size"]
    assert res["n_anchors"].tolist() == gold["n_anchors"]
    assert res["n_in_model"] == gold["n_in_model"]
    assert "%.2f/%.2f" % (res["filtered"] / res["spectrum_size"] * 100, res["retained"] / res["spectrum_size"] * 100) == gold["filtered_retained_pct"]
    assert "%.2f" % (res["n_in_model"] * 100.0 / res["spectrum_size"]) == gold["pct_in_model"]
    print("config1 objective", res["objective"], "dp runs", res["n_dp_runs"], "recomb", res["recombination_count"])
    if "objective_highs" in gold:
        assert res["objective"] == gold["objective_highs"]


# --------------------------------------------------------------------------- mid-scale solve parity

def _syn_oracle_graph(oracle, g):
    G = oracle.Graph(seg_names=[str(i) for i in range(g.n_vtx)],
                     node_seq=[bytes(g.seq_concat[g.seq_off[v]:g.seq_off[v + 1]]) for v in range(g.n_vtx)],
                     adj=[g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist() for v in range(g.n_vtx)],
                     paths=[g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.n_walks)],
                     hap_names=g.hap_names)
    oracle.kahn(G)
    return G


def test_synthetic_vs_highs_golden(oracle, ctx_factory):
    """Objective equality with HiGHS on the reference's restated -q0 program (tests/golden/
    solve_golden.json, made by tests/golden/make_solve_golden.py), at sizes brute force cannot reach."""
    from phi_amd import synth
    gold = json.load(open(os.path.join(GOLDEN, "solve_golden.json")))
    cache = {}
    for case in gold:
        name = case["config"]
        if name not in cache:
            gk, rk = synth.CONFIGS[name]
            g = synth.make_graph(**gk)
            bases, off, _ = synth.make_reads(g, **rk)
            cache[name] = (g, _syn_oracle_graph(oracle, g), bases, off)
        g, G, bases, off = cache[name]
        ctx = ctx_factory(k=case["k"], w=case["w"], threshold=case["T"], recombination=case["R"])
        A = g.arrays()
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        ctx.add_reads((bases, off))
        reads = [bytes(bases[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        st, res, m = _check_against_oracle(oracle, ctx, G, reads, case["k"], case["w"], case["T"], case["R"])
        assert res["spectrum_size"] == case["spectrum_size"] and res["n_in_model"] == case["n_in_model"]
        assert res["objective"] == case["objective"], (case, res["objective"], res["n_dp_runs"])


def _probe_counts(ctx, reads):
    import torch
    ctx.reset_reads()
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    d_b = torch.from_numpy(np.frombuffer(b"".join(reads), np.uint8).copy()).cuda()
    d_o = torch.from_numpy(off).cuda()
    ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), int(off[-1]))
    torch.cuda.synchronize()
    return ctx.reads_stats()


@pytest.mark.parametrize("k,w", [(2, 1), (5, 6), (4, 3), (3, 25), (31, 25), (7, 30), (21, 60)])
def test_probe_minimiser_returns_to_its_value_across_a_base_outside_acgt(oracle, ctx_factory, k, w):
    """Windows around a base outside ACGT go to the byte-wise path; the first 2-bit window after such a
    stretch must be compared with ITS predecessor window, not with the last 2-bit candidate before the
    stretch (`TTnTAa`, k=2, w=1: the last window's minimiser equals the first's, yet it is emitted)."""
    import torch
    rng = np.random.default_rng(100 * k + w)
    g = random_graph(rng, n_sites=5, n_walks=3, seg_len=(5, 40), alt_len=(1, 8))
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=5)
    _bind_explicit_stream(ctx)
    _set_graph(ctx, g)
    fixed = [b"TTnTAa", b"tAGcGGCGaCngCccGaaCacGggnCACCcncTCCnNGtNngacGgNGccNNaatgCCGCTga", b"ttaGaANcNTGgtTctG"]
    for r in fixed:
        if len(r) >= k + w - 1:
            assert _probe_counts(ctx, [r])["n_emitted"] == len(oracle.sketch(r, k, w)[0]), r
    # low-entropy sequences make a minimiser come back to an earlier value often
    for trial in range(60):
        alpha = [b"ACGTN", b"AACCGTn", b"ATN", b"ACGTNacgtn"][trial % 4]
        reads = [bytes(rng.choice(list(alpha), size=int(rng.integers(k + w - 1, 700))).tolist()) for _ in range(int(rng.integers(1, 6)))]
        per = [oracle.sketch(r, k, w)[0] for r in reads]
        st = _probe_counts(ctx, reads)
        assert st["n_emitted"] == sum(len(x) for x in per), (trial, reads)
        assert st["n_distinct"] == len(np.unique(np.concatenate(per))), (trial, reads)


def test_probe_low_complexity_at_default_k_w(oracle, ctx_factory):
    """The (31, 25) instance takes window minima with v_min_f64 on the k-mer values' bit patterns: k-mers
    that start with runs of A (top bits clear: zero and denormal doubles), poly-T (reverse complement 0),
    tandem repeats (ties everywhere) must give the reference's hashes, hit flags and spectrum."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(3125)
    k, w = 31, 25

    def rseq(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    low = [b"A" * 90, b"T" * 75, b"AAAAAC" * 20, b"AT" * 60, b"A" * 40 + b"C" + b"A" * 40, b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAG" * 4,
           b"C" * 64, b"ACGT" * 30]
    g = random_graph(rng, n_sites=10, n_walks=5, seg_len=(60, 120), alt_len=(1, 8))
    # splice the low-complexity blocks into backbone vertices of the graph
    for i, blk in enumerate(low):
        v = (3 * i) % len(g.node_seq)
        g.node_seq[v] = g.node_seq[v][:len(g.node_seq[v]) // 2] + blk + g.node_seq[v][len(g.node_seq[v]) // 2:]
    reads = mosaic_reads(rng, g, n_reads=120, read_len=150, n_seg=2)
    reads += [blk + rseq(60) + blk for blk in low] + [rseq(40) + b"A" * 70 + rseq(45), b"A" * 200, b"T" * 200, b"AC" * 100]
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=10)
    _bind_explicit_stream(ctx)
    _set_graph(ctx, g)
    ctx.add_reads(reads)
    torch.cuda.synchronize()
    walk_h = np.concatenate([oracle.sketch(b"".join(g.node_seq[v] for v in path), k, w)[0] for path in g.paths])
    _, first = np.unique(walk_h, return_index=True)
    uniq = walk_h[np.sort(first)]
    per_read = [oracle.sketch(r, k, w)[0] for r in reads]
    read_h = np.unique(np.concatenate(per_read))
    st = ctx.reads_stats()
    assert st["n_emitted"] == sum(len(x) for x in per_read)
    assert st["n_distinct"] == len(read_h)
    p, n = ctx.hits_buffer()
    hit = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy()
    assert n == len(uniq)
    assert np.array_equal(hit, np.isin(uniq, read_h).astype(np.uint8))
    p, m = ctx.spectrum_export()
    sp = torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64)
    assert np.array_equal(np.sort(sp), read_h[~np.isin(read_h, uniq)])
    for h in range(len(g.paths)):
        wh, wp = ctx.walk_minimizers(h)
        eh, ep = oracle.sketch(b"".join(g.node_seq[v] for v in g.paths[h]), k, w)
        assert np.array_equal(wh, eh) and np.array_equal(wp, ep)


# --------------------------------------------------------------------------- torch / multi-GPU plumbing

def test_torch_views_of_device_buffers(oracle, ctx_factory):
    """bench.py and phi_amd/dist.py wrap the hit vector and the exported spectrum as torch tensors
    (zero copy) for the RCCL all-reduce / all-gather: check the views see the kernels' data."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(11)
    g = random_graph(rng, n_sites=8, n_walks=4, seg_len=(10, 30))
    reads = mosaic_reads(rng, g, n_reads=50, read_len=36, n_seg=2)
    k, w = 9, 4
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=10)
    _bind_explicit_stream(ctx)
    _set_graph(ctx, g)
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    d_b = torch.from_numpy(np.frombuffer(b"".join(reads), np.uint8).copy()).cuda()
    d_o = torch.from_numpy(off).cuda()
    ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), int(off[-1]))
    torch.cuda.synchronize()
    # hit vector: one byte per distinct walk minimiser, in first-occurrence order
    p, n = ctx.hits_buffer()
    hit = torch.as_tensor(pdist.DevArray(p, n), device="cuda")
    walk_h = np.concatenate([oracle.sketch(b"".join(g.node_seq[v] for v in path), k, w)[0] for path in g.paths])
    _, first = np.unique(walk_h, return_index=True)
    uniq = walk_h[np.sort(first)]
    read_h = np.unique(np.concatenate([oracle.sketch(r, k, w)[0] for r in reads]))
    assert n == len(uniq)
    assert np.array_equal(hit.cpu().numpy(), np.isin(uniq, read_h).astype(np.uint8))
    pdist.allreduce_hits(hit)                       # world size 1: a no-op that must not fail
    # exported spectrum = the distinct read hashes that are NOT walk minimisers (those are the hit
    # flags): together they are the reference's Sp_R
    p, m = ctx.spectrum_export()
    sp = torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone()
    assert np.array_equal(np.sort(sp.cpu().numpy().view(np.uint64)), read_h[~np.isin(read_h, uniq)])
    assert ctx.reads_stats()["n_distinct"] == len(read_h)
    # importing a copy of it (what another rank would send) leaves the set unchanged
    ctx.spectrum_import(sp.data_ptr(), m)
    torch.cuda.synchronize()
    assert ctx.reads_stats()["n_distinct"] == len(read_h)
    # a foreign hash that is a walk minimiser sets its flag, any other joins the set
    unseen = uniq[~np.isin(uniq, read_h)]
    assert len(unseen) > 0
    extra = torch.from_numpy(np.array([12345, 67890, unseen[0], read_h[0]], np.uint64).view(np.int64)).cuda()
    ctx.spectrum_import(extra.data_ptr(), 4)
    torch.cuda.synchronize()
    assert int(hit.cpu().numpy().sum()) == int(np.isin(uniq, read_h).sum()) + 1
    res = ctx.solve()
    assert res["spectrum_size"] == len(read_h) + 3


# --------------------------------------------------------------------------- full size (BASELINE config C2)

@pytest.mark.parametrize("seed", range(6))
def test_two_read_shards_merge_to_the_single_context_result(oracle, ctx_factory, seed):
    """The multi-GPU exchange on one GPU: two contexts score one half of the reads each, their hit vectors are
    OR-ed in place (what the RCCL all-reduce(MAX) does) and their spectrum lists exchanged
    (phi_spectrum_export / _import); both must then solve to what one context with all the reads gives."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(4200 + seed)
    k, w = int(rng.choice([5, 9, 15])), int(rng.integers(2, 7))
    g = random_graph(rng, n_sites=int(rng.integers(10, 40)), n_walks=int(rng.integers(2, 8)), seg_len=(4, 30), alt_len=(1, 8), p_del=0.2)
    reads = mosaic_reads(rng, g, n_reads=int(rng.integers(20, 120)), read_len=int(rng.integers(k + w + 4, 90)), n_seg=3, err=0.02)
    reads += [bytes(rng.choice(list(b"ACGTN"), size=200).tolist()) for _ in range(3)]      # foreign reads: spectrum-only hashes
    R, T = int(rng.choice([0, 2, 5, 100])), float(rng.choice([1.0, 0.6]))
    cut = len(reads) // 2

    def make(rs):
        c = ctx_factory(k=k, w=w, threshold=T, recombination=R)
        _bind_explicit_stream(c)
        _set_graph(c, g)
        c.add_reads(rs)
        return c
    whole, a, b = make(reads), make(reads[:cut]), make(reads[cut:])
    want = whole.solve()
    # exchange: hit vectors (in place) and spectrum lists
    views = []
    for c in (a, b):
        p, n = c.hits_buffer()
        views.append(torch.as_tensor(pdist.DevArray(p, n), device="cuda"))
    merged = torch.maximum(views[0], views[1])
    views[0].copy_(merged); views[1].copy_(merged)
    lists = []
    for c in (a, b):
        p, m = c.spectrum_export()
        lists.append(torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone() if m else torch.zeros(0, dtype=torch.int64, device="cuda"))
    if lists[1].numel(): a.spectrum_import(lists[1].data_ptr(), lists[1].numel())
    if lists[0].numel(): b.spectrum_import(lists[0].data_ptr(), lists[0].numel())
    torch.cuda.synchronize()
    for c in (a, b):
        got = c.solve()
        for key in ("objective", "upper_bound", "optimal", "spectrum_size", "filtered", "retained", "n_in_model", "n_covered", "n_switches"):
            assert got[key] == want[key], (key, got[key], want[key])
        assert np.array_equal(got["n_anchors"], want["n_anchors"])
        assert np.array_equal(got["path_vtx"], want["path_vtx"]) and np.array_equal(got["path_hap"], want["path_hap"])


def _evaluate_path_numpy(A, res, kept, cost):
    """Independent re-evaluation of the reference's objective for the returned path, from the
    kept anchors the library reports (hash, walk, first / last walk index), in numpy: the path must follow
    walks and graph edges; a minimiser counts once if one of its anchors spanning >= 2 vertices lies inside
    one segment of its own walk; every change of segment is one w-node traversal."""
    walk_off, walk_vtx, rank = A["walk_off"], A["walk_vtx"], A["top_rank"]
    pv, ph = res["path_vtx"], res["path_hap"]
    assert len(pv) > 0 and np.all(np.diff(rank[pv]) > 0)                     # topological order, no vertex twice
    # index of every path vertex on its walk (ranks increase along a walk)
    t = np.empty(len(pv), np.int64)
    for h in np.unique(ph):
        sel = np.nonzero(ph == h)[0]
        wv = walk_vtx[walk_off[h]:walk_off[h + 1]]
        pos = np.searchsorted(rank[wv], rank[pv[sel]])
        assert np.all(pos < len(wv)) and np.array_equal(wv[pos], pv[sel]), f"path leaves walk {h}"
        t[sel] = pos
    brk = np.nonzero((ph[1:] != ph[:-1]) | (t[1:] != t[:-1] + 1))[0] + 1     # a new segment starts here
    seg_lo = np.r_[0, brk]
    seg_hi = np.r_[brk, len(pv)] - 1
    # start on a walk's first vertex, end on a walk's last, jumps along graph edges
    assert t[0] == 0 and t[-1] == walk_off[ph[-1] + 1] - walk_off[ph[-1]] - 1
    for b in brk:
        u, v = int(pv[b - 1]), int(pv[b])
        assert v in A["adj"][A["adj_off"][u]:A["adj_off"][u + 1]], f"no edge {u}->{v}"
    kh, kw, k0, k1 = kept
    multi = k1 > k0
    covered = np.zeros(len(kh), bool)
    for lo, hi in zip(seg_lo, seg_hi):
        h = ph[lo]
        covered |= multi & (kw == h) & (k0 >= t[lo]) & (k1 <= t[hi])
    n_cov = len(np.unique(kh[covered]))
    return n_cov - cost * len(brk), n_cov, len(brk)


def _sorted_anchor_table(h, wk, t0, t1):
    """Kept anchors as a (n, 4) table in canonical order: a multiset comparison is then array equality."""
    o = np.lexsort((t1, t0, wk, h))
    return np.stack([h[o].astype(np.uint64), wk[o].astype(np.uint64), t0[o].astype(np.uint64), t1[o].astype(np.uint64)], axis=1)


@pytest.mark.parametrize("config", ["C2", "C3", "C4", "C5s", "C2r"])
def test_full_size_vs_oracle(oracle, ctx_factory, monkeypatch, config):
    """The CPU oracle at the sizes of BASELINE.json's configurations (its anchor and filter stages run over walks /
    spectrum ids in parallel: the whole of C2 takes it ~2 s on the GPU box's 16 threads, C5s ~15 s), on the exact
    read sets bench.py scores.  Everything stages 1-2 produce, against the reference's loops ILP_index.cpp:559-573
    (walk minimisers), :617-655 (read spectrum, anchors), :670-743 (filter, counters):
      * the (hash, position) arrays of EVERY walk, and the per-walk counts;
      * |Sp_R|, and the read hashes that are not walk minimisers as a set (the others are the hit flags);
      * filtered / retained / minimisers in the model, anchors per walk;
      * the kept anchors (hash, walk, first entry, last entry) as a multiset;
    with the reads scored by the one-chunk kernel AND by the pooled kernel (PHI_SKETCH_POOL_MIN picks; the default is
    the one-chunk kernel below 12.6 Mbases per batch), handed over whole and in three batches.  C2r = C2's model over REAL
    sequence (the CHM13 MHC contig of the reference's test data as the backbone: low complexity, tandem repeats)."""
    import torch
    from phi_amd import dist as pdist
    from phi_amd import synth
    gk, rk = synth.CONFIGS[config]
    g = synth.make_graph(**gk)
    bases, off, truth = synth.make_reads(g, **rk)
    A = g.arrays()
    st = oracle.run_stage12_arrays(A, bases, off, 31, 25, 1.0)
    all_walk_hashes = np.unique(st.m_hash)
    want_missing = st.spectrum[~np.isin(st.spectrum, all_walk_hashes)]
    want_kept = _sorted_anchor_table(st.spectrum[st.a_r], st.a_h, st.a_t0, st.a_t1)
    n = len(off) - 1
    cuts = [0, n // 3, 2 * n // 3, n]
    parts = [(bases[off[a]:off[b]], off[a:b + 1] - off[a]) for a, b in zip(cuts[:-1], cuts[1:])]
    for leg, (pool_min, batches) in enumerate((("1", [(bases, off)]), (str(1 << 40), parts))):
        monkeypatch.setenv("PHI_SKETCH_POOL_MIN", pool_min)                 # "1": every batch pooled; 2^40 chunks: never
        ctx = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        for b, o in batches:
            ctx.add_reads((b, o))
        res = ctx.solve()
        monkeypatch.delenv("PHI_SKETCH_POOL_MIN")
        assert np.array_equal(res["n_minimizers"], st.n_minimizers)
        assert res["spectrum_size"] == len(st.spectrum)
        assert (res["filtered"], res["retained"], res["n_in_model"]) == (st.filtered, st.retained, st.n_in_model)
        assert np.array_equal(res["n_anchors"], st.n_anchors)
        assert ctx.reads_stats()["n_distinct"] == len(st.spectrum)
        p, m = ctx.spectrum_export()
        got_missing = np.sort(torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64))
        assert np.array_equal(got_missing, want_missing)
        p, nu = ctx.hits_buffer()
        hits = torch.as_tensor(pdist.DevArray(p, nu), device="cuda").cpu().numpy()
        assert nu == len(all_walk_hashes) and int(hits.sum()) == len(st.spectrum) - len(want_missing)
        kh, kw, k0, k1 = ctx.kept_anchors()
        assert len(kh) == len(st.a_r)
        assert np.array_equal(_sorted_anchor_table(kh, kw, k0, k1), want_kept)
        if leg == 0:
            for h in range(g.n_walks):
                gh, gp = ctx.walk_minimizers(h)
                lo, hi = st.m_off[h], st.m_off[h + 1]
                assert np.array_equal(gh, st.m_hash[lo:hi]), f"walk {h} hashes"
                assert np.array_equal(gp, st.m_pos[lo:hi]), f"walk {h} positions"
        assert res["optimal"] == 1
        ctx.close()


@pytest.mark.parametrize("config", ["C2", "C3", "C4", "C5s", "C2r"])
def test_full_size_properties(ctx_factory, config):
    """At the sizes of BASELINE.json's configurations (synMHC-49: 49 walks x 5.2 Mbp with 1x / 10x short reads
    and 5x long noisy reads; 200 walks with 30x reads at the MHC's length) the CPU oracle checks stages 1-2
    (test_full_size_vs_oracle); no CPU solver finishes the exact solve at these sizes, and
    the domain's size-independent properties stand in for it: the read set is a SET of canonical k-mers (order,
    strand, batching and repetition of reads change nothing), the solve carries its own certificate
    (objective == proven bound), the path's objective is recounted in numpy from the kept anchors, the
    per-walk minimisers of the index equal a direct sketch of the walk's sequence, and the generator's truth
    walks come back."""
    from phi_amd import synth
    gk, rk = synth.CONFIGS[config]
    g = synth.make_graph(**gk)
    bases, off, truth = synth.make_reads(g, **rk)
    A = g.arrays()

    def solve(batches, extra=False):
        ctx = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        for b, o in batches:
            ctx.add_reads((b, o))
        res = ctx.solve()
        if extra:
            kept = ctx.kept_anchors()
            obj, n_cov, n_sw = _evaluate_path_numpy(A, res, kept, 100)
            assert (obj, n_cov, n_sw) == (res["objective"], res["n_covered"], res["n_switches"])
            assert res["n_anchors"].tolist() == np.bincount(kept[1], minlength=g.n_walks).tolist()
            assert len(np.unique(kept[0][kept[3] > kept[2]])) == res["n_in_model"]
            # two walks of the index against the stand-alone sketch of their sequences
            for h in (0, g.n_walks - 1):
                wh, wp = ctx.walk_minimizers(h)
                sh, sp, _ = ctx.sketch([g.walk_sequence(h).tobytes()], 31, 25)
                assert np.array_equal(wh, sh) and np.array_equal(wp, sp)
                assert len(wh) == res["n_minimizers"][h]
            seq = ctx.path_sequence(res["hap_len"])
            assert len(seq) == res["hap_len"] == int((A["seq_off"][res["path_vtx"] + 1] - A["seq_off"][res["path_vtx"]]).sum())
        ctx.close()
        hap = res["path_hap"]
        walks = [int(x) for x in hap[np.r_[True, hap[1:] != hap[:-1]]]]
        key = tuple(int(res[k]) for k in ("objective", "upper_bound", "optimal", "spectrum_size", "filtered", "n_in_model", "n_covered", "hap_len"))
        return key, walks, res["n_anchors"].tolist(), res["n_minimizers"].tolist()

    base = solve([(bases, off)], extra=True)
    assert base[0][2] == 1 and base[0][0] == base[0][1]              # proven optimal
    assert base[1] == truth["walks"]                                  # the mosaic is recovered
    # the same reads twice, and split in three batches
    assert solve([(bases, off), (bases, off)]) == base
    n = len(off) - 1
    cuts = [0, n // 3, 2 * n // 3, n]
    parts = [(bases[off[a]:off[b]], off[a:b + 1] - off[a]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert solve(parts) == base
    # every read reverse-complemented, read order reversed
    comp = np.zeros(256, np.uint8)
    comp[np.frombuffer(b"ACGT", np.uint8)] = np.frombuffer(b"TGCA", np.uint8)
    rc = comp[bases][::-1].copy()                                      # reverses the read order as well
    lens = np.diff(off)[::-1]
    off_rc = np.zeros(len(off), np.int64)
    np.cumsum(lens, out=off_rc[1:])
    assert solve([(rc, off_rc)]) == base


def test_two_hundred_walks_slice_vs_oracle_and_highs(oracle, ctx_factory):
    """Config 5's generator (200 walks, 30x reads) at a size the CPU checkers finish: every stage counter and
    kept anchor against the oracle, the objective against HiGHS on the reference's restated program."""
    from phi_amd import synth
    gk, rk = synth.CONFIGS["C5s"]
    g = synth.make_graph(**dict(gk, backbone_len=3_000, max_sv=200, site_spacing=150, block_len=800))
    bases, off, truth = synth.make_reads(g, **dict(rk, coverage=20.0))
    raw = bases.tobytes()
    reads = [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    seqc = g.seq_concat.tobytes()
    og = oracle.Graph(seg_names=[f"s{v + 1}" for v in range(g.n_vtx)],
                      node_seq=[seqc[g.seq_off[v]:g.seq_off[v + 1]] for v in range(g.n_vtx)],
                      adj=[g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist() for v in range(g.n_vtx)],
                      paths=[g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.n_walks)],
                      hap_names=list(g.hap_names))
    oracle.kahn(og)
    k, w, T, R = 15, 10, 1.0, 6
    ctx = ctx_factory(k=k, w=w, threshold=T, recombination=R)
    _set_graph(ctx, og)
    ctx.add_reads(reads)
    st, res, m = _check_against_oracle(oracle, ctx, og, reads, k, w, T, R)
    assert g.n_walks == 200 and res["n_in_model"] > 20
    best, _, _ = m.milp_solve(time_limit=300.0)
    assert res["objective"] == best, (res["objective"], best)


def _oracle_graph(oracle, g):
    """phi_amd.synth graph (flat arrays) -> oracle.Graph."""
    seqc = g.seq_concat.tobytes()
    og = oracle.Graph(seg_names=[f"s{v + 1}" for v in range(g.n_vtx)],
                      node_seq=[seqc[g.seq_off[v]:g.seq_off[v + 1]] for v in range(g.n_vtx)],
                      adj=[g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist() for v in range(g.n_vtx)],
                      paths=[g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.n_walks)],
                      hap_names=list(g.hap_names))
    oracle.kahn(og)
    return og


def test_native_generator_small_vs_oracle_and_highs(oracle, ctx_factory):
    """The native generator (libphi_synth.so: what makes the chromosome-scale configuration) at a size the CPU
    checkers finish: every stage against the oracle, the objective against HiGHS."""
    from phi_amd import synth
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS["C5n-tiny"]
    g = synth.NativeGraph(**gk)
    truth = g.sample(s_seed, n_mosaic)
    bases, off = g.reads(r_seed, 0, g.n_reads(cov))
    raw = bases.tobytes()
    reads = [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    og = _oracle_graph(oracle, g)
    for (k, w, R) in ((31, 25, 100), (13, 7, 8)):
        ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
        _set_graph(ctx, og)
        ctx.add_reads(reads)
        st, res, m = _check_against_oracle(oracle, ctx, og, reads, k, w, 1.0, R)
        assert res["n_in_model"] > 10
        best, _, _ = m.milp_solve(time_limit=300.0)
        assert res["objective"] == best, (k, w, R, res["objective"], best)


@pytest.mark.parametrize("config", ["C5n-mid", "C5"])
def test_chromosome_scale_properties(ctx_factory, config):
    """BASELINE.json's config 5 (synthetic 200-walk chr6-scale GFA, 30x reads) from the native generator, at a
    tenth of its size and at its stated size (170 Mbp backbone: 34 Gbases of walks, 1.2 G walk entries, 5.1 Gbases
    of reads).  Size-independent properties: the solve's certificate, the numpy recount of the path's objective
    from the kept anchors, truth walks recovered, per-walk minimisers of the de-duplicated index against a direct
    sketch of a walk's sequence, and batch-order independence (the reads in three batches, last batch first)."""
    from phi_amd import synth
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS[config]
    g = synth.NativeGraph(**gk)
    truth = g.sample(s_seed, n_mosaic)
    n = g.n_reads(cov)
    A = g.arrays()

    def solve(order, extra):
        ctx = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        cuts = [0, n // 3, 2 * n // 3, n]
        for j in order:
            b, o = g.reads(r_seed, cuts[j], cuts[j + 1])
            ctx.add_reads((b, o))
            del b, o
        res = ctx.solve()
        info = ctx.index_stats()
        if extra:
            assert info["n_entries"] == len(A["walk_vtx"]) and info["n_classes"] < info["n_entries"] // 20
            assert info["n_walk_minimizers"] == int(res["n_minimizers"].sum())
            kept = ctx.kept_anchors()
            obj, n_cov, n_sw = _evaluate_path_numpy(A, res, kept, 100)
            assert (obj, n_cov, n_sw) == (res["objective"], res["n_covered"], res["n_switches"])
            assert res["n_anchors"].tolist() == np.bincount(kept[1], minlength=g.n_walks).tolist()
            del kept
            h = g.n_walks - 1
            wh, wp = ctx.walk_minimizers(h)
            sh, sp, _ = ctx.sketch([g.walk_sequence(h).tobytes()], 31, 25)
            assert np.array_equal(wh, sh) and np.array_equal(wp, sp) and len(wh) == res["n_minimizers"][h]
        ctx.close()
        hap = res["path_hap"]
        walks = [int(x) for x in hap[np.r_[True, hap[1:] != hap[:-1]]]]
        key = tuple(int(res[k]) for k in ("objective", "upper_bound", "optimal", "spectrum_size", "filtered", "n_in_model", "n_covered", "hap_len"))
        return key, walks, res["n_anchors"].tolist()

    base = solve([0, 1, 2], True)
    assert base[0][2] == 1 and base[0][0] == base[0][1]              # proven optimal
    assert base[1] == truth["walks"]                                  # the mosaic is recovered
    assert solve([2, 0, 1], False) == base                            # the reads in another order of batches


@pytest.mark.parametrize("pooled", [False, True])
def test_novel_hash_log_spills_grows_and_starts_over(oracle, ctx_factory, monkeypatch, pooled):
    """The read kernels LOG the read hashes that are not walk minimisers (a fixed number of entries per 512-window chunk,
    1.5x what random sequence emits); what a chunk has beyond goes to the generation's overflow list, a full list makes
    phi_add_reads grow it and replay the batch, and a log past its budget is entered into the set and starts over.  The
    reference's std::map has no limit (ILP_index.cpp:622-635).  PHI_NOV_SHIFT=2 gives chunk logs of four entries, so that
    ordinary random reads (~39 novel hashes per chunk) spill; PHI_OVLIST_CAP=50 makes the list run full at once (replays:
    the list grows to twice what the batch wanted); PHI_NOVLOG_BUDGET=4096 makes the log start over with every batch.
    Reads with N take the byte-wise routine, whose novel hashes all go to the list.  |Sp_R|, the emitted count (a replay
    must not count twice), the exported list and the whole solve against the oracle; then a second generation on the
    same context."""
    import torch
    from phi_amd import dist as pdist
    rng = np.random.default_rng(606)
    g = random_graph(rng, n_sites=10, n_walks=3, seg_len=(30, 60), alt_len=(2, 8))
    k, w = 15, 10
    reads = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(60, 140))).tolist()) for _ in range(9000)]
    reads += mosaic_reads(rng, g, n_reads=40, read_len=60, n_seg=2)
    reads += [bytes(rng.choice(list(b"ACGTN"), size=300, p=[0.24, 0.24, 0.24, 0.24, 0.04]).tolist()) for _ in range(200)]
    monkeypatch.setenv("PHI_NOV_SHIFT", "2")
    monkeypatch.setenv("PHI_OVLIST_CAP", "50")
    monkeypatch.setenv("PHI_NOVLOG_BUDGET", "4096")
    if pooled:
        monkeypatch.setenv("PHI_SKETCH_POOL_MIN", "1")
        monkeypatch.setenv("PHI_SKETCH_WAVES", "7")
    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=3)
    _set_graph(ctx, g)
    ctx.add_reads(reads[:10])
    ctx.add_reads(reads[10:5000])                                # spills, fills the list: grown, replayed
    ctx.add_reads(reads[5000:])                                  # the log starts over (what it held went into the set first)
    sk = [oracle.sketch(r, k, w)[0] for r in reads]
    st = ctx.reads_stats()
    assert st["n_distinct"] == len(np.unique(np.concatenate(sk))) > 65536
    assert st["n_emitted"] == sum(len(x) for x in sk)            # the replays do not count twice
    st12, res, m = _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 3)
    walk_h = np.unique(st12.m_hash)
    p, n = ctx.spectrum_export()
    got = np.sort(torch.as_tensor(pdist.DevArray(p, n, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64))
    assert np.array_equal(got, st12.spectrum[~np.isin(st12.spectrum, walk_h)])
    # the next generation: a log and a list that start empty, a set that forgets the old one
    ctx.reset_reads()
    ctx.add_reads(reads[8000:])
    want = len(np.unique(np.concatenate(sk[8000:])))
    assert ctx.reads_stats()["n_distinct"] == want and ctx.solve()["spectrum_size"] == want


@pytest.mark.parametrize("block_steps", ["1", "3", "16", None])
def test_dp_blocks_in_parallel_equal_the_whole_chain(oracle, ctx_factory, monkeypatch, block_steps):
    """Up to 64 walks the chain of compact steps is cut at clean cuts and the blocks are solved in parallel (transfer-matrix
    rows, host chain, second pass from the true entry vectors).  PHI_DP_BLOCK_STEPS forces blocks of a few steps so that
    small graphs have many cuts; PHI_DP_NOBLOCKS=1 keeps the chain whole.  Same objective, bound, proof and counters, a
    feasible path of that value, over graphs with repeats, walks ending inside the graph, R = 0 .. 100."""
    from oracle import solve_oracle as S
    cases = []
    for seed in range(10):
        rng = np.random.default_rng(7700 + seed)
        k, w = int(rng.integers(5, 12)), int(rng.integers(1, 7))
        rep = bytes(rng.choice(list(b"ACGT"), size=k + 4).tolist()) if seed % 2 else None
        g = random_graph(rng, n_sites=int(rng.integers(30, 120)), n_walks=int(rng.integers(2, 40)), seg_len=(3, 40),
                         alt_len=(1, 10), p_del=0.25, repeat=rep)
        if seed % 3 == 0:
            g.paths[1] = g.paths[1][: len(g.paths[1]) - 3]               # ends on an interior vertex
        reads = mosaic_reads(rng, g, n_reads=300, read_len=k + w + 30, n_seg=int(rng.integers(2, 6)), err=0.01)
        R = int(rng.choice([0, 1, 3, 10, 100]))
        cases.append((g, reads, k, w, R))
    for g, reads, k, w, R in cases:
        out = {}
        for mode in ("blocks", "whole"):
            if mode == "whole":
                monkeypatch.setenv("PHI_DP_NOBLOCKS", "1")
            elif block_steps is not None:
                monkeypatch.setenv("PHI_DP_BLOCK_STEPS", block_steps)
            ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
            ctx.set_solve_budget(64)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            res = ctx.solve()
            monkeypatch.delenv("PHI_DP_NOBLOCKS", raising=False)
            monkeypatch.delenv("PHI_DP_BLOCK_STEPS", raising=False)
            out[mode] = res
        a, b = out["blocks"], out["whole"]
        for key in ("spectrum_size", "filtered", "n_in_model"):
            assert a[key] == b[key]
        if a["optimal"] and b["optimal"]:
            assert a["objective"] == b["objective"], (k, w, R, a["objective"], b["objective"])
        st = oracle.run_stage12(g, reads, k, w, 1.0)
        m = S.Model(g, st, R)
        for res in (a, b):
            obj, cov, nsw = m.objective(S.states_from_path(res["path_vtx"], res["path_hap"]))
            assert obj == res["objective"] and obj <= res["upper_bound"]


@pytest.mark.parametrize("block_steps", ["2", "5", None])
def test_dp_blocks_on_class_lanes_equal_the_whole_chain(oracle, ctx_factory, monkeypatch, block_steps):
    """More than 64 walks: the blocks' transfer rows run on CLASS lanes (walks that do the same inside a block share a
    lane), the chain runs on the device, the second pass on walk lanes.  Against the whole chain (PHI_DP_NOBLOCKS=1) on
    graphs of 65 .. 256 walks: same objective and bound, a feasible path of that value.  Graphs with few, short
    minimisers (k ~ 5, w ~ 20) have many cuts that no anchor spans: with blocks of 2 steps their classes fit 64 lanes
    (PHI_DP_STRICT turns a fallback into an error).  Dense graphs and longer blocks may exceed 64 classes and keep the
    whole chain: the other path under test."""
    from oracle import solve_oracle as S
    for seed in range(10):
        rng = np.random.default_rng(9100 + seed)
        sparse = seed >= 4                                       # few, short minimisers: many cuts that no anchor spans
        if sparse:
            k, w = int(rng.integers(4, 7)), int(rng.integers(14, 26))
        else:
            k, w = int(rng.integers(5, 12)), int(rng.integers(1, 7))
        n_walks = int(rng.choice([65, 70, 100, 128, 129, 200, 256]))
        rep = bytes(rng.choice(list(b"ACGT"), size=k + 4).tolist()) if seed % 2 else None
        g = random_graph(rng, n_sites=int(rng.integers(30, 90)), n_walks=n_walks, seg_len=(8, 40) if sparse else (3, 40), alt_len=(1, 10),
                         p_del=0.25, repeat=rep)
        if seed % 3 == 0:
            g.paths[1] = g.paths[1][: len(g.paths[1]) - 3]               # ends on an interior vertex
        reads = mosaic_reads(rng, g, n_reads=300, read_len=k + w + 30, n_seg=int(rng.integers(2, 6)), err=0.01)
        R = int(rng.choice([0, 1, 3, 10, 100]))
        out = {}
        for mode in ("blocks", "whole"):
            if mode == "whole":
                monkeypatch.setenv("PHI_DP_NOBLOCKS", "1")
            elif block_steps is not None:
                monkeypatch.setenv("PHI_DP_BLOCK_STEPS", block_steps)
                if block_steps == "2" and sparse:
                    monkeypatch.setenv("PHI_DP_STRICT", "1")
            ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
            ctx.set_solve_budget(32)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            try:
                res = ctx.solve()
            except Exception as e:
                raise AssertionError(f"seed {seed} mode {mode} walks {n_walks} k {k} w {w} R {R}: {e}")
            info = ctx.solve_stats()
            if mode == "whole":
                assert info["dp_mode"] == 1 and info["n_blocks"] == 0
            elif block_steps == "2" and sparse:
                assert info["dp_mode"] == 3 and info["n_blocks"] >= 4 and 1 <= info["max_classes"] <= 64, info
            for name in ("PHI_DP_NOBLOCKS", "PHI_DP_BLOCK_STEPS", "PHI_DP_STRICT"):
                monkeypatch.delenv(name, raising=False)
            out[mode] = res
        a, b = out["blocks"], out["whole"]
        # the chain over the blocks cut into segments (unit rows in parallel, segment chain, replay): 3 segments, and as many as
        # there are blocks -- field for field what the one-workgroup chain gives
        # ... with three unit vectors per workgroup (the default), and with one, two and four
        for n_seg, units in (("3", None), ("100000", None), ("3", "1"), ("5", "2"), ("3", "4")):
            monkeypatch.setenv("PHI_DP_CHAIN_SEGMENTS", n_seg)
            if units is not None:
                monkeypatch.setenv("PHI_DP_CHAIN_UNITS", units)
            if block_steps is not None:
                monkeypatch.setenv("PHI_DP_BLOCK_STEPS", block_steps)
            ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
            ctx.set_solve_budget(32)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            sres = ctx.solve()
            sinfo = ctx.solve_stats()
            monkeypatch.delenv("PHI_DP_CHAIN_SEGMENTS")
            monkeypatch.delenv("PHI_DP_CHAIN_UNITS", raising=False)
            monkeypatch.delenv("PHI_DP_BLOCK_STEPS", raising=False)
            for key in a:
                if isinstance(a[key], np.ndarray):
                    assert np.array_equal(a[key], sres[key]), (seed, n_seg, key)
                else:
                    assert a[key] == sres[key], (seed, n_seg, key, a[key], sres[key])
            ctx.close()
        for key in ("spectrum_size", "filtered", "n_in_model"):
            assert a[key] == b[key]
        if a["optimal"] and b["optimal"]:
            assert a["objective"] == b["objective"], (seed, n_walks, k, w, R, a["objective"], b["objective"])
        st = oracle.run_stage12(g, reads, k, w, 1.0)
        m = S.Model(g, st, R)
        for res in (a, b):
            obj, cov, nsw = m.objective(S.states_from_path(res["path_vtx"], res["path_hap"]))
            assert obj == res["objective"] and obj <= res["upper_bound"]


def test_solve_on_device_resident_anchors_equals_the_host_solve(oracle, ctx_factory, monkeypatch):
    """A model of >= 2^16 anchors that all span an edge is solved with its anchors left in HBM: per-anchor DP arrays,
    repeat set, relaxation weights and path cover counts by kernels (solve_dev.hip), the host copy fetched only when the
    branch and bound proper starts.  PHI_SOLVE_DEVICE=1 / 0 forces either mode at any size: same result field by field
    (objective, bound, proof, DP runs, path, counters, kept anchors) on instances that close at the root, that tighten
    their constant sets, and that branch (R = 0 with repeats; a budget that runs out)."""
    cases = []
    for seed in range(12):
        rng = np.random.default_rng(8800 + seed)
        k, w = int(rng.integers(5, 12)), int(rng.integers(1, 7))
        rep = bytes(rng.choice(list(b"ACGT"), size=k + 4).tolist()) if seed % 2 else None
        n_walks = int(rng.choice([3, 8, 30, 70, 150]))
        g = random_graph(rng, n_sites=int(rng.integers(25, 120)), n_walks=n_walks, seg_len=(3, 14), alt_len=(1, 8), p_del=0.25, repeat=rep)
        reads = mosaic_reads(rng, g, n_reads=200, read_len=k + w + 25, n_seg=int(rng.integers(2, 5)), err=0.01)
        cases.append((g, reads, k, w, int(rng.choice([0, 1, 2, 8, 100])), float(rng.choice([1.0, 0.6])), 200))
    rng = np.random.default_rng(31)                              # the hard instance of the budget test, cut short
    g = random_graph(rng, n_sites=400, n_walks=8, seg_len=(8, 16), alt_len=(3, 6), p_del=0.0)
    cases.append((g, mosaic_reads(rng, g, n_reads=400, read_len=60, n_seg=4, err=0.01), 9, 3, 0, 1.0, 40))
    n_branching = 0
    for g, reads, k, w, R, T, budget in cases:
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("PHI_SOLVE_DEVICE", mode)
            ctx = ctx_factory(k=k, w=w, threshold=T, recombination=R)
            ctx.set_solve_budget(budget)
            _set_graph(ctx, g)
            ctx.add_reads(reads)
            res = ctx.solve()
            out[mode] = (res, ctx.kept_anchors())
            monkeypatch.delenv("PHI_SOLVE_DEVICE")
        (a, ka), (b, kb) = out["1"], out["0"]
        for key in a:
            if isinstance(a[key], np.ndarray):
                assert np.array_equal(a[key], b[key]), key
            else:
                assert a[key] == b[key], (key, a[key], b[key])
        for x, y in zip(ka, kb):
            assert np.array_equal(x, y)
        n_branching += a["n_dp_runs"] > 8
    assert n_branching >= 1                                      # some case went past the root node


def test_solve_without_a_host_copy_of_the_walk_entries(oracle, ctx_factory, monkeypatch):
    """A chromosome-scale graph keeps no host copy of its walk entries inside the context (5.3 GB at config 5, beside the
    caller's own): the backtrack reads single entries and the decode whole stretches from the device copy, the branch and
    bound proper fetches the array when it starts.  PHI_HOST_WALKS_MAX=0 makes every graph such a graph: the same results
    field by field, on instances that close at the root, that switch walks (R = 0 .. 3), and that branch."""
    cases = []
    for seed in range(8):
        rng = np.random.default_rng(6600 + seed)
        k, w = int(rng.integers(5, 12)), int(rng.integers(1, 7))
        rep = bytes(rng.choice(list(b"ACGT"), size=k + 4).tolist()) if seed % 2 else None
        g = random_graph(rng, n_sites=int(rng.integers(25, 100)), n_walks=int(rng.choice([3, 8, 30, 70])), seg_len=(3, 14), alt_len=(1, 8), p_del=0.25, repeat=rep)
        reads = mosaic_reads(rng, g, n_reads=200, read_len=k + w + 25, n_seg=int(rng.integers(2, 5)), err=0.01)
        cases.append((g, reads, k, w, int(rng.choice([0, 1, 2, 3])), 200))
    n_branching = 0
    for g, reads, k, w, R, budget in cases:
        out = {}
        for mode in ("0", None):
            if mode is not None:
                monkeypatch.setenv("PHI_HOST_WALKS_MAX", mode)
            ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=R)
            ctx.set_solve_budget(budget)
            _set_graph(ctx, g)
            monkeypatch.delenv("PHI_HOST_WALKS_MAX", raising=False)
            ctx.add_reads(reads)
            out[mode] = (ctx.solve(), ctx.path_sequence(ctx.solve()["hap_len"]))
            ctx.close()
        (a, sa), (b, sb) = out["0"], out[None]
        for key in a:
            if isinstance(a[key], np.ndarray):
                assert np.array_equal(a[key], b[key]), key
            else:
                assert a[key] == b[key], (key, a[key], b[key])
        assert sa == sb
        n_branching += a["n_dp_runs"] > 4
    assert n_branching >= 1


def test_read_state_double_buffers_through_awkward_sequences(oracle, ctx_factory):
    """phi_reset_reads swaps the context's two sets of read buffers; the next read launch empties the set left behind.
    Sequences that stress the bookkeeping: resets with nothing in between (the half that comes to the front was never
    emptied), a one-read batch after a large generation (a handful of waves empty a large log), a large batch after a
    tiny one, batches of only empty / too-short reads, and a second phi_set_graph.  After every generation the spectrum,
    the emitted count and the solve are those of a fresh context with the same reads."""
    rng = np.random.default_rng(4242)
    k, w = 13, 7
    g = random_graph(rng, n_sites=30, n_walks=5, seg_len=(20, 60), alt_len=(1, 8))

    def rseq(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    big = mosaic_reads(rng, g, n_reads=400, read_len=120, n_seg=3, err=0.01) + [rseq(4000) for _ in range(20)]
    one = [mosaic_reads(rng, g, n_reads=1, read_len=90, n_seg=1)[0]]
    mid = mosaic_reads(rng, g, n_reads=60, read_len=100, n_seg=2) + [rseq(300) + b"N" + rseq(200)]
    plan = [("big", [big]), ("reset twice, then one read", "double"), ("one", [one]), ("big again", [big]), ("nothing but short reads", [[b"ACGT", b"", b"AC"]]),
            ("mid in three batches", [mid[:20], mid[20:40], mid[40:]]), ("triple reset", "triple"), ("mid", [mid]), ("one", [one]), ("big", [big])]

    def fresh(batches):
        c = ctx_factory(k=k, w=w, threshold=1.0, recombination=5)
        _set_graph(c, g)
        for b in batches:
            c.add_reads(b)
        return c.reads_stats(), c.solve()

    ctx = ctx_factory(k=k, w=w, threshold=1.0, recombination=5)
    _set_graph(ctx, g)
    for step, (name, batches) in enumerate(plan):
        if batches == "double":
            ctx.reset_reads(); ctx.reset_reads()
            assert ctx.reads_stats()["n_distinct"] == 0
            continue
        if batches == "triple":
            ctx.reset_reads(); ctx.reset_reads(); ctx.reset_reads()
            assert ctx.solve()["spectrum_size"] == 0
            continue
        ctx.reset_reads()
        for b in batches:
            ctx.add_reads(b)
        want_stats, want = fresh(batches)
        assert ctx.reads_stats() == want_stats, name
        got = ctx.solve()
        for key in ("objective", "spectrum_size", "filtered", "n_in_model", "n_covered"):
            assert got[key] == want[key], (name, key)
        if step == 4:
            _set_graph(ctx, g)                                   # the index again: the read state starts over
            assert ctx.reads_stats()["n_reads"] == 0
    reads = [r for b in [mid] for r in b]
    ctx.reset_reads()
    ctx.add_reads(reads)
    _check_against_oracle(oracle, ctx, g, reads, k, w, 1.0, 5)
