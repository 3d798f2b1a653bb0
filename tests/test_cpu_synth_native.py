"""CPU tests of the native generator of the synthetic configurations (libphi_synth.so, phi_amd/csrc/host/synth.cpp):
it is what makes BASELINE.json's chromosome-scale configuration; here its graphs are checked to be what
phi_set_graph accepts (forward edges, walks along edges from the source to the sink, nodes <= 30 bp) and its
counter-based draws to be independent of threads and chunking."""
import subprocess
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def synth():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "phi_amd", "csrc", "host"), "-s", os.path.join("..", "..", "libphi_synth.so")])
    from phi_amd import synth as S
    return S


def _check_graph(g):
    lens = np.diff(g.seq_off)
    assert lens.min() >= 1 and lens.max() <= 30 and g.seq_off[0] == 0 and g.seq_off[-1] == len(g.seq_concat)
    assert set(np.unique(g.seq_concat).tolist()) <= set(b"ACGT")
    src = np.repeat(np.arange(g.n_vtx), np.diff(g.adj_off))
    assert np.all(g.adj > src)                                    # ids are topological ranks: every edge goes forward
    for v in range(g.n_vtx):                                      # targets ascending, no duplicate edge
        t = g.adj[g.adj_off[v]:g.adj_off[v + 1]]
        assert np.all(np.diff(t) > 0)
    indeg = np.bincount(g.adj, minlength=g.n_vtx)
    assert np.nonzero(indeg == 0)[0].tolist() == [0] and np.nonzero(np.diff(g.adj_off) == 0)[0].tolist() == [g.n_vtx - 1]
    edges = set(zip(src.tolist(), g.adj.tolist()))
    for h in range(g.n_walks):
        w = g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]]
        assert w[0] == 0 and w[-1] == g.n_vtx - 1
        assert all((int(a), int(b)) in edges for a, b in zip(w[:-1], w[1:]))


def test_native_graph_is_a_valid_pangenome(synth):
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS["C5n-tiny"]
    g = synth.NativeGraph(**gk)
    _check_graph(g)
    assert g.n_walks == 12 and len({g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tobytes() for h in range(g.n_walks)}) > 6
    # the same arrays whatever the thread count
    g1 = synth.NativeGraph(**dict(gk, threads=1))
    g3 = synth.NativeGraph(**dict(gk, threads=3))
    for name in ("seq_concat", "seq_off", "adj_off", "adj", "walk_off", "walk_vtx"):
        assert np.array_equal(getattr(g1, name), getattr(g, name)) and np.array_equal(getattr(g3, name), getattr(g, name)), name
    # another seed, another graph
    g2 = synth.NativeGraph(**dict(gk, seed=gk["seed"] + 1))
    assert not np.array_equal(g2.seq_concat[:1000], g.seq_concat[:1000])


def test_native_reads_are_chunk_and_thread_independent(synth):
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS["C5n-small"]
    g = synth.NativeGraph(**gk)
    _check_graph(g)
    truth = g.sample(s_seed, n_mosaic)
    assert len(set(truth["walks"])) == n_mosaic and truth["hap_len"] > 300_000
    n = g.n_reads(cov)
    b, off = g.reads(r_seed, 0, n)
    assert len(b) == n * 150 and off[-1] == n * 150 and set(np.unique(b).tolist()) <= set(b"ACGT")
    parts = [g.reads(r_seed, lo, min(n, lo + 777), threads=t)[0] for lo, t in zip(range(0, n, 777), [1, 2, 5] * n)]
    assert np.array_equal(np.concatenate(parts), b)
    # reads are pieces of the sample: most reads of the forward strand are exact substrings of a walk or differ by a few bases
    seqs = [g.walk_sequence(h).tobytes() for h in truth["walks"]]
    raw = b.tobytes()
    exact = sum(any(raw[i * 150:(i + 1) * 150] in s for s in seqs) for i in range(200))
    assert 20 < exact < 160                                        # ~half are reverse-complemented, ~half of the rest carry an error
    other = g.reads(r_seed + 1, 0, 50)[0]
    assert not np.array_equal(other, b[:len(other)])
