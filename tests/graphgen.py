"""Small random acyclic pangenome graphs with walks, for parity tests (test infrastructure)."""
import numpy as np

from oracle import oracle as O


def random_graph(rng, n_sites=6, n_walks=4, seg_len=(1, 12), alt_len=(1, 6), p_del=0.2, repeat=None):
    """Backbone segments separated by bi-allelic sites; every walk goes source -> sink.
    Vertices are numbered in topological order.  `repeat`: optional bytes inserted into two
    backbone segments to create minimisers with anchors at two loci."""
    def rseq(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    node_seq, adj = [], []

    def new_node(s):
        node_seq.append(s)
        adj.append([])
        return len(node_seq) - 1
    backbone, sites = [], []
    for i in range(n_sites + 1):
        s = rseq(int(rng.integers(seg_len[0], seg_len[1] + 1)))
        if repeat is not None and i in (1, n_sites - 1):
            s = s + repeat + rseq(2)
        backbone.append(new_node(s))
        if i < n_sites:
            ref = new_node(rseq(int(rng.integers(alt_len[0], alt_len[1] + 1))))
            alt = None if rng.random() < p_del else new_node(rseq(int(rng.integers(alt_len[0], alt_len[1] + 1))))
            sites.append((ref, alt))
    for i, (ref, alt) in enumerate(sites):
        a, b = backbone[i], backbone[i + 1]
        adj[a].append(ref)
        adj[ref].append(b)
        if alt is None:
            adj[a].append(b)
        else:
            adj[a].append(alt)
            adj[alt].append(b)
    paths = []
    for _ in range(n_walks):
        p = []
        for i in range(n_sites + 1):
            p.append(backbone[i])
            if i < n_sites:
                ref, alt = sites[i]
                if rng.random() < 0.5:
                    p.append(ref)
                elif alt is not None:
                    p.append(alt)
        paths.append(p)
    # renumber topologically (already: ids grow left to right except the optional alt) via Kahn
    g = O.Graph(seg_names=[f"s{i + 1}" for i in range(len(node_seq))], node_seq=node_seq,
                adj=[sorted(set(a)) for a in adj], paths=paths,
                hap_names=[f"hap{h}.{h}" for h in range(n_walks)])
    O.kahn(g)
    return g


def walk_sequence(g, h):
    return b"".join(g.node_seq[v] for v in g.paths[h])


def mosaic_reads(rng, g, n_reads=30, read_len=24, n_seg=2, err=0.0):
    """Reads from a mosaic of walks (each site region taken from one walk)."""
    L = [walk_sequence(g, h) for h in range(g.n_walks)]
    order = rng.permutation(g.n_walks)[:n_seg]
    # stitch by fractions of each walk's length (not a graph path in general; fine for sketches)
    pieces = []
    for i, h in enumerate(order):
        s = L[h]
        a, b = len(s) * i // n_seg, len(s) * (i + 1) // n_seg
        pieces.append(s[a:b])
    hap = b"".join(pieces)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(n_reads):
        if len(hap) <= read_len:
            r = hap
        else:
            a = int(rng.integers(0, len(hap) - read_len + 1))
            r = hap[a:a + read_len]
        r = bytearray(r)
        for j in range(len(r)):
            if rng.random() < err:
                r[j] = int(rng.choice(list(b"ACGT")))
        r = bytes(r)
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        reads.append(r)
    return reads
