"""Deterministic synthetic stand-ins for the benchmark configurations of BASELINE.json.

The real 49-haplotype MHC Minigraph-Cactus graph (reference: data/preprocess.py:34-55,
data/chop_graph.sh) needs network downloads and external tools, so the benchmark uses the
generators specified in SURVEY.md section 8(d):

  synMHC-49   backbone of uniform ACGT; bi-allelic variant sites every ~250 bp (80 % SNP, 15 %
              indel 1-50 bp, 5 % SV 50-5000 bp, some SV insertions copied from elsewhere on the
              backbone to create true repeats); walks pick alleles through a block model (founder
              haplotypes per ~20 kb block, so sharing resembles LD blocks); nodes chopped to
              <= 30 bp as chop_graph.sh:62 does; every walk runs source -> sink.
  reads       drawn from a mosaic of walks, substitution / indel errors, both strands.

PRNG: numpy PCG64 seeded per SURVEY 8(d) (graph 4901, 1x reads 4902, 10x reads 4903, ...).
Output = the flat arrays of phi_set_graph (include/phi_amd.h) and (concat, offsets) reads.
"""
import numpy as np

_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b
_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _rand_seq(rng, n):
    return _ACGT[rng.integers(0, 4, size=n)]


class SynGraph:
    """Flat arrays + per-walk unit choices (kept so reads can be drawn without re-walking nodes)."""

    def __init__(self):
        self.seq_concat = None      # uint8
        self.seq_off = None         # int64 [n_vtx+1]
        self.adj_off = None
        self.adj = None
        self.walk_off = None
        self.walk_vtx = None
        self.top_rank = None
        self.n_vtx = 0
        self.n_walks = 0
        self.hap_names = []

    def arrays(self):
        return dict(seq_concat=self.seq_concat, seq_off=self.seq_off, adj_off=self.adj_off, adj=self.adj,
                    walk_off=self.walk_off, walk_vtx=self.walk_vtx, top_rank=self.top_rank)

    def walk_sequence(self, h):
        v = self.walk_vtx[self.walk_off[h]:self.walk_off[h + 1]]
        lens = (self.seq_off[v + 1] - self.seq_off[v]).astype(np.int64)
        starts = self.seq_off[v]
        idx = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(int(lens.sum()))
        return self.seq_concat[idx]


def load_backbone(path):
    """The first record of a FASTA (gz or plain) as upper-case bytes; bases outside ACGT become A (the generator's allele
    arithmetic is over ACGT)."""
    import gzip
    import os
    if not os.path.isabs(path):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
    raw = (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")).read()
    recs = raw.split(b">")
    body = recs[1].split(b"\n", 1)[1] if len(recs) > 1 else raw
    a = np.frombuffer(body.replace(b"\n", b"").replace(b"\r", b"").upper(), np.uint8).copy()
    a[~np.isin(a, _ACGT)] = ord("A")
    return a


def make_graph(backbone_len=5_000_000, n_walks=49, seed=4901, site_spacing=250, chop=30, block_len=20_000,
               n_founders=6, max_sv=5000, backbone_fasta=None):
    """backbone_fasta: a REAL sequence as the backbone (low complexity, tandem repeats, segmental duplications: what
    uniform-random bases have none of) instead of backbone_len random bases; everything else is the same model."""
    rng = np.random.default_rng(seed)
    if backbone_fasta is not None:
        backbone = load_backbone(backbone_fasta)
        backbone_len = len(backbone)
    else:
        backbone = _rand_seq(rng, backbone_len)
    # variant sites: position on the backbone, type, alleles
    n_sites = max(1, backbone_len // site_spacing)
    pos = np.sort(rng.choice(np.arange(chop + 1, backbone_len - max_sv - chop - 1), size=n_sites, replace=False))
    # keep sites apart so that a backbone segment of >= 1 base separates the ref alleles
    units = []          # list of (ref_seq or None, alt_seq or None) per site; backbone pieces between
    pieces = []         # backbone pieces (uint8 arrays), len = n_kept_sites + 1
    site_alleles = []   # (ref uint8 array possibly empty, alt uint8 array possibly empty)
    cur = 0
    kept_pos = []
    for p in pos.tolist():
        if p < cur + 1:
            continue
        u = rng.random()
        if u < 0.80:                                   # SNP
            ref = backbone[p:p + 1]
            alt = _ACGT[[(int(np.searchsorted(_ACGT, ref[0])) + int(rng.integers(1, 4))) % 4]]
            ref_len = 1
        elif u < 0.95:                                 # indel 1-50 bp
            L = int(rng.integers(1, 51))
            if rng.random() < 0.5:                     # deletion of L ref bases
                ref, alt, ref_len = backbone[p:p + L], backbone[0:0], L
            else:                                      # insertion of L new bases
                ref, alt, ref_len = backbone[0:0], _rand_seq(rng, L), 0
        else:                                          # SV 50-5000 bp
            L = int(rng.integers(50, max_sv + 1))
            r = rng.random()
            if r < 0.4:
                ref, alt, ref_len = backbone[p:p + L], backbone[0:0], L
            elif r < 0.7:
                ref, alt, ref_len = backbone[0:0], _rand_seq(rng, L), 0
            else:                                      # duplicated segment: copy from elsewhere
                q = int(rng.integers(0, backbone_len - L))
                ref, alt, ref_len = backbone[0:0], backbone[q:q + L].copy(), 0
        if p + ref_len >= backbone_len - chop:
            break
        pieces.append(backbone[cur:p])
        site_alleles.append((ref, alt))
        kept_pos.append(p)
        cur = p + ref_len
    pieces.append(backbone[cur:])
    n_sites = len(site_alleles)

    # nodes: chop every unit into <= chop bp nodes with consecutive ids; record id ranges per unit
    seqs = []
    node_count = 0
    piece_rng = np.zeros((n_sites + 1, 2), np.int64)
    allele_rng = np.zeros((n_sites, 2, 2), np.int64)     # [site][allele] -> [first, end)

    def add_unit(arr):
        nonlocal node_count
        first = node_count
        for a in range(0, len(arr), chop):
            seqs.append(arr[a:a + chop])
            node_count += 1
        return first, node_count
    for i in range(n_sites + 1):
        piece_rng[i] = add_unit(pieces[i])
        if i < n_sites:
            allele_rng[i, 0] = add_unit(site_alleles[i][0])
            allele_rng[i, 1] = add_unit(site_alleles[i][1])
    n_vtx = node_count
    lens = np.fromiter((len(s) for s in seqs), np.int64, n_vtx)
    seq_off = np.zeros(n_vtx + 1, np.int64)
    np.cumsum(lens, out=seq_off[1:])
    seq_concat = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)

    # edges: inside units consecutive ids; unit ends -> next unit starts (alleles may be empty)
    src, dst = [], []
    inner = np.ones(n_vtx, bool)
    ends = np.concatenate([piece_rng[:, 1], allele_rng[:, :, 1].ravel()])
    inner[ends[ends > 0] - 1] = False
    inner_src = np.nonzero(inner)[0]
    src.append(inner_src)
    dst.append(inner_src + 1)
    pe = piece_rng[:-1, 1] - 1                          # last node of piece i
    pn = piece_rng[1:, 0]                               # first node of piece i+1
    for al in (0, 1):
        first, end = allele_rng[:, al, 0], allele_rng[:, al, 1]
        nonempty = end > first
        src.append(pe[nonempty]); dst.append(first[nonempty])
        src.append(end[nonempty] - 1); dst.append(pn[nonempty])
        empty = ~nonempty
        src.append(pe[empty]); dst.append(pn[empty])
    src = np.concatenate(src)
    dst = np.concatenate(dst)
    e = np.unique(np.stack([src, dst], 1), axis=0)
    adj_off = np.zeros(n_vtx + 1, np.int64)
    np.cumsum(np.bincount(e[:, 0], minlength=n_vtx), out=adj_off[1:])
    adj = e[:, 1].astype(np.int32)

    # walks: founder haplotypes per block, walks follow a founder and sometimes hop at block seams
    block_id = np.asarray(kept_pos, np.int64) // block_len
    n_blocks = int(block_id.max()) + 1 if n_sites else 1
    founder_alleles = rng.random((n_founders, n_sites)) < 0.35        # alt-allele frequency per founder
    choice = np.zeros((n_walks, n_sites), np.int8)
    for h in range(n_walks):
        f = int(rng.integers(0, n_founders))
        fb = np.zeros(n_blocks, np.int64)
        for b in range(n_blocks):
            if rng.random() < 0.3:
                f = int(rng.integers(0, n_founders))
            fb[b] = f
        ch = founder_alleles[fb[block_id], np.arange(n_sites)]
        priv = rng.random(n_sites) < 0.01                             # private mutations
        choice[h] = (ch ^ priv).astype(np.int8)
    walk_list = []
    for h in range(n_walks):
        first = np.empty(2 * n_sites + 1, np.int64)
        end = np.empty(2 * n_sites + 1, np.int64)
        first[0::2], end[0::2] = piece_rng[:, 0], piece_rng[:, 1]
        first[1::2] = allele_rng[np.arange(n_sites), choice[h], 0]
        end[1::2] = allele_rng[np.arange(n_sites), choice[h], 1]
        n = end - first
        tot = int(n.sum())
        base = np.repeat(first - np.concatenate(([0], np.cumsum(n)[:-1])), n)
        walk_list.append((base + np.arange(tot)).astype(np.int32))
    walk_off = np.zeros(n_walks + 1, np.int64)
    np.cumsum([len(w) for w in walk_list], out=walk_off[1:])
    walk_vtx = np.concatenate(walk_list)

    g = SynGraph()
    g.seq_concat, g.seq_off, g.adj_off, g.adj = seq_concat, seq_off, adj_off, adj
    g.walk_off, g.walk_vtx = walk_off, walk_vtx
    g.top_rank = np.arange(n_vtx, dtype=np.int32)        # ids were issued left to right
    g.n_vtx, g.n_walks = n_vtx, n_walks
    g.hap_names = [f"syn{h:03d}.{h % 2}" for h in range(n_walks)]
    g.choice = choice
    return g


def make_reads(g, coverage=1.0, read_len=150, seed=4902, sub_err=0.005, n_mosaic=3, long_reads=False,
               indel_err=0.0, sample_seed=None):
    """Reads from a mosaic of n_mosaic walks.  Returns (uint8 concat, int64 offsets, truth dict).
    `seed` fixes the sample (which walks, where they are stitched); `sample_seed`, when given,
    draws another set of reads from the same sample (read shards of the ranks of a multi-GPU job)."""
    rng = np.random.default_rng(seed)
    hs = rng.choice(g.n_walks, size=n_mosaic, replace=False)
    seqs = [g.walk_sequence(int(h)) for h in hs]
    cuts = np.sort(rng.random(n_mosaic - 1))
    parts = []
    for i, s in enumerate(seqs):
        a = 0 if i == 0 else int(cuts[i - 1] * len(s))
        b = len(s) if i == n_mosaic - 1 else int(cuts[i] * len(s))
        parts.append(s[a:b])
    hap = np.concatenate(parts)
    L = len(hap)
    if sample_seed is not None:
        rng = np.random.default_rng(sample_seed)
    target = int(coverage * L)
    if long_reads:
        lens = np.minimum(np.maximum(rng.lognormal(np.log(8000), 0.6, size=max(1, target // 6000)).astype(np.int64), 500), L)
        lens = lens[np.cumsum(lens) <= max(target, int(lens[0]))]
    else:
        lens = np.full(max(1, (target + read_len - 1) // read_len), read_len, np.int64)
    starts = (rng.random(len(lens)) * (L - lens + 1)).astype(np.int64)
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    idx = np.repeat(starts - off[:-1], lens) + np.arange(int(off[-1]))
    bases = hap[idx].copy()
    # substitution errors
    e = rng.random(len(bases)) < sub_err
    ne = int(e.sum())
    if ne:
        cur = np.searchsorted(_ACGT, bases[e])
        bases[e] = _ACGT[(cur + rng.integers(1, 4, size=ne)) % 4]
    # reverse-complement half of the reads
    rc = rng.random(len(lens)) < 0.5
    if rc.any():
        rid = np.repeat(np.arange(len(lens)), lens)
        within = np.arange(int(off[-1])) - off[:-1][rid]
        src = np.where(rc[rid], off[:-1][rid] + lens[rid] - 1 - within, np.arange(int(off[-1])))
        out = bases[src]
        out = np.where(rc[rid], _COMP[out], out)
        bases = out.astype(np.uint8)
    if indel_err > 0:                                   # ONT-like indels: drop / duplicate bases
        keep = rng.random(len(bases)) >= indel_err / 2
        dup = rng.random(len(bases)) < indel_err / 2
        rid = np.repeat(np.arange(len(lens)), lens)
        rep = keep.astype(np.int64) + (keep & dup)
        bases = np.repeat(bases, rep)
        newlen = np.bincount(rid, weights=rep, minlength=len(lens)).astype(np.int64)
        off = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(newlen, out=off[1:])
    return bases, off, dict(walks=hs.tolist(), cuts=cuts.tolist(), hap_len=L)


def graph_from_gfa(path):
    """A GFA file (through the host reader of libphi_host.so) as a SynGraph, so that make_reads can
    draw synthetic reads from a real graph's walks (C1 with the generator's reads)."""
    from .ilp_index import Graph
    G = Graph(path)
    g = SynGraph()
    g.seq_concat = G.seq_concat
    g.seq_off, g.adj_off, g.adj = G.seq_off, G.adj_off, G.adj
    g.walk_off, g.walk_vtx, g.top_rank = G.walk_off, G.walk_vtx, G.top_order_map
    g.n_vtx, g.n_walks, g.hap_names = G.n_vtx, G.num_walks, list(G.hap_id2name)
    return g


def write_gfa(g, path):
    """GFA 1.1 with S, L (0M overlaps, forward strands) and W lines -- what the reference's reader and
    phi_gfa_read accept.  Segment names are 1-based vertex ids."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wb") as f:
        f.write(b"H\tVN:Z:1.1\n")
        so = g.seq_off
        raw = g.seq_concat.tobytes()
        for v in range(g.n_vtx):
            f.write(b"S\t%d\t%s\n" % (v + 1, raw[so[v]:so[v + 1]]))
        for u in range(g.n_vtx):
            for x in range(g.adj_off[u], g.adj_off[u + 1]):
                f.write(b"L\t%d\t+\t%d\t+\t0M\n" % (u + 1, g.adj[x] + 1))
        for h in range(g.n_walks):
            vs = g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]]
            name = g.hap_names[h] if h < len(g.hap_names) else f"hap{h}.0"
            sample, _, hap = name.rpartition(".")
            f.write(b"W\t%s\t%s\tchr\t0\t%d\t" % (sample.encode(), hap.encode() or b"0", int((so[vs + 1] - so[vs]).sum())))
            f.write(b"".join(b">%d" % (v + 1) for v in vs.tolist()))
            f.write(b"\n")


def write_reads(bases, off, path, fastq=False):
    """Reads as FASTA (or FASTQ with constant qualities), optionally gzipped by extension."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    raw = bases.tobytes()
    with op(path, "wb") as f:
        for r in range(len(off) - 1):
            s = raw[off[r]:off[r + 1]]
            if fastq:
                f.write(b"@r%d\n%s\n+\n%s\n" % (r, s, b"I" * len(s)))
            else:
                f.write(b">r%d\n%s\n" % (r, s))


CONFIGS = {
    # name: (graph kwargs, reads kwargs) -- SURVEY.md 8(d) table
    "C2": (dict(backbone_len=5_000_000, n_walks=49, seed=4901), dict(coverage=1.0, seed=4902)),
    "C3": (dict(backbone_len=5_000_000, n_walks=49, seed=4901), dict(coverage=10.0, seed=4903)),
    "C4": (dict(backbone_len=5_000_000, n_walks=49, seed=4901),
           dict(coverage=5.0, seed=4904, long_reads=True, sub_err=0.03, indel_err=0.02)),
    "C5": (dict(backbone_len=170_000_000, n_walks=200, seed=20001), dict(coverage=30.0, seed=20002)),
    # config 5 at the MHC's length: the same 200 walks and 30x reads over a 5 Mbp backbone (what one GPU box
    # generates and holds in minutes; the DP takes the dense path of > 128 walks)
    "C5s": (dict(backbone_len=5_000_000, n_walks=200, seed=20001), dict(coverage=30.0, seed=20002)),
    "C2w100": (dict(backbone_len=5_000_000, n_walks=100, seed=4901), dict(coverage=1.0, seed=4902)),
    # C2 over REAL sequence: the backbone is the CHM13 MHC contig of the reference's own test data (4.92 Mbp; the
    # reference's test/MHC-CHM13.0.fa.gz), same variant / block-coalescent model, same read generator, 49 walks, 1x reads
    "C2r": (dict(backbone_fasta="tests/golden/data/MHC-CHM13.0.fa.gz", n_walks=49, seed=4901), dict(coverage=1.0, seed=4902)),
    "C3r": (dict(backbone_fasta="tests/golden/data/MHC-CHM13.0.fa.gz", n_walks=49, seed=4901), dict(coverage=10.0, seed=4903)),
    # small variants for tests and smoke runs
    "tiny": (dict(backbone_len=60_000, n_walks=7, seed=11, max_sv=800), dict(coverage=2.0, seed=12)),
    "small": (dict(backbone_len=400_000, n_walks=16, seed=21, max_sv=2000), dict(coverage=1.0, seed=22)),
}


# ---------------------------------------------------------------------------------------------- native generator
class NativeGraph(SynGraph):
    """The same model made by libphi_synth.so (phi_amd/csrc/host/synth.cpp): counter-based generator, threads,
    chunks -- what the chromosome-scale configuration (C5: 170 Mbp x 200 walks = 1.2 G walk entries, 34 M reads)
    needs.  Its graphs are NOT those of make_graph for the same seed (another generator); the arrays are views
    of the library's memory and live as long as this object."""

    def __init__(self, backbone_len, n_walks, seed, site_spacing=250, chop=30, block_len=20_000, n_founders=6, max_sv=5000, threads=0):
        import ctypes as C
        import os
        super().__init__()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libphi_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m phi_amd.build`")
        L = C.CDLL(path)
        vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
        L.phi_syn_graph.argtypes = [i64, i32, u64, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
        L.phi_syn_free.argtypes = [vp]
        L.phi_syn_free.restype = None
        for name, rt in (("n_vtx", i32), ("n_walks", i32), ("n_sites", i64)):
            f = getattr(L, "phi_syn_" + name); f.argtypes = [vp]; f.restype = rt
        for name in ("seq", "seq_off", "adj_off", "adj", "walk_off", "walk_vtx", "topo_rank"):
            f = getattr(L, "phi_syn_" + name); f.argtypes = [vp]; f.restype = vp
        L.phi_syn_sample.argtypes = [vp, u64, i32, vp, vp]
        L.phi_syn_sample.restype = i64
        L.phi_syn_reads.argtypes = [vp, u64, i64, i64, i32, C.c_double, vp, i32]
        L.phi_syn_write_gfa.argtypes = [vp, C.c_char_p, i32]
        L.phi_syn_write_gfa.restype = i64
        L.phi_syn_write_reads.argtypes = [vp, u64, i64, i64, i32, C.c_double, C.c_char_p, i32, i32]
        L.phi_syn_write_reads.restype = i64
        self._L, self._h = L, vp()
        rc = L.phi_syn_graph(backbone_len, n_walks, seed, site_spacing, chop, block_len, n_founders, max_sv, threads, C.byref(self._h))
        if rc:
            raise RuntimeError(f"phi_syn_graph failed ({rc})")
        self.n_vtx, self.n_walks = L.phi_syn_n_vtx(self._h), L.phi_syn_n_walks(self._h)

        def view(name, ctype, n):
            return np.ctypeslib.as_array(C.cast(getattr(L, "phi_syn_" + name)(self._h), C.POINTER(ctype)), shape=(n,))
        self.seq_off = view("seq_off", C.c_int64, self.n_vtx + 1)
        self.seq_concat = view("seq", C.c_uint8, int(self.seq_off[-1]))
        self.adj_off = view("adj_off", C.c_int64, self.n_vtx + 1)
        self.adj = view("adj", C.c_int32, int(self.adj_off[-1]))
        self.walk_off = view("walk_off", C.c_int64, self.n_walks + 1)
        self.walk_vtx = view("walk_vtx", C.c_int32, int(self.walk_off[-1]))
        self.top_rank = view("topo_rank", C.c_int32, self.n_vtx)
        self.hap_names = [f"syn{h:03d}.{h % 2}" for h in range(self.n_walks)]
        self.hap_len = 0

    def close(self):
        if self._h:
            self._L.phi_syn_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sample(self, seed, n_mosaic=3):
        """Fix the sample the reads come from: a mosaic of n_mosaic walks.  Returns the truth dict of make_reads."""
        import ctypes as C
        walks = (C.c_int32 * n_mosaic)()
        cuts = (C.c_double * max(1, n_mosaic - 1))()
        self.hap_len = self._L.phi_syn_sample(self._h, seed, n_mosaic, walks, cuts)
        if self.hap_len < 0:
            raise RuntimeError("phi_syn_sample failed")
        return dict(walks=list(walks), cuts=list(cuts)[:n_mosaic - 1], hap_len=self.hap_len)

    def write_gfa(self, path, threads=0):
        """The graph as a GFA 1.1 file (names = 1-based vertex ids, the layout of write_gfa above), written natively."""
        import os
        n = self._L.phi_syn_write_gfa(self._h, os.fsencode(path), threads)
        if n < 0:
            raise RuntimeError(f"cannot write {path}")
        return n

    def write_reads(self, path, seed, r_lo, r_hi, read_len=150, sub_err=0.005, fastq=False, threads=0):
        import os
        n = self._L.phi_syn_write_reads(self._h, seed, r_lo, r_hi, read_len, sub_err, os.fsencode(path), int(fastq), threads)
        if n < 0:
            raise RuntimeError(f"cannot write {path}")
        return n

    def n_reads(self, coverage, read_len=150):
        return max(1, (int(coverage * self.hap_len) + read_len - 1) // read_len)

    def reads(self, seed, r_lo, r_hi, read_len=150, sub_err=0.005, out=None, threads=0):
        """Reads r_lo .. r_hi-1 of read set `seed` as (uint8 bases, int64 offsets); any range gives the same bytes."""
        n = r_hi - r_lo
        if out is None:
            out = np.empty(n * read_len, np.uint8)
        if self._L.phi_syn_reads(self._h, seed, r_lo, r_hi, read_len, sub_err, out.ctypes.data, threads):
            raise RuntimeError("phi_syn_reads failed")
        return out, np.arange(n + 1, dtype=np.int64) * read_len


# native configurations: (graph kwargs, sample seed, n_mosaic, read-set seed, coverage)
NATIVE_CONFIGS = {
    # BASELINE.json config 5 at its stated size: synCHR6-200, 170 Mbp backbone, 200 walks, 30x 150-bp reads
    "C5": (dict(backbone_len=170_000_000, n_walks=200, seed=20001), 20002, 3, 20003, 30.0),
    # the same generator at sizes the CPU checkers and quick GPU runs take
    "C5n-mid": (dict(backbone_len=20_000_000, n_walks=200, seed=20001), 20002, 3, 20003, 30.0),
    "C5n-small": (dict(backbone_len=400_000, n_walks=40, seed=31, max_sv=2000), 32, 3, 33, 4.0),
    "C5n-tiny": (dict(backbone_len=30_000, n_walks=12, seed=41, max_sv=500, site_spacing=120, block_len=4000), 42, 2, 43, 6.0),
}
