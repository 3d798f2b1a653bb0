"""oracle.py -- Python face of the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker.  The product (phi_amd/) never imports it.

It restates, in this repo's own words, the parts of the reference that feed the hot path
(paths relative to /root/reference):

  GFA S/L/W parsing            src/gfa-io.cpp:214-432, 462-508 ; walk flip :64-115
  arc completion               src/gfa-base.cpp:269-304, 421-430
  graph flattening             src/ILP_index.cpp:20-155   (read_gfa)
  read loading                 src/ILP_index.cpp:313-328  (kseq FASTA/FASTQ)
  output record name           src/misc.cpp:58-87         (get_hap_name)
  stages 1-2                   oracle/phi_oracle.c        (C, called through ctypes)
  model / objective            oracle/solve_oracle.py
"""
import ctypes as C
import gzip
import os
from collections import deque
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        L = _load("liboracle.so")
        L.orc_hash128_to_64.restype = C.c_uint64
        L.orc_hash128_to_64.argtypes = [C.c_char_p, C.c_int]
        L.orc_sketch.restype = C.c_int64
        L.orc_sketch.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_run.restype = C.c_void_p
        L.orc_run.argtypes = [C.c_int32, C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_stage_seconds.restype = C.c_double
        L.orc_stage_seconds.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_max_threads.restype = C.c_int
        L.orc_sketch_reads.restype = C.c_int64
        L.orc_sketch_reads.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        for name in ("spectrum_size", "filtered", "retained", "n_in_model", "n_kept"):
            f = getattr(L, "orc_" + name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p]
        for name in ("spectrum", "n_minimizers", "n_anchors", "a_r", "a_h", "a_t0", "a_t1", "a_pos",
                     "m_off", "m_hash", "m_pos"):
            f = getattr(L, "orc_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libphi_ref.so"))


def ref():
    """The reference's own Gurobi-free TUs (oracle/_ref/libphi_ref.so)."""
    global _ref
    if _ref is None:
        L = _load(os.path.join("_ref", "libphi_ref.so"))
        L.ref_hash128_to_64.restype = C.c_uint64
        L.ref_hash128_to_64.argtypes = [C.c_char_p, C.c_int]
        L.ref_gfa_read.restype = C.c_void_p
        L.ref_gfa_read.argtypes = [C.c_char_p]
        for name in ("n_seg", "n_walk"):
            f = getattr(L, "ref_gfa_" + name)
            f.restype = C.c_uint32
            f.argtypes = [C.c_void_p]
        for name in ("seg_name", "seg_seq", "walk_sample"):
            f = getattr(L, "ref_gfa_" + name)
            f.restype = C.c_char_p
            f.argtypes = [C.c_void_p, C.c_uint32]
        for name in ("seg_len", "walk_hap", "walk_n_v"):
            f = getattr(L, "ref_gfa_" + name)
            f.restype = C.c_int32
            f.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_gfa_arc_n.restype = C.c_uint32
        L.ref_gfa_arc_n.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_gfa_arc_w.restype = C.c_uint32
        L.ref_gfa_arc_w.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ref_gfa_walk_v.restype = C.POINTER(C.c_uint32)
        L.ref_gfa_walk_v.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_get_hap_name.restype = C.c_int
        L.ref_get_hap_name.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        _ref = L
    return _ref


# ---------------------------------------------------------------------------- hash / sketch

def hash128_to_64(b: bytes) -> int:
    return int(lib().orc_hash128_to_64(b, len(b)))


def sketch(seq: bytes, k: int, w: int):
    """(hash[], pos[]) of one sequence: compute_hashes / index_kmers minus the vertex map."""
    L = lib()
    n = L.orc_sketch(seq, len(seq), k, w, None, None, 0)
    h = np.zeros(n, np.uint64)
    p = np.zeros(n, np.int64)
    if n:
        L.orc_sketch(seq, len(seq), k, w, h.ctypes.data, p.ctypes.data, n)
    return h, p


# ---------------------------------------------------------------------------- graph

@dataclass
class Graph:
    """Flattened forward-strand graph: the arrays ILP_index::read_gfa leaves behind."""
    seg_names: list
    node_seq: list                 # bytes per vertex, original case
    adj: list                      # list[list[int]] forward adjacency (targets, orientation dropped)
    paths: list                    # list[list[int]] per walk
    hap_names: list                # sample + "." + hap
    top_order: list = field(default_factory=list)
    top_rank: list = field(default_factory=list)

    @property
    def n_vtx(self):
        return len(self.node_seq)

    @property
    def n_walks(self):
        return len(self.paths)

    def arrays(self):
        """The flat arrays of the C ABI (include/phi_amd.h phi_set_graph)."""
        seq_off = np.zeros(self.n_vtx + 1, np.int64)
        np.cumsum([len(s) for s in self.node_seq], out=seq_off[1:])
        seq_concat = b"".join(self.node_seq)
        adj_off = np.zeros(self.n_vtx + 1, np.int64)
        np.cumsum([len(a) for a in self.adj], out=adj_off[1:])
        adj = np.fromiter((x for a in self.adj for x in a), np.int32, int(adj_off[-1]))
        walk_off = np.zeros(self.n_walks + 1, np.int64)
        np.cumsum([len(p) for p in self.paths], out=walk_off[1:])
        walk_vtx = np.fromiter((x for p in self.paths for x in p), np.int32, int(walk_off[-1]))
        top_rank = np.asarray(self.top_rank, np.int32)
        return dict(seq_concat=seq_concat, seq_off=seq_off, adj_off=adj_off, adj=adj,
                    walk_off=walk_off, walk_vtx=walk_vtx, top_rank=top_rank)


def _open_text(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


class WalkError(Exception):
    """read_gfa exit(1): a walk holds a reverse-strand vertex (ILP_index.cpp:104-107)."""


def parse_gfa(path) -> Graph:
    """gfa_read + ILP_index::read_gfa, restated.

    Segment ids are first-seen order over S- and L-lines (gfa-base.cpp:75-96); W-lines resolve
    names against the segments seen so far (gfa-io.cpp:402-409); every arc gets its complement
    (gfa-base.cpp:269-304); only arcs leaving forward vertices survive flattening and the
    target's orientation is dropped (ILP_index.cpp:72-84).  Duplicate L-lines are merged.
    """
    name2id = {}
    names, seqs = [], []
    arcs = []           # (v, w) oriented vertex ids
    walks = []          # (sample, hap, [oriented v])

    def add_seg(name):
        i = name2id.get(name)
        if i is None:
            i = len(names)
            name2id[name] = i
            names.append(name)
            seqs.append(None)
        return i

    with _open_text(path) as f:
        for raw in f:
            line = raw.rstrip(b"\r\n")
            if len(line) < 3 or line[1:2] != b"\t":
                continue
            t = line[0:1]
            fld = line.split(b"\t")
            if t == b"S" and len(fld) >= 3:
                sid = add_seg(fld[1])
                seqs[sid] = None if fld[2][:1] == b"*" else fld[2]
            elif t == b"L" and len(fld) >= 5:
                if fld[2] not in (b"+", b"-") or fld[4] not in (b"+", b"-"):
                    continue
                v = add_seg(fld[1]) << 1 | (fld[2] != b"+")
                w = add_seg(fld[3]) << 1 | (fld[4] != b"+")
                arcs.append((v, w))
            elif t == b"W" and len(fld) >= 7:
                vs = []
                s = fld[6]
                i = 0
                while i < len(s):
                    if s[i:i + 1] in (b">", b"<"):
                        j = i + 1
                        while j < len(s) and s[j:j + 1] not in (b">", b"<"):
                            j += 1
                        sid = name2id.get(s[i + 1:j], -1)
                        if sid >= 0:
                            vs.append(sid << 1 | (s[i:i + 1] == b"<"))
                        i = j
                    else:
                        i += 1
                try:
                    hap = int(fld[2])
                except ValueError:
                    hap = 0
                walks.append((fld[1], hap, vs))

    n_seg = len(names)
    # gfa_walk_flip: first-seen strand per segment, flip walks that mostly disagree
    strand = [0] * n_seg
    for _, _, vs in walks:
        for v in vs:
            if strand[v >> 1] == 0:
                strand[v >> 1] = -1 if v & 1 else 1
    for wi, (smp, hap, vs) in enumerate(walks):
        agree = sum(1 for v in vs if (-1 if v & 1 else 1) == strand[v >> 1])
        if agree < len(vs) - agree:
            walks[wi] = (smp, hap, [v ^ 1 for v in reversed(vs)])

    # arcs + complements, segments without a sequence are deleted (gfa_fix_no_seg)
    deleted = [s is None or len(s) == 0 for s in seqs]
    arcset = set()
    for v, w in arcs:
        if deleted[v >> 1] or deleted[w >> 1]:
            continue
        arcset.add((v, w))
        arcset.add((w ^ 1, v ^ 1))
    adj = [[] for _ in range(n_seg)]
    for v, w in sorted(arcset):
        if v & 1 == 0:
            adj[v >> 1].append(w >> 1)

    paths, hap_names = [], []
    for wi, (smp, hap, vs) in enumerate(walks):
        for v in vs:
            if v & 1:
                raise WalkError(f"Walk {wi} has reverse strand vertices {v}")
        paths.append([v >> 1 for v in vs])
        hap_names.append(smp.decode() + "." + str(hap))

    g = Graph(seg_names=[n.decode() for n in names],
              node_seq=[s if s is not None else b"" for s in seqs],
              adj=adj, paths=paths, hap_names=hap_names)
    kahn(g)
    return g


def kahn(g: Graph):
    """Kahn's algorithm with a FIFO queue (ILP_index.cpp:115-154)."""
    indeg = [0] * g.n_vtx
    for a in g.adj:
        for v in a:
            indeg[v] += 1
    q = deque(i for i in range(g.n_vtx) if indeg[i] == 0)
    order = []
    while q:
        u = q.popleft()
        order.append(u)
        for v in g.adj[u]:
            indeg[v] -= 1
            if indeg[v] == 0:
                q.append(v)
    rank = [0] * g.n_vtx
    for i, u in enumerate(order):
        rank[u] = i
    g.top_order, g.top_rank = order, rank


def ref_parse_gfa(path) -> Graph:
    """Same flattening, but segments/arcs/walks come from the reference's own gfa_read()."""
    R = ref()
    g = R.ref_gfa_read(os.fsencode(path))
    if not g:
        raise RuntimeError("reference gfa_read failed")
    n_seg = R.ref_gfa_n_seg(g)
    names = [R.ref_gfa_seg_name(g, s).decode() for s in range(n_seg)]
    seqs = [(R.ref_gfa_seg_seq(g, s) or b"") for s in range(n_seg)]
    adj = [[] for _ in range(n_seg)]
    for s in range(n_seg):
        v = s << 1
        for i in range(R.ref_gfa_arc_n(g, v)):
            adj[s].append(R.ref_gfa_arc_w(g, v, i) >> 1)
    paths, hap_names = [], []
    for w in range(R.ref_gfa_n_walk(g)):
        n = R.ref_gfa_walk_n_v(g, w)
        vs = R.ref_gfa_walk_v(g, w)
        vv = [vs[i] for i in range(n)]
        for v in vv:
            if v & 1:
                raise WalkError(f"Walk {w} has reverse strand vertices {v}")
        paths.append([v >> 1 for v in vv])
        hap_names.append(R.ref_gfa_walk_sample(g, w).decode() + "." + str(R.ref_gfa_walk_hap(g, w)))
    G = Graph(seg_names=names, node_seq=seqs, adj=adj, paths=paths, hap_names=hap_names)
    kahn(G)
    return G


def ref_read_reads(path):
    """(name, sequence) of every record as the reference's own kseq.h reads them (oracle/_ref)."""
    R = ref()
    sz = max(os.path.getsize(path) * 12 + 4096, 1 << 16)            # gzip expands
    names = C.create_string_buffer(sz)
    seqs = C.create_string_buffer(sz)
    off = (C.c_int64 * (sz // 2 + 2))()
    R.ref_read_reads.restype = C.c_int64
    R.ref_read_reads.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.c_int64]
    n = R.ref_read_reads(os.fsencode(path), names, sz, seqs, sz, off, sz // 2 + 2)
    if n < 0:
        raise RuntimeError(f"reference kseq reader failed ({n})")
    nm = names.raw.split(b"\0", n)[:n]
    return [(nm[i], seqs.raw[off[i]:off[i + 1]]) for i in range(n)]


def ref_hap_name(gfa_path: str, reads_path: str) -> str:
    buf = C.create_string_buffer(4096)
    n = ref().ref_get_hap_name(gfa_path.encode(), reads_path.encode(), buf, 4096)
    assert n >= 0
    return buf.value.decode()


def hap_name(gfa_path: str, reads_path: str) -> str:
    """get_hap_name (misc.cpp:58-87)."""
    def base(p):
        i = max(p.rfind("/"), p.rfind("\\"))
        return p[i + 1:] if i >= 0 else p
    name = base(gfa_path)
    i = name.rfind(".")
    if i >= 0:
        name = name[:i]
    name += "_" + base(reads_path)
    i = name.rfind(".")
    if i >= 0:
        name = name[:i]
    return name


# ---------------------------------------------------------------------------- reads

def read_reads(path):
    """The reference's FASTA/FASTQ reading (ILP_index.cpp:313-328 over kseq.h:192-233) restated, malformed input
    included: list of (name, sequence).  The next header is the next '>' or '@' byte wherever it stands; the name
    ends at the first white space; sequence lines are kept as they are (a trailing CR goes when the string is
    longer than one byte, kseq.h:146) and end at a line that starts with '>', '@' or '+'; empty lines are skipped;
    after '+', whole lines are read as quality until they cover the sequence; a quality string of another length,
    or none, ends the reading (kseq_read returns -2 and the caller's loop stops)."""
    with _open_text(path) as f:
        data = f.read()
    n = len(data)
    pos = 0
    out = []
    last_char = 0
    # kseq reads blocks of 65 536 bytes and flags the end when a block comes back short (kseq.h:81,113,242): when every byte
    # is consumed the flag is up already, unless the length is a multiple of the block size -- then only the next read, of
    # 0 bytes, raises it (zero_read)
    zero_read = False

    def kseq_eof():
        return pos >= n and (zero_read or n % 65536 != 0)

    def line_into(buf: bytearray):
        nonlocal pos, zero_read
        if kseq_eof():
            return -1
        if pos >= n:
            zero_read = True
        nl = data.find(b"\n", pos)
        end = nl if nl >= 0 else n
        buf += data[pos:end]
        pos = end + 1 if nl >= 0 else n
        if len(buf) > 1 and buf[-1] == 13:
            del buf[-1]
        return len(buf)

    while True:
        if last_char == 0:
            while pos < n and data[pos] not in (62, 64):        # '>' '@'
                pos += 1
            if pos >= n:
                break
            last_char = data[pos]
            pos += 1
        # name: up to the first white space
        if kseq_eof():
            break
        if pos >= n:
            zero_read = True
        i = pos
        while i < n and data[i] not in b" \t\n\v\f\r":
            i += 1
        name = data[pos:i]
        delim = data[i] if i < n else 0
        pos = i + 1 if i < n else n
        if delim != 10:
            line_into(bytearray())                              # the comment
        seq = bytearray()
        c = -1
        while True:
            if pos >= n:
                c = -1
                zero_read = True
                break
            c = data[pos]
            pos += 1
            if c in (62, 43, 64):                               # '>' '+' '@'
                break
            if c == 10:
                continue
            seq.append(c)
            line_into(seq)
        if c in (62, 64):
            last_char = c
        if c != 43:
            out.append((name, bytes(seq)))                       # FASTA (at the end of the file: the next turn stops)
            continue
        # FASTQ: the rest of the '+' line, then whole quality lines
        nl = data.find(b"\n", pos)
        if nl < 0:
            break                                               # -2: no quality string
        pos = nl + 1
        qual = bytearray()
        while line_into(qual) >= 0 and len(qual) < len(seq):
            pass
        last_char = 0
        if len(qual) != len(seq):
            break                                               # -2: quality of another length
        out.append((name, bytes(seq)))
    return out


# ---------------------------------------------------------------------------- stages 1-2

@dataclass
class Stage12:
    n_minimizers: np.ndarray
    n_anchors: np.ndarray
    spectrum: np.ndarray
    filtered: int
    retained: int
    n_in_model: int
    a_r: np.ndarray
    a_h: np.ndarray
    a_t0: np.ndarray
    a_t1: np.ndarray
    a_pos: np.ndarray
    m_off: np.ndarray
    m_hash: np.ndarray
    m_pos: np.ndarray


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    ct = {np.int32: C.c_int32, np.int64: C.c_int64, np.uint64: C.c_uint64}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()


def run_stage12(g: Graph, reads, k=31, w=25, threshold=1.0) -> Stage12:
    """reads: list of sequences (bytes) or of (name, seq)."""
    L = lib()
    A = g.arrays()
    seqs = [r[1] if isinstance(r, tuple) else r for r in reads]
    read_off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=read_off[1:])
    reads_concat = b"".join(seqs)
    h = L.orc_run(g.n_vtx, A["seq_concat"], A["seq_off"].ctypes.data, g.n_walks,
                  A["walk_off"].ctypes.data, A["walk_vtx"].ctypes.data,
                  reads_concat, read_off.ctypes.data, len(seqs), k, w, C.c_float(threshold))
    try:
        nk = L.orc_n_kept(h)
        nw = g.n_walks
        m_off = _arr(L.orc_m_off(h), nw + 1, np.int64)
        nm = int(m_off[-1]) if nw else 0
        return Stage12(
            n_minimizers=_arr(L.orc_n_minimizers(h), nw, np.int64),
            n_anchors=_arr(L.orc_n_anchors(h), nw, np.int64),
            spectrum=_arr(L.orc_spectrum(h), L.orc_spectrum_size(h), np.uint64),
            filtered=L.orc_filtered(h), retained=L.orc_retained(h), n_in_model=L.orc_n_in_model(h),
            a_r=_arr(L.orc_a_r(h), nk, np.int32), a_h=_arr(L.orc_a_h(h), nk, np.int32),
            a_t0=_arr(L.orc_a_t0(h), nk, np.int32), a_t1=_arr(L.orc_a_t1(h), nk, np.int32),
            a_pos=_arr(L.orc_a_pos(h), nk, np.int64),
            m_off=m_off, m_hash=_arr(L.orc_m_hash(h), nm, np.uint64), m_pos=_arr(L.orc_m_pos(h), nm, np.int64))
    finally:
        L.orc_free(h)


def run_stage12_arrays(A, bases, read_off, k=31, w=25, threshold=1.0, threads=None, want_minimizers=True) -> Stage12:
    """Stages 1-2 on the flat arrays of the C ABI (a generator's graph: no per-vertex Python objects), reads as
    (uint8 concat, int64 offsets): what the full-size parity tests and bench.py's parity check call.  The result also carries
    `stage_s`: wall seconds of the walk sketch, read sketch + spectrum, anchors, filter."""
    L = lib()
    if threads:
        L.orc_set_threads(int(threads))
    seq_off = np.ascontiguousarray(A["seq_off"], np.int64)
    walk_off = np.ascontiguousarray(A["walk_off"], np.int64)
    walk_vtx = np.ascontiguousarray(A["walk_vtx"], np.int32)
    read_off = np.ascontiguousarray(read_off, np.int64)
    seq = A["seq_concat"]
    seq = seq.tobytes() if isinstance(seq, np.ndarray) else bytes(seq)
    raw = bases.tobytes() if isinstance(bases, np.ndarray) else bytes(bases)
    nw = len(walk_off) - 1
    h = L.orc_run(len(seq_off) - 1, seq, seq_off.ctypes.data, nw, walk_off.ctypes.data, walk_vtx.ctypes.data,
                  raw, read_off.ctypes.data, len(read_off) - 1, k, w, C.c_float(threshold))
    try:
        nk = L.orc_n_kept(h)
        m_off = _arr(L.orc_m_off(h), nw + 1, np.int64)
        nm = int(m_off[-1]) if (nw and want_minimizers) else 0
        st = Stage12(
            n_minimizers=_arr(L.orc_n_minimizers(h), nw, np.int64),
            n_anchors=_arr(L.orc_n_anchors(h), nw, np.int64),
            spectrum=_arr(L.orc_spectrum(h), L.orc_spectrum_size(h), np.uint64),
            filtered=L.orc_filtered(h), retained=L.orc_retained(h), n_in_model=L.orc_n_in_model(h),
            a_r=_arr(L.orc_a_r(h), nk, np.int32), a_h=_arr(L.orc_a_h(h), nk, np.int32),
            a_t0=_arr(L.orc_a_t0(h), nk, np.int32), a_t1=_arr(L.orc_a_t1(h), nk, np.int32),
            a_pos=_arr(L.orc_a_pos(h), nk, np.int64),
            m_off=m_off, m_hash=_arr(L.orc_m_hash(h), nm, np.uint64), m_pos=_arr(L.orc_m_pos(h), nm, np.int64))
        st.stage_s = [L.orc_stage_seconds(h, i) for i in range(4)]
        return st
    finally:
        L.orc_free(h)
