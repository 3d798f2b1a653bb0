"""Host-side mirror of the reference's operator interface for this path: class ILP_index
(src/ILP_index.h:48-94) as src/main.cpp:114-140 drives it -- same member names, same argument
meaning, same log lines and FASTA output -- with the work done by the two native libraries:

    read_gfa / read_ip_reads   -> libphi_host.so (include/phi_host.h)
    ILP_function               -> libphi_amd.so  (include/phi_amd.h: HIP kernels on one MI355X)

Nothing here computes on the CPU; without the HIP library or a GPU the calls raise.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _capi
from .context import Context, PhiError

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libphi_host.so")

HOST_SYMBOLS = [
    "phi_gfa_read", "phi_graph_free", "phi_graph_n_vtx", "phi_graph_n_walks", "phi_graph_n_edges",
    "phi_graph_seq_concat", "phi_graph_seq_off", "phi_graph_adj_off", "phi_graph_adj", "phi_graph_walk_off",
    "phi_graph_walk_vtx", "phi_graph_topo_rank", "phi_graph_hap_name", "phi_graph_seg_name", "phi_reads_read",
    "phi_reads_free", "phi_reads_count", "phi_reads_bases", "phi_reads_off", "phi_reads_name", "phi_hap_name",
    "phi_write_fasta", "phi_reads_stream_open", "phi_reads_stream_next", "phi_reads_stream_reads",
    "phi_reads_stream_bases", "phi_reads_stream_close", "phi_reads_stream_open_blocks",
    "phi_text_stream_open", "phi_text_stream_read", "phi_text_stream_close",
    "phi_gfa_read_deferred", "phi_graph_walks_deferred", "phi_graph_walk_texts", "phi_graph_name_index", "phi_graph_resolve_walks",
    "phi_graph_set_walk_off",
]

WALK_TEXT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int32)

TEXT_BLOCK_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_void_p))

_host = None


def host_lib():
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -m phi_amd.build`")
    L = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    L.phi_gfa_read.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
    L.phi_graph_free.restype = None
    L.phi_graph_free.argtypes = [vp]
    L.phi_graph_n_vtx.restype = C.c_int32
    L.phi_graph_n_walks.restype = C.c_int32
    L.phi_graph_n_edges.restype = C.c_int64
    for n in ("n_vtx", "n_walks", "n_edges"):
        getattr(L, "phi_graph_" + n).argtypes = [vp]
    for n in ("seq_concat", "seq_off", "adj_off", "adj", "walk_off", "walk_vtx", "topo_rank"):
        f = getattr(L, "phi_graph_" + n)
        f.restype = vp
        f.argtypes = [vp]
    for n in ("hap_name", "seg_name"):
        f = getattr(L, "phi_graph_" + n)
        f.restype = C.c_char_p
        f.argtypes = [vp, C.c_int32]
    L.phi_reads_read.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
    L.phi_reads_free.restype = None
    L.phi_reads_free.argtypes = [vp]
    L.phi_reads_count.restype = C.c_int64
    L.phi_reads_count.argtypes = [vp]
    for n in ("bases", "off"):
        f = getattr(L, "phi_reads_" + n)
        f.restype = vp
        f.argtypes = [vp]
    L.phi_reads_name.restype = C.c_char_p
    L.phi_reads_name.argtypes = [vp, C.c_int64]
    L.phi_reads_stream_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
    L.phi_reads_stream_next.restype = C.c_int64
    L.phi_reads_stream_next.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_char_p, C.c_int]
    for n in ("reads", "bases"):
        f = getattr(L, "phi_reads_stream_" + n)
        f.restype = C.c_int64
        f.argtypes = [vp]
    L.phi_reads_stream_open_blocks.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, C.POINTER(vp), C.c_char_p, C.c_int]
    L.phi_text_stream_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
    L.phi_text_stream_read.restype = C.c_int64
    L.phi_text_stream_read.argtypes = [vp, vp, C.c_int64, C.c_char_p, C.c_int]
    L.phi_text_stream_close.restype = None
    L.phi_text_stream_close.argtypes = [vp]
    L.phi_reads_stream_close.restype = None
    L.phi_reads_stream_close.argtypes = [vp]
    L.phi_gfa_read_deferred.argtypes = [C.c_char_p, C.POINTER(vp), vp, vp, C.c_char_p, C.c_int]
    L.phi_graph_walks_deferred.argtypes = [vp]
    L.phi_graph_walk_texts.argtypes = [vp, vp, C.c_int32]
    L.phi_graph_name_index.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(C.c_int64)]
    L.phi_graph_resolve_walks.argtypes = [vp, C.c_char_p, C.c_int]
    L.phi_graph_set_walk_off.argtypes = [vp, vp]
    L.phi_hap_name.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    L.phi_write_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64]
    for n in HOST_SYMBOLS:
        getattr(L, n)
    _host = L
    return L


def stream_reads(path, bases_cap=64 << 20, reads_cap=1 << 20):
    """Chunks (uint8 bases, int64 offsets) of a FASTA/FASTQ file through phi_reads_stream_* -- the reader
    the command line feeds phi_add_reads with, chunk by chunk."""
    L = host_lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.phi_reads_stream_open(os.fsencode(path), C.byref(h), err, 512)
    if rc:
        raise HostError(rc, err.value.decode())
    bases = np.zeros(bases_cap, np.uint8)
    off = np.zeros(reads_cap + 1, np.int64)
    try:
        while True:
            n = L.phi_reads_stream_next(h, bases.ctypes.data, bases_cap, off.ctypes.data, reads_cap, err, 512)
            if n < 0:
                raise HostError(int(n), err.value.decode())
            if n == 0:
                break
            yield bases[:off[n]].copy(), off[:n + 1].copy()
    finally:
        L.phi_reads_stream_close(h)


def text_chunks(path, chunk_bytes=64 << 20):
    """The (inflated) text of a reads file, chunk by chunk, through phi_text_stream_*: what the command line hands to
    phi_add_reads_text."""
    L = host_lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.phi_text_stream_open(os.fsencode(path), C.byref(h), err, 512)
    if rc:
        raise HostError(rc, err.value.decode())
    buf = np.zeros(chunk_bytes, np.uint8)
    try:
        while True:
            n = L.phi_text_stream_read(h, buf.ctypes.data, chunk_bytes, err, 512)
            if n < 0:
                raise HostError(int(n), err.value.decode())
            if n == 0:
                break
            yield buf[:n].copy()
    finally:
        L.phi_text_stream_close(h)


def reads_of_text(prefix, blocks=(), bases_cap=64 << 20, reads_cap=1 << 20, stream_offset=0):
    """kseq's records (uint8 bases, int64 offsets) of the text `prefix` followed by the byte blocks `blocks`, through
    phi_reads_stream_open_blocks: the exact host state machine over text that is in memory (stream_offset: bytes of
    the stream before prefix[0])."""
    L = host_lib()
    it = iter(blocks)
    keep = []

    def nxt(_user, out):
        try:
            b = next(it)
        except StopIteration:
            return 0
        a = np.frombuffer(bytes(b), np.uint8)
        keep[:] = [a]
        out[0] = a.ctypes.data
        return len(a)
    cb = TEXT_BLOCK_FN(nxt)
    pre = np.frombuffer(bytes(prefix), np.uint8)
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.phi_reads_stream_open_blocks(pre.ctypes.data if len(pre) else None, len(pre), C.cast(cb, C.c_void_p), None, stream_offset, C.byref(h), err, 512)
    if rc:
        raise HostError(rc, err.value.decode())
    bases = np.zeros(bases_cap, np.uint8)
    off = np.zeros(reads_cap + 1, np.int64)
    out_b, out_o = [], [np.zeros(1, np.int64)]
    try:
        while True:
            n = L.phi_reads_stream_next(h, bases.ctypes.data, bases_cap, off.ctypes.data, reads_cap, err, 512)
            if n < 0:
                raise HostError(int(n), err.value.decode())
            if n == 0:
                break
            out_b.append(bases[:off[n]].copy())
            out_o.append(off[1:n + 1] + out_o[-1][-1])
    finally:
        L.phi_reads_stream_close(h)
    return (np.concatenate(out_b) if out_b else np.zeros(0, np.uint8)), np.concatenate(out_o)


def _view(ptr, n, ctype, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))


class HostError(RuntimeError):
    def __init__(self, status, detail):
        self.status = status
        super().__init__(f"host error {status}: {detail}")


def get_hap_name(gfa_name, reads_name):
    """get_hap_name (misc.cpp:58-87)."""
    buf = C.create_string_buffer(4096)
    n = host_lib().phi_hap_name(os.fsencode(gfa_name), os.fsencode(reads_name), buf, 4096)
    if n < 0:
        raise HostError(n, "name too long")
    return buf.value.decode()


class Graph:
    """gfa_t + the arrays ILP_index::read_gfa fills (adj_list, node_seq, paths, top_order_map, hap_id2name)."""

    def __init__(self, path):
        L = host_lib()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.phi_gfa_read(os.fsencode(path), C.byref(h), err, 512)
        if rc:
            raise HostError(rc, err.value.decode())
        try:
            self.n_vtx = L.phi_graph_n_vtx(h)
            self.num_walks = L.phi_graph_n_walks(h)
            ne = L.phi_graph_n_edges(h)
            self.seq_off = _view(L.phi_graph_seq_off(h), self.n_vtx + 1, C.c_int64, np.int64).copy()
            self.seq_concat = _view(L.phi_graph_seq_concat(h), int(self.seq_off[-1]), C.c_uint8, np.uint8).copy()
            self.adj_off = _view(L.phi_graph_adj_off(h), self.n_vtx + 1, C.c_int64, np.int64).copy()
            self.adj = _view(L.phi_graph_adj(h), ne, C.c_int32, np.int32).copy()
            self.walk_off = _view(L.phi_graph_walk_off(h), self.num_walks + 1, C.c_int64, np.int64).copy()
            self.walk_vtx = _view(L.phi_graph_walk_vtx(h), int(self.walk_off[-1]) if self.num_walks else 0, C.c_int32, np.int32).copy()
            self.top_order_map = _view(L.phi_graph_topo_rank(h), self.n_vtx, C.c_int32, np.int32).copy()
            self.hap_id2name = [L.phi_graph_hap_name(h, w).decode() for w in range(self.num_walks)]
            self.seg_names = [L.phi_graph_seg_name(h, v).decode() for v in range(self.n_vtx)]
        finally:
            L.phi_graph_free(h)


class DeferredGraph:
    """A GFA read with its walks left as TEXT (phi_gfa_read_deferred): resolve_on_device(ctx) sends the W-lines' walk fields
    to the context's GPU and resolves them there (include/phi_amd.h phi_walk_text_*); when the text is not of the kind
    the device takes -- or on request -- resolve_on_host() does what Graph does.  set_graph(ctx) hands the graph over."""

    def __init__(self, path, on_text=None):
        L = host_lib()
        self._L = L
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        self._cb = WALK_TEXT_FN(on_text) if on_text else None
        rc = L.phi_gfa_read_deferred(os.fsencode(path), C.byref(self._h), C.cast(self._cb, C.c_void_p) if self._cb else None, None, err, 512)
        if rc:
            raise HostError(rc, err.value.decode())
        h = self._h
        self.n_vtx = L.phi_graph_n_vtx(h)
        self.num_walks = L.phi_graph_n_walks(h)
        ne = L.phi_graph_n_edges(h)
        self.seq_off = _view(L.phi_graph_seq_off(h), self.n_vtx + 1, C.c_int64, np.int64).copy()
        self.seq_concat = _view(L.phi_graph_seq_concat(h), int(self.seq_off[-1]), C.c_uint8, np.uint8).copy()
        self.adj_off = _view(L.phi_graph_adj_off(h), self.n_vtx + 1, C.c_int64, np.int64).copy()
        self.adj = _view(L.phi_graph_adj(h), ne, C.c_int32, np.int32).copy()
        self.top_order_map = _view(L.phi_graph_topo_rank(h), self.n_vtx, C.c_int32, np.int32).copy()
        self.hap_id2name = [L.phi_graph_hap_name(h, w).decode() for w in range(self.num_walks)]
        self.walk_off = None
        self.walk_vtx = None                          # stays None when the device resolved the walks
        self.on_device = False

    def close(self):
        if self._h:
            self._L.phi_graph_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def walk_texts(self):
        """(address, bytes) of every walk field in the mapped file."""
        n = self._L.phi_graph_walk_texts(self._h, None, 0)
        buf = (C.c_int64 * (2 * max(n, 1)))()
        self._L.phi_graph_walk_texts(self._h, buf, n)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]

    def resolve_on_device(self, ctx, upload=True):
        """True when the device resolved the walks; False when the text is irregular (then: resolve_on_host)."""
        L = self._L
        prefix, pn, tbl, nn = C.c_void_p(), C.c_int32(), C.c_void_p(), C.c_int64()
        if L.phi_graph_name_index(self._h, C.byref(prefix), C.byref(pn), C.byref(tbl), C.byref(nn)) != 0:
            return False
        n = self.num_walks
        if upload:
            buf = (C.c_int64 * (2 * max(n, 1)))()
            L.phi_graph_walk_texts(self._h, buf, n)
            ctx._chk(ctx._L.phi_walk_text_upload(ctx._h, buf, n))
        woff = np.zeros(n + 1, np.int64)
        irr = C.c_uint32()
        ctx._chk(ctx._L.phi_walk_text_resolve(ctx._h, C.cast(prefix, C.c_char_p), pn.value, tbl, nn.value, self.n_vtx, woff.ctypes.data, C.byref(irr)))
        self.irregular = irr.value
        if irr.value:
            return False
        self.walk_off = woff
        L.phi_graph_set_walk_off(self._h, woff.ctypes.data)
        self.on_device = True
        return True

    def resolve_on_host(self):
        err = C.create_string_buffer(512)
        rc = self._L.phi_graph_resolve_walks(self._h, err, 512)
        if rc:
            raise HostError(rc, err.value.decode())
        self.walk_off = _view(self._L.phi_graph_walk_off(self._h), self.num_walks + 1, C.c_int64, np.int64).copy()
        self.walk_vtx = _view(self._L.phi_graph_walk_vtx(self._h), int(self.walk_off[-1]) if self.num_walks else 0, C.c_int32, np.int32).copy()

    def set_graph(self, ctx):
        ctx.set_graph(self.seq_concat, self.seq_off, self.adj_off, self.adj, self.walk_off, self.walk_vtx, self.top_order_map)


def read_reads(path):
    """(uint8 bases, int64 offsets, names) of a FASTA/FASTQ file (ILP_index::read_ip_reads)."""
    L = host_lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.phi_reads_read(os.fsencode(path), C.byref(h), err, 512)
    if rc:
        raise HostError(rc, err.value.decode())
    try:
        n = L.phi_reads_count(h)
        off = _view(L.phi_reads_off(h), n + 1, C.c_int64, np.int64).copy()
        bases = _view(L.phi_reads_bases(h), int(off[-1]), C.c_uint8, np.uint8).copy()
        names = [L.phi_reads_name(h, i).decode() for i in range(n)]
    finally:
        L.phi_reads_free(h)
    return bases, off, names


class ILP_index:
    """Mirror of class ILP_index.  Usage follows main.cpp:114-140:

        idx = ILP_index(gfa_path); idx.read_gfa()
        idx.k_mer, idx.window, idx.recombination, idx.threshold, idx.hap_file, idx.hap_name = ...
        reads = []; idx.read_ip_reads(reads, reads_path)
        idx.ILP_function(reads)
    """

    def __init__(self, g, device=0, log=sys.stderr):
        self.g = g                      # path of the GFA (the reference holds a parsed gfa_t*)
        self.device = device
        self.log = log
        # support variables with the reference's defaults (main.cpp:42-47, options.cpp:7-16)
        self.num_threads = 4
        self.hap_file = ""
        self.hap_name = ""
        self.debug = False
        self.k_mer = 31
        self.window = 25
        self.recombination = 100
        self.is_qclp = 1
        self.is_naive_exp = 0
        self.threshold = 1.0
        self.is_mixed = True
        self.max_occ = 5000
        self.graph = None
        self.result = None
        self._t0 = time.time()

    # -- ILP_index::read_gfa (ILP_index.cpp:20-155)
    def read_gfa(self):
        self.graph = Graph(self.g)
        self.n_vtx = self.graph.n_vtx
        self.num_walks = self.graph.num_walks
        self.hap_id2name = self.graph.hap_id2name
        self.top_order_map = self.graph.top_order_map

    # -- ILP_index::read_ip_reads (ILP_index.cpp:313-328): appends (name, sequence) pairs
    def read_ip_reads(self, ip_reads, ip_reads_file):
        bases, off, names = read_reads(ip_reads_file)
        raw = bases.tobytes()
        for i, nm in enumerate(names):
            ip_reads.append((nm, raw[off[i]:off[i + 1]]))

    def _stamp(self, msg):
        w = time.time() - self._t0
        cpu = time.process_time()
        print(f"[M::ILP_function::{w:.3f}*{cpu / max(w, 1e-9):.2f}] {msg}", file=self.log)

    # -- ILP_index::ILP_function (ILP_index.cpp:528-1601)
    def ILP_function(self, ip_reads):
        if self.graph is None:
            raise PhiError(_capi.PHI_ERR_STATE, "ILP_function before read_gfa")
        G = self.graph
        seqs = [r[1] if isinstance(r, tuple) else r for r in ip_reads]
        self._stamp(f"Graph has {G.n_vtx} vertices, {G.num_walks} walks and read has {len(seqs)} reads")
        ctx = Context(self.device)
        try:
            flags = (_capi.PHI_FLAG_QCLP if self.is_qclp else 0) | (_capi.PHI_FLAG_MIXED if self.is_mixed else 0)
            ctx.set_params(k=self.k_mer, w=self.window, threshold=self.threshold, recombination=self.recombination, flags=flags)
            ctx.set_graph(G.seq_concat, G.seq_off, G.adj_off, G.adj, G.walk_off, G.walk_vtx, G.top_order_map)
            ctx.add_reads(seqs)
            res = ctx.solve()
            hap = ctx.path_sequence(res["hap_len"])
            sharing = ctx.walk_sharing(G.num_walks) if self.debug else None
        finally:
            ctx.close()
        self.result = res
        print("Number of Minimizers", file=self.log)
        for h in range(G.num_walks):
            print(f"{G.hap_id2name[h]} : {int(res['n_minimizers'][h])}", file=self.log)
        if sharing is not None:                                # -d1 (ILP_index.cpp:591-604)
            hist, n_distinct = sharing
            print("Shared fraction of unique kmers by haplotypes", file=self.log)
            for i in range(1, G.num_walks + 1):
                print("[Haplotypes: %d, fraction of unique shared kmers: %.5f]" % (i, np.float32(hist[i]) / np.float32(n_distinct)), file=self.log)
        self._stamp("Haplotypes sketched")
        self._stamp(f"Indexed reads with spectrum size: {res['spectrum_size']}")
        print("Number of Anchors", file=self.log)
        for h in range(G.num_walks):
            print(f"{G.hap_id2name[h]} : {int(res['n_anchors'][h])}", file=self.log)
        sp = max(res["spectrum_size"], 1)
        self._stamp("Filtered/Retained Minimizers: %.2f/%.2f%%" % (np.float32(res["filtered"]) / np.float32(sp) * 100,
                                                                   np.float32(res["retained"]) / np.float32(sp) * 100))
        self._stamp("QP model started" if self.is_qclp else "ILP model started")
        self._stamp("%.2f%% Minimizers are in ILP" % (res["n_in_model"] * 100.0 / sp))
        self._stamp("Minimizer constraints added to the model")
        self._stamp("Using Mixed Integer Programming" if self.is_mixed else "Using Integer Programming")
        self._stamp("Optimized expanded graph constructed")
        self._stamp("Model optimized")
        print(f"Recombination count: {res['recombination_count']}", file=self.log)
        print("Recombined haplotypes: " + self.recombined_segments(res), file=self.log)
        L = host_lib()
        rc = L.phi_write_fasta(os.fsencode(self.hap_file), self.hap_name.encode(), hap, len(hap))
        if rc:
            raise HostError(rc, f"cannot write {self.hap_file}")
        self._stamp(f"Haplotype of size: {len(hap)} written to: {self.hap_file}")
        return res

    def recombined_segments(self, res):
        """The '>(name,[start,end])' list of ILP_index.cpp:1508-1550 (output coordinates)."""
        G = self.graph
        out = []
        str_id = prev = 0
        vt, hp = res["path_vtx"], res["path_hap"]
        if len(vt) == 0:
            return ""
        prev_hap = int(hp[0])
        for i in range(len(vt)):
            str_id += int(G.seq_off[vt[i] + 1] - G.seq_off[vt[i]])
            if i > 0 and int(hp[i]) != prev_hap:
                out.append(f">({G.hap_id2name[prev_hap]},[{prev},{str_id - 1}])")
                prev_hap, prev = int(hp[i]), str_id
        out.append(f">({G.hap_id2name[prev_hap]},[{prev},{str_id - 1}])")
        return "".join(out)


def main(argv=None):
    """`python -m phi_amd.ilp_index -g G -r R -o O [...]`: the reference's main.cpp flow in Python."""
    import argparse
    ap = argparse.ArgumentParser(prog="PHI")
    ap.add_argument("-g", required=True)
    ap.add_argument("-r", required=True)
    ap.add_argument("-o", required=True)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("-w", type=int, default=25)
    ap.add_argument("-R", type=int, default=100)
    ap.add_argument("-q", type=int, default=1)
    ap.add_argument("-m", type=int, default=1)
    ap.add_argument("-T", type=float, default=1.0)
    ap.add_argument("-t", type=int, default=4)
    ap.add_argument("-d", type=int, default=0)
    a = ap.parse_args(argv)
    idx = ILP_index(a.g)
    idx.read_gfa()
    idx.num_threads, idx.hap_file, idx.debug = a.t, a.o, bool(a.d)
    idx.hap_name = get_hap_name(a.g, a.r)
    idx.k_mer, idx.window, idx.recombination = a.k, a.w, a.R
    idx.is_qclp, idx.threshold, idx.is_mixed = a.q, a.T, bool(a.m)
    reads = []
    idx.read_ip_reads(reads, a.r)
    idx.ILP_function(reads)


if __name__ == "__main__":
    main()
