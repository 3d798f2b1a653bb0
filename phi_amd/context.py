"""Thin object wrapper over the C ABI: one Context == one phi_ctx (one GPU)."""
import ctypes as C

import numpy as np

from . import _capi


class PhiError(RuntimeError):
    def __init__(self, status, detail):
        self.status = status
        self.detail = detail
        super().__init__(f"{_capi.load().phi_strerror(status).decode()} ({status}): {detail}")


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class Context:
    def __init__(self, device=0):
        self._L = _capi.load()
        self._h = C.c_void_p()
        rc = self._L.phi_ctx_create(device, C.byref(self._h))
        if rc:
            raise PhiError(rc, "phi_ctx_create failed (no HIP device?)")
        self.device = device

    def close(self):
        if self._h:
            self._L.phi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise PhiError(rc, self._L.phi_last_error(self._h).decode())

    # ------------------------------------------------------------------ configuration
    def set_stream(self, hip_stream):
        """Run this context's work on the caller's HIP stream (its handle as an integer); None restores the
        context's private stream.  The null stream has handle 0 and cannot be named through the C ABI (NULL
        means "private stream" there), so 0 is refused instead of silently selecting a stream that is not
        ordered against the caller's work: create an explicit stream (torch.cuda.Stream()) and pass that."""
        if hip_stream is not None and int(hip_stream) == 0:
            raise ValueError("the null stream (handle 0) cannot be shared with a context: pass an explicit stream, or None for the private one")
        self._chk(self._L.phi_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_params(self, k=31, w=25, threshold=1.0, recombination=100, flags=_capi.PHI_FLAG_QCLP | _capi.PHI_FLAG_MIXED):
        self._chk(self._L.phi_set_params(self._h, k, w, C.c_float(threshold), recombination, flags))
        self.k, self.w = k, w

    def set_graph(self, seq_concat, seq_off, adj_off, adj, walk_off, walk_vtx, top_rank):
        """Arrays of phi_set_graph: bytes + int64/int32 numpy arrays."""
        seq_off = np.ascontiguousarray(seq_off, np.int64)
        adj_off = np.ascontiguousarray(adj_off, np.int64)
        adj = np.ascontiguousarray(adj, np.int32)
        walk_off = np.ascontiguousarray(walk_off, np.int64)
        walk_vtx = np.ascontiguousarray(walk_vtx, np.int32) if walk_vtx is not None else None     # (None: resolved on the device, phi_walk_text_resolve)
        top_rank = np.ascontiguousarray(top_rank, np.int32)
        buf = np.frombuffer(seq_concat, np.uint8) if not isinstance(seq_concat, np.ndarray) else seq_concat
        self.n_vtx, self.n_walks = len(seq_off) - 1, len(walk_off) - 1
        self._chk(self._L.phi_set_graph(self._h, self.n_vtx, _ptr(buf), _ptr(seq_off), _ptr(adj_off), _ptr(adj),
                                        self.n_walks, _ptr(walk_off), _ptr(walk_vtx), _ptr(top_rank)))

    def index_stats(self):
        """Sizes of the de-duplicated walk index (classes of walk entries with equal context) and its GPU time."""
        r = _capi.PhiIndexInfo()
        self._chk(self._L.phi_index_stats(self._h, C.byref(r)))
        return {n: getattr(r, n) for n, _ in _capi.PhiIndexInfo._fields_}

    def solve_stats(self):
        """How the last solve ran its DP: mode (0 every vertex, 1 event chain, 2 blocks on walk lanes, 3 blocks with rows
        on class lanes), blocks, class lanes per block."""
        r = _capi.PhiSolveInfo()
        self._chk(self._L.phi_solve_stats(self._h, C.byref(r)))
        return {n: getattr(r, n) for n, _ in _capi.PhiSolveInfo._fields_}

    # ------------------------------------------------------------------ reads
    def add_reads(self, seqs):
        """seqs: list of bytes, or (concat bytes/uint8 array, int64 offsets)."""
        if isinstance(seqs, tuple):
            concat, off = seqs
            concat = np.frombuffer(concat, np.uint8) if not isinstance(concat, np.ndarray) else concat
            off = np.ascontiguousarray(off, np.int64)
        else:
            off = np.zeros(len(seqs) + 1, np.int64)
            np.cumsum([len(s) for s in seqs], out=off[1:])
            concat = np.frombuffer(b"".join(seqs), np.uint8)
        self._chk(self._L.phi_add_reads(self._h, _ptr(concat), _ptr(off), len(off) - 1))

    def reads_text_begin(self, max_chunk_bytes=64 << 20):
        self._chk(self._L.phi_reads_text_begin(self._h, max_chunk_bytes))

    def add_reads_text(self, text):
        """The next bytes of a FASTA / FASTQ text; the records are found on the device.  True when the text is irregular
        (nothing more is taken: finish on the host reader with what reads_text_end hands back + the rest of the stream)."""
        buf = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else text
        irr = C.c_int32()
        self._chk(self._L.phi_add_reads_text(self._h, _ptr(buf), len(buf), C.byref(irr)))
        return bool(irr.value)

    def add_reads_text_parked(self, park, index):
        """add_reads_text with the index-th piece of a TextPark as its bytes."""
        irr = C.c_int32()
        self._chk(self._L.phi_add_reads_text_parked(self._h, park._h, index, C.byref(irr)))
        return bool(irr.value)

    def reads_text_end(self):
        """(bytes handed over but not taken, stream bytes taken as whole records)."""
        p, n, t = C.c_void_p(), C.c_int64(), C.c_int64()
        self._chk(self._L.phi_reads_text_end(self._h, C.byref(p), C.byref(n), C.byref(t)))
        return (C.string_at(p.value, n.value) if n.value else b""), t.value

    def reads_text_last_batch(self):
        """(uint8 bases, int64 offsets) of the records the device took from the last piece of text."""
        nr, nb = C.c_int64(), C.c_int64()
        self._chk(self._L.phi_reads_text_last_batch(self._h, None, 0, None, 0, C.byref(nr), C.byref(nb)))
        bases, off = np.zeros(nb.value, np.uint8), np.zeros(nr.value + 1, np.int64)
        if nr.value:
            self._chk(self._L.phi_reads_text_last_batch(self._h, _ptr(bases), nb.value, _ptr(off), nr.value, C.byref(nr), C.byref(nb)))
        return bases, off

    def reads_text_detach_carry(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._L.phi_reads_text_detach_carry(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value) if n.value else b""

    def add_reads_device(self, d_bases, d_read_off, n_reads, n_bases):
        self._chk(self._L.phi_add_reads_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_read_off), n_reads, n_bases))

    def reset_reads(self):
        self._chk(self._L.phi_reset_reads(self._h))

    def reads_stats(self):
        a, b, e, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.phi_reads_stats(self._h, C.byref(a), C.byref(b), C.byref(e), C.byref(d)))
        return dict(n_reads=a.value, n_bases=b.value, n_emitted=e.value, n_distinct=d.value)

    # ------------------------------------------------------------------ multi-GPU hooks
    def hits_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._L.phi_hits_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def spectrum_export(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._L.phi_spectrum_export(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def spectrum_import(self, d_hashes, n):
        self._chk(self._L.phi_spectrum_import(self._h, C.c_void_p(d_hashes), n))

    def spectrum_set_size(self, n):
        self._chk(self._L.phi_spectrum_set_size(self._h, n))

    # ------------------------------------------------------------------ RCCL inside the library
    @staticmethod
    def comm_unique_id():
        """128 bytes (ncclUniqueId) made by one rank; the others receive them out of band."""
        L = _capi.load()
        buf = C.create_string_buffer(_capi.PHI_COMM_ID_BYTES)
        rc = L.phi_comm_unique_id(buf, _capi.PHI_COMM_ID_BYTES)
        if rc:
            raise PhiError(rc, "phi_comm_unique_id failed (librccl missing?)")
        return buf.raw

    def comm_init(self, uid, rank, n_ranks):
        assert len(uid) == _capi.PHI_COMM_ID_BYTES
        self._chk(self._L.phi_comm_init(self._h, C.c_char_p(uid), rank, n_ranks))

    def comm_info(self):
        r, n = C.c_int32(), C.c_int32()
        self._chk(self._L.phi_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_allreduce_hits(self):
        self._chk(self._L.phi_comm_allreduce_hits(self._h))

    def comm_exchange(self):
        self._chk(self._L.phi_comm_exchange(self._h))

    # ------------------------------------------------------------------ contexts of one process exchanging through peer-mapped memory
    @staticmethod
    def peers_create(n_ranks):
        g = C.c_void_p()
        rc = _capi.load().phi_peers_create(n_ranks, C.byref(g))
        if rc:
            raise PhiError(rc, "phi_peers_create")
        return g

    @staticmethod
    def peers_destroy(group):
        _capi.load().phi_peers_destroy(group)

    def peers_join(self, group, rank):
        self._chk(self._L.phi_peers_join(self._h, group, rank))

    def peers_allreduce_hits(self):
        self._chk(self._L.phi_peers_allreduce_hits(self._h))

    def peers_exchange(self):
        self._chk(self._L.phi_peers_exchange(self._h))

    # ------------------------------------------------------------------ processes of one node exchanging through mapped hit vectors
    @staticmethod
    def ipc_unique_id():
        """128 bytes (the name of a shared-memory block) made by one rank; the others receive them out of band."""
        L = _capi.load()
        buf = C.create_string_buffer(_capi.PHI_COMM_ID_BYTES)
        rc = L.phi_ipc_unique_id(buf, _capi.PHI_COMM_ID_BYTES)
        if rc:
            raise PhiError(rc, "phi_ipc_unique_id failed")
        return buf.raw

    def ipc_init(self, uid, rank, n_ranks):
        assert len(uid) == _capi.PHI_COMM_ID_BYTES
        self._chk(self._L.phi_ipc_init(self._h, C.c_char_p(uid), rank, n_ranks))

    def ipc_info(self):
        r, n = C.c_int32(), C.c_int32()
        self._chk(self._L.phi_ipc_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def ipc_allreduce_hits(self):
        self._chk(self._L.phi_ipc_allreduce_hits(self._h))

    def ipc_flush(self):
        self._chk(self._L.phi_ipc_flush(self._h))

    def ipc_exchange(self):
        self._chk(self._L.phi_ipc_exchange(self._h))

    def ipc_check(self):
        self._chk(self._L.phi_ipc_check(self._h))

    def ipc_destroy(self):
        self._chk(self._L.phi_ipc_destroy(self._h))

    def comm_destroy(self):
        self._chk(self._L.phi_comm_destroy(self._h))

    # ------------------------------------------------------------------ solve
    def set_solve_budget(self, max_dp_runs):
        """DP runs the exact search may use (<= 0: no limit, as the reference's model.optimize())."""
        self._chk(self._L.phi_set_solve_budget(self._h, int(max_dp_runs)))

    def solve(self):
        r = _capi.PhiResult()
        self._chk(self._L.phi_solve(self._h, C.byref(r)))
        nw, npth = r.n_walks, r.n_path
        out = {f: getattr(r, f) for f in ("objective", "upper_bound", "optimal", "n_dp_runs", "n_covered", "n_path",
                                           "recombination_count", "n_switches", "hap_len", "n_walks", "spectrum_size",
                                           "filtered", "retained", "n_in_model")}
        out["path_vtx"] = np.ctypeslib.as_array(r.path_vtx, shape=(npth,)).copy() if npth else np.zeros(0, np.int32)
        out["path_hap"] = np.ctypeslib.as_array(r.path_hap, shape=(npth,)).copy() if npth else np.zeros(0, np.int32)
        out["n_minimizers"] = np.ctypeslib.as_array(r.n_minimizers, shape=(nw,)).copy()
        out["n_anchors"] = np.ctypeslib.as_array(r.n_anchors, shape=(nw,)).copy()
        return out

    def path_sequence(self, hap_len):
        buf = C.create_string_buffer(max(int(hap_len), 1))
        self._chk(self._L.phi_path_sequence(self._h, buf, hap_len))
        return buf.raw[:hap_len]

    # ------------------------------------------------------------------ introspection
    def sketch(self, seqs, k, w):
        """Stand-alone minimiser sketch: (hash[], pos[], seq[]) sorted by (seq, pos)."""
        off = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=off[1:])
        concat = np.frombuffer(b"".join(seqs), np.uint8)
        n = C.c_int64()
        self._chk(self._L.phi_sketch(self._h, _ptr(concat), _ptr(off), len(seqs), k, w, None, None, None, 0, C.byref(n)))
        h = np.zeros(n.value, np.uint64)
        p = np.zeros(n.value, np.int64)
        s = np.zeros(n.value, np.int32)
        if n.value:
            self._chk(self._L.phi_sketch(self._h, _ptr(concat), _ptr(off), len(seqs), k, w, _ptr(h), _ptr(p), _ptr(s),
                                         n.value, C.byref(n)))
        return h, p, s

    def walk_minimizers(self, walk):
        n = C.c_int64()
        self._chk(self._L.phi_walk_minimizers(self._h, walk, None, None, 0, C.byref(n)))
        h = np.zeros(n.value, np.uint64)
        p = np.zeros(n.value, np.int64)
        if n.value:
            self._chk(self._L.phi_walk_minimizers(self._h, walk, _ptr(h), _ptr(p), n.value, C.byref(n)))
        return h, p

    def walk_sharing(self, n_walks):
        """-d1 histogram: hist[c] = distinct walk minimisers occurring in exactly c walks."""
        hist = np.zeros(n_walks + 1, np.int64)
        n = C.c_int64()
        self._chk(self._L.phi_walk_sharing(self._h, _ptr(hist), n_walks + 1, C.byref(n)))
        return hist, n.value

    def kept_anchors(self):
        n = C.c_int64()
        self._chk(self._L.phi_kept_anchors(self._h, None, None, None, None, 0, C.byref(n)))
        h = np.zeros(n.value, np.uint64)
        wk = np.zeros(n.value, np.int32)
        t0 = np.zeros(n.value, np.int32)
        t1 = np.zeros(n.value, np.int32)
        if n.value:
            self._chk(self._L.phi_kept_anchors(self._h, _ptr(h), _ptr(wk), _ptr(t0), _ptr(t1), n.value, C.byref(n)))
        return h, wk, t0, t1

    def walk_entries(self):
        """(tests) host copy of the walk entries on the device."""
        n = C.c_int64()
        self._chk(self._L.phi_walk_entries(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        if n.value:
            self._chk(self._L.phi_walk_entries(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def device_synchronize(self):
        self._chk(self._L.phi_device_synchronize(self._h))

    def prof_enable(self, on=True):
        self._chk(self._L.phi_prof_enable(self._h, int(on)))

    def prof_read(self):
        n, ms, b = C.c_int64(), C.c_double(), C.c_int64()
        self._chk(self._L.phi_prof_read(self._h, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value


class TextPark:
    """Pieces of a reads text in device memory before any context wants them (include/phi_amd.h phi_text_park_*)."""

    def __init__(self, device=0):
        self._L = _capi.load()
        self._h = C.c_void_p()
        rc = self._L.phi_text_park_create(device, C.byref(self._h))
        if rc:
            raise PhiError(rc, "phi_text_park_create failed")

    def add(self, text):
        buf = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else text
        idx = C.c_int32()
        rc = self._L.phi_text_park_add(self._h, _ptr(buf), len(buf), C.byref(idx))
        if rc:
            raise PhiError(rc, "phi_text_park_add failed")
        return idx.value

    def fetch(self, index):
        n = self._L.phi_text_park_bytes(self._h, index)
        if n < 0:
            raise PhiError(-1, "no such piece")
        out = np.zeros(n, np.uint8)
        rc = self._L.phi_text_park_fetch(self._h, index, _ptr(out), n)
        if rc:
            raise PhiError(rc, "phi_text_park_fetch failed")
        return out.tobytes()

    def release(self, index):
        rc = self._L.phi_text_park_release(self._h, index)
        if rc:
            raise PhiError(rc, "phi_text_park_release failed")

    def close(self):
        if self._h:
            self._L.phi_text_park_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
