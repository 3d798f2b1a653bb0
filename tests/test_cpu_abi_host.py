"""CPU tests: the C-ABI libraries load and export what the headers declare (no GPU compute), the
host-side readers agree with the reference parser's goldens and with the oracle, and the device
k-mer / hash primitives (phi_dev.h compiled for the host) agree with the known answers."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(phi_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    from phi_amd import build as B
    B.build_device()
    B.build_host()
    return True


def test_phi_amd_abi_exports_every_declared_symbol(built):
    from phi_amd import _capi
    names = _declared("phi_amd.h")
    assert len(names) >= 20
    assert sorted(_capi.SYMBOLS) == names            # the binding covers the header exactly
    L = C.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    L2 = _capi.load()
    assert L2.phi_strerror(0) == b"ok" and L2.phi_strerror(-6) == b"walk does not follow the graph"


def test_phi_amd_fails_loudly_without_gpu(built):
    """No CPU fallback: without a HIP device the context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import phi_amd
    with pytest.raises(phi_amd.PhiError) as e:
        phi_amd.Context(0)
    assert e.value.status == phi_amd.PHI_ERR_DEVICE


def test_product_never_imports_the_oracle():
    """The checker must not leak into the product path."""
    pkg = os.path.join(ROOT, "phi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f == "build.py":
                continue        # only runs `make -C oracle` (building the checker is not using it)
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt and "phi_oracle" not in txt, f


def test_phi_host_abi_exports(built):
    from phi_amd import ilp_index as H
    names = _declared("phi_host.h")
    assert sorted(H.HOST_SYMBOLS) == names
    L = H.host_lib()
    for n in names:
        assert hasattr(L, n), n


def test_host_gfa_reader_vs_reference_goldens(built, oracle):
    from phi_amd import ilp_index as H
    gold = json.load(open(os.path.join(GOLDEN, "gfa_flatten.json")))
    g = H.Graph(os.path.join(DATA, "test.gfa"))
    t = gold["test.gfa"]
    assert g.seg_names == t["seg_names"] and g.hap_id2name == t["hap_names"]
    assert [bytes(g.seq_concat[g.seq_off[v]:g.seq_off[v + 1]]).decode() for v in range(g.n_vtx)] == t["node_seq"]
    assert [sorted(g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist()) for v in range(g.n_vtx)] == t["adj"]
    assert [g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.num_walks)] == t["paths"]
    g = H.Graph(os.path.join(DATA, "MHC_4.gfa.gz"))
    m = gold["MHC_4.gfa.gz"]
    assert g.n_vtx == m["n_vtx"] and len(g.adj) == m["n_edges"] and g.hap_id2name == m["hap_names"]
    assert hashlib.sha256(g.seq_concat.tobytes()).hexdigest() == m["sha256_seq"]
    assert hashlib.sha256(g.seq_off.tobytes()).hexdigest() == m["sha256_seq_off"]
    assert hashlib.sha256(g.walk_vtx.tobytes()).hexdigest() == m["sha256_walk_vtx"]
    adj_sorted = [sorted(g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist()) for v in range(g.n_vtx)]
    assert hashlib.sha256(json.dumps(adj_sorted).encode()).hexdigest() == m["sha256_adj_sorted"]
    # topological ranks: a permutation with every edge going forward; identical to the oracle's Kahn order
    assert sorted(g.top_order_map.tolist()) == list(range(g.n_vtx))
    og = oracle.parse_gfa(os.path.join(DATA, "MHC_4.gfa.gz"))
    assert g.top_order_map.tolist() == og.top_rank


def test_host_gfa_reader_w_line_tags(built, oracle, tmp_path, monkeypatch):
    """Optional tags behind the walk of a W-line (gfa-io.cpp reads the walk up to the next tab).  The reader does not search
    a walk for that tab on its own -- the pass that counts the steps looks for it, and a walk that has one is cut off there
    and its pieces are made again: the graph must be the one of the same file without the tags (and the reference parser's),
    with pieces of a few bytes as well as whole lines."""
    from phi_amd import ilp_index as H
    src = open(os.path.join(DATA, "test.gfa")).read().splitlines()
    tagged = [l + "\tXX:Z:a<b>c\tYY:i:7" if l.startswith("W\t") and i % 2 == 0 else l for i, l in enumerate(src)]
    assert tagged != src
    (tmp_path / "plain.gfa").write_text("\n".join(src) + "\n")
    (tmp_path / "tagged.gfa").write_text("\n".join(tagged) + "\n")
    want = H.Graph(str(tmp_path / "plain.gfa"))
    ref = oracle.parse_gfa(str(tmp_path / "tagged.gfa"))
    for piece in (None, "16", "40"):
        if piece: monkeypatch.setenv("PHI_GFA_PIECE", piece)
        g = H.Graph(str(tmp_path / "tagged.gfa"))
        assert g.hap_id2name == want.hap_id2name and g.seg_names == want.seg_names
        assert g.walk_off.tolist() == want.walk_off.tolist() and g.walk_vtx.tolist() == want.walk_vtx.tolist()
        assert g.adj_off.tolist() == want.adj_off.tolist() and g.adj.tolist() == want.adj.tolist()
        assert [g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.num_walks)] == ref.paths


def test_host_reader_over_memory_and_blocks_equals_kseq_vectors(built):
    """phi_reads_stream_open_blocks (the way a stream whose beginning the device has taken is finished on the exact state
    machine): the records of prefix + blocks are kseq's records of the whole text, however the text is cut."""
    from phi_amd import ilp_index as H
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))
    rng = np.random.default_rng(3)
    for case in gold["texts"]:
        text = bytes.fromhex(case["text_hex"])
        want = [bytes.fromhex(s_) for _, s_ in case["records"]]
        for _ in range(4):
            cuts = sorted(rng.integers(0, len(text) + 1, size=int(rng.integers(0, 5))).tolist())
            parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
            b, o = H.reads_of_text(parts[0], [p for p in parts[1:] if p])
            raw = bytes(b)
            assert [raw[o[i]:o[i + 1]] for i in range(len(o) - 1)] == want, (case["text_hex"], cuts)


def _boundary_text(c):
    return bytes.fromhex(c["head_hex"]) + bytes.fromhex(c["unit_hex"]) * c["count"] + bytes.fromhex(c["tail_hex"])


def test_kseq_block_boundary_vectors(built, oracle, tmp_path):
    """kseq flags the end of its stream when a 65 536-byte block comes back short (kseq.h:81,113,242), so what a file's
    very last bytes mean -- a bare header character, a lone CR -- depends on the file's size modulo 65 536.  The vectors
    were made by the reference's own kseq.h (tests/golden/make_golden.py); every reader here must agree: from a file, from
    memory blocks cut anywhere, from the middle of a stream (stream_offset), and the oracle's restatement."""
    import hashlib
    from phi_amd import ilp_index as H
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))["block_boundary"]
    assert len(gold) == 48 and sum(c["n_records"] == c["count"] + 2 for c in gold if c["tail_hex"] in ("3e", "40")) == 4   # the bare header character at a multiple of 65 536
    rng = np.random.default_rng(8)
    for c in gold:
        text = _boundary_text(c)
        assert len(text) == c["size"]

        def check(b, o, what):
            raw = bytes(b)
            recs = [raw[o[i]:o[i + 1]] for i in range(len(o) - 1)]
            assert len(recs) == c["n_records"], (what, c["tail_hex"], c["size"], len(recs))
            assert hashlib.sha256(b"\0".join(recs)).hexdigest() == c["sha256_seqs"], (what, c["tail_hex"], c["size"])
            assert [r.hex() for r in recs[-2:]] == [x[1] for x in c["last_records"]], (what, c["tail_hex"], c["size"])
        p = tmp_path / "k.fa"
        p.write_bytes(text)
        b, o, names = H.read_reads(str(p))
        check(b, o, "phi_reads_read")
        assert names[-1] == c["last_records"][-1][0]
        got = list(H.stream_reads(str(p), bases_cap=4096, reads_cap=100))
        lens = np.concatenate([np.diff(y) for _, y in got]) if got else np.zeros(0, np.int64)
        check(np.concatenate([x for x, _ in got]) if got else np.zeros(0, np.uint8), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), "phi_reads_stream")
        # blocks cut anywhere
        cuts = sorted(rng.integers(0, len(text) + 1, size=3).tolist())
        parts = [text[a:b_] for a, b_ in zip([0] + cuts, cuts + [len(text)])]
        check(*H.reads_of_text(parts[0], [x for x in parts[1:] if x]), "blocks")
        # the middle of a stream: the records before the cut are the device's, the rest is read with the stream's offset
        n_skip = int(rng.integers(1, c["count"]))
        at = len(bytes.fromhex(c["head_hex"])) + 8 * n_skip
        hb, ho = H.reads_of_text(text[at:], stream_offset=at)
        check(np.concatenate([np.frombuffer(b"A" + b"ACGT" * n_skip, np.uint8), hb]),
              np.concatenate([[0], 1 + np.arange(n_skip, dtype=np.int64) * 4, ho + 1 + 4 * n_skip]), "from the middle")
        orc = oracle.read_reads(str(p))
        assert len(orc) == c["n_records"] and [[a.decode("latin1"), s_.hex()] for a, s_ in orc[-2:]] == c["last_records"], ("oracle", c["tail_hex"], c["size"])


def test_host_gfa_reader_errors(built, tmp_path):
    from phi_amd import ilp_index as H
    with pytest.raises(H.HostError) as e:
        H.Graph(str(tmp_path / "missing.gfa"))
    assert e.value.status == -1
    # the third walk disagrees with the strand the first walk fixed for s2: reverse-strand vertex
    p = tmp_path / "rev.gfa"
    p.write_text("S\ts1\tACGT\nS\ts2\tGGA\nS\ts3\tTT\nL\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts3\t+\t0M\n"
                 "W\ta\t0\tc\t0\t1\t>s1>s2>s3\nW\tb\t0\tc\t0\t1\t>s1<s2>s3\n")
    with pytest.raises(H.HostError) as e:
        H.Graph(str(p))
    assert e.value.status == -2
    p = tmp_path / "cyc.gfa"
    p.write_text("S\ts1\tACGT\nS\ts2\tGGA\nL\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts1\t+\t0M\nW\ta\t0\tc\t0\t1\t>s1>s2\n")
    with pytest.raises(H.HostError) as e:
        H.Graph(str(p))
    assert e.value.status == -3
    # a link onto the REVERSE strand of its target (L b - a -: the link a -> b written from the other strand): the forward arc it
    # implies is the complement, which the reference appends behind its sorted arc array and finds again only sometimes
    # (gfa-base.cpp:269-303) -- refused (-5), not guessed; L a - b + adds nothing on either side and is accepted
    segs = "S\ts1\tACGT\nS\ts2\tGGA\nS\ts3\tTT\n"
    for links in ("L\ts1\t+\ts2\t+\t0M\nL\ts3\t-\ts2\t-\t0M\n", "L\ts3\t-\ts2\t-\t0M\nL\ts1\t+\ts2\t+\t0M\n", "L\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts3\t-\t0M\n"):
        p = tmp_path / "revlink.gfa"
        p.write_text(segs + links + "W\ta\t0\tc\t0\t1\t>s1>s2\n")
        with pytest.raises(H.HostError) as e:
            H.Graph(str(p))
        assert e.value.status == -5 and "reverse strand" in str(e.value)
    p = tmp_path / "revsrc.gfa"
    p.write_text(segs + "L\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts3\t+\t0M\nL\ts3\t-\ts1\t+\t0M\nW\ta\t0\tc\t0\t1\t>s1>s2>s3\n")
    g = H.Graph(str(p))
    assert [g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist() for v in range(3)] == [[1], [2], []]
    # a fully reversed walk is flipped to the forward strand (gfa_walk_flip), CRLF and no final newline
    p = tmp_path / "flip.gfa"
    p.write_bytes(b"S\ts1\tACGT\r\nS\ts2\tGGA\r\nL\ts1\t+\ts2\t+\t0M\r\nW\ta\t0\tc\t0\t1\t>s1>s2\r\nW\tb\t1\tc\t0\t1\t<s2<s1")
    g = H.Graph(str(p))
    assert g.walk_vtx.tolist() == [0, 1, 0, 1] and g.hap_id2name == ["a.0", "b.1"]


def test_host_reads_reader(built, oracle, tmp_path):
    from phi_amd import ilp_index as H
    bases, off, names = H.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))
    exp = oracle.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))
    assert len(names) == len(exp) == 16401
    raw = bases.tobytes()
    assert [raw[off[i]:off[i + 1]] for i in range(len(names))] == [s for _, s in exp]
    assert [n.encode() for n in names] == [n for n, _ in exp]
    p = tmp_path / "r.fa"
    p.write_text(">a desc\nACGT\nAC GT\n>b\n\n>c\nTT\n@q1 x\nGGCC\n+\nIIII\n@q2\nAA\nCC\n+q2\nII\nII\n")
    bases, off, names = H.read_reads(str(p))
    raw = bases.tobytes()
    assert names == ["a", "b", "c", "q1", "q2"]
    # (kseq keeps a sequence line as it is: the blank inside "AC GT" stays, kseq.h:209-213)
    assert [raw[off[i]:off[i + 1]] for i in range(5)] == [b"ACGTAC GT", b"", b"TT", b"GGCC", b"AACC"]
    if oracle.ref_available():
        assert oracle.ref_read_reads(str(p)) == oracle.read_reads(str(p))
    assert [(n.decode(), s) for n, s in oracle.read_reads(str(p))] == list(zip(names, [raw[off[i]:off[i + 1]] for i in range(5)]))


def test_host_hap_name_and_fasta(built, tmp_path):
    from phi_amd import ilp_index as H
    for c in json.load(open(os.path.join(GOLDEN, "hap_names.json"))):
        assert H.get_hap_name(c["gfa"], c["reads"]) == c["name"]
    L = H.host_lib()
    seq = b"ACGT" * 50 + b"A"
    out = tmp_path / "o.fa"
    assert L.phi_write_fasta(str(out).encode(), b"name_x", seq, len(seq)) == 0
    lines = out.read_text().split("\n")
    assert lines[0] == ">name_x LN:201" and [len(x) for x in lines[1:]] == [80, 80, 41, 0]
    assert "".join(lines[1:]).encode() == seq
    assert L.phi_write_fasta(str(out).encode(), b"empty", b"", 0) == 0
    assert out.read_text() == ">empty LN:0\n"           # the reference's output when the solve fails (:1583-1598)


def test_device_primitives_on_host(oracle, tmp_path):
    """phi_dev.h is host/device code: compile it with g++ and check the 2-bit k-mer hash, the
    reverse complement and the base codes against the reference's murmur known answers."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "phi_dev.h"\nextern "C" {\n'
                   'uint64_t t_hash(uint64_t v, int k) { return phi_kmer_hash(v, k); }\n'
                   'uint64_t t_rc(uint64_t v, int k) { return phi_revcomp(v, k); }\n'
                   'uint32_t t_code(uint32_t c) { return phi_code(c); }\n'
                   'int t_ok(uint32_t c) { return phi_is_acgt(c); }\n'
                   'uint64_t t_ext(const uint64_t* w, long i) { return phi_extract64(w, i); }\n}\n')
    lib = tmp_path / "libt.so"
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "phi_amd", "csrc"), "-o", str(lib), str(src)])
    L = C.CDLL(str(lib))
    L.t_hash.restype = C.c_uint64
    L.t_hash.argtypes = [C.c_uint64, C.c_int]
    L.t_rc.restype = C.c_uint64
    L.t_rc.argtypes = [C.c_uint64, C.c_int]
    L.t_ext.restype = C.c_uint64
    L.t_ext.argtypes = [C.c_void_p, C.c_long]

    def val(s):
        v = 0
        for ch in s:
            v = (v << 2) | b"ACGT".index(ch)
        return v
    n_checked = 0
    for v in json.load(open(os.path.join(GOLDEN, "murmur_vectors.json"))):
        b = bytes.fromhex(v["hex"])
        if 1 <= len(b) <= 32 and set(b) <= set(b"ACGT"):
            assert L.t_hash(val(b), len(b)) == int(v["hash"])
            n_checked += 1
    assert n_checked >= 30
    rng = np.random.default_rng(9)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for k in range(1, 33):
        s = bytes(rng.choice(list(b"ACGT"), size=k).tolist())
        assert L.t_hash(val(s), k) == oracle.hash128_to_64(s)
        assert L.t_rc(val(s), k) == val(s.translate(comp)[::-1])
    for ch in b"ACGTacgt":
        assert L.t_code(ch) == b"ACGT".index(bytes([ch]).upper()) and L.t_ok(ch)
    for ch in b"NnRYxX*-0":
        assert not L.t_ok(ch)
    seq = bytes(rng.choice(list(b"ACGT"), size=200).tolist())
    words = np.zeros(9, np.uint64)
    for i, ch in enumerate(seq):
        words[i // 32] |= np.uint64(b"ACGT".index(ch)) << np.uint64(62 - 2 * (i % 32))
    for i in (0, 1, 31, 32, 33, 100, 167):
        assert L.t_ext(words.ctypes.data, i) == val(seq[i:i + 32])


def test_synthetic_gfa_round_trip(tmp_path):
    """phi_amd.synth.write_gfa / write_reads -> phi_gfa_read / phi_reads_read give back the generator's
    arrays (segment ids are first-seen order, so even the numbering agrees); plain and gzip."""
    import numpy as np
    from phi_amd import ilp_index as H
    from phi_amd import synth
    gk, rk = synth.CONFIGS["tiny"]
    g = synth.make_graph(**gk)
    bases, off, _ = synth.make_reads(g, **rk)
    for ext in ("", ".gz"):
        gfa, fq = str(tmp_path / ("t.gfa" + ext)), str(tmp_path / ("t.fq" + ext))
        synth.write_gfa(g, gfa)
        synth.write_reads(bases, off, fq, fastq=True)
        G = H.Graph(gfa)
        assert (G.n_vtx, G.num_walks) == (g.n_vtx, g.n_walks)
        assert np.array_equal(G.seq_off, g.seq_off) and np.array_equal(G.seq_concat, g.seq_concat)
        assert np.array_equal(G.adj_off, g.adj_off)
        for v in range(g.n_vtx):                       # the reader sorts each adjacency list
            assert sorted(g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist()) == G.adj[G.adj_off[v]:G.adj_off[v + 1]].tolist()
        assert np.array_equal(G.walk_off, g.walk_off) and np.array_equal(G.walk_vtx, g.walk_vtx)
        assert G.hap_id2name == g.hap_names
        # a topological order: every edge goes forward
        src = np.repeat(np.arange(g.n_vtx), np.diff(G.adj_off))
        assert (G.top_order_map[src] < G.top_order_map[G.adj]).all()
        reads = []
        idx = H.ILP_index.__new__(H.ILP_index)
        H.ILP_index.read_ip_reads(idx, reads, fq)
        assert len(reads) == len(off) - 1
        assert b"".join(r[1] for r in reads) == bases.tobytes()
        assert [len(r[1]) for r in reads] == np.diff(off).tolist()


def test_host_reads_stream_equals_whole_file_reader(built, tmp_path):
    """phi_reads_stream_* (chunks into caller buffers) against phi_reads_read on the reference's
    fixtures and on awkward files, with chunk sizes that cut records, lines and quality blocks."""
    from phi_amd import ilp_index as H
    rng = np.random.default_rng(5)
    files = [os.path.join(DATA, "CHM13_reads.fq.gz"), os.path.join(DATA, "read.fa")]
    # multi-line FASTA, CRLF, empty lines, lower case, a record without sequence, no trailing newline
    p = tmp_path / "multi.fa"
    recs = []
    for i in range(40):
        L = int(rng.integers(0, 400))
        seq = bytes(rng.choice(list(b"ACGTacgtN"), size=L).tolist())
        lines = [seq[j:j + 60] for j in range(0, L, 60)]
        recs.append(b">r%d some comment\r\n" % i + b"\r\n".join(lines) + (b"\n\n" if i % 3 == 0 else b"\n"))
    p.write_bytes(b"".join(recs) + b">last\nACGT")
    files.append(str(p))
    # FASTQ whose quality lines start with '@' and '+', multi-line sequence and quality
    q = tmp_path / "tricky.fq"
    out = []
    for i in range(30):
        L = int(rng.integers(1, 300))
        seq = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        qual = bytes(rng.choice(list(b"@+>I5"), size=L).tolist())
        w = int(rng.integers(20, 80))
        out.append(b"@q%d\n" % i + b"\n".join(seq[j:j + w] for j in range(0, L, w)) + b"\n+\n" +
                   b"\n".join(qual[j:j + w] for j in range(0, L, w)) + b"\n")
    q.write_bytes(b"".join(out))
    files.append(str(q))
    for f in files:
        bases, off, _ = H.read_reads(f)
        longest = int(np.diff(off).max()) if len(off) > 1 else 1
        for bases_cap, reads_cap in [(64 << 20, 1 << 20), (max(longest, 1), 7), (max(longest, 1) + 13, 1000), (4096, 3)]:
            if bases_cap < longest:
                continue
            got_b, got_len = [], []
            for b, o in H.stream_reads(f, bases_cap=bases_cap, reads_cap=reads_cap):
                assert o[0] == 0 and len(o) - 1 <= reads_cap and o[-1] <= bases_cap and o[-1] == len(b)
                got_b.append(b)
                got_len.append(np.diff(o))
            gb = np.concatenate(got_b) if got_b else np.zeros(0, np.uint8)
            gl = np.concatenate(got_len) if got_len else np.zeros(0, np.int64)
            assert np.array_equal(gl, np.diff(off)), (f, bases_cap, reads_cap)
            assert np.array_equal(gb, bases), (f, bases_cap, reads_cap)
    # a read longer than the chunk is an error, not a truncation
    with pytest.raises(H.HostError):
        list(H.stream_reads(files[0], bases_cap=100, reads_cap=10))


def test_reads_readers_on_the_reference_kseq_vectors(built, oracle, tmp_path):
    """tests/golden/kseq_vectors.json: FASTA / FASTQ texts, well-formed and malformed, with the records the
    reference's own kseq.h returns (make_golden.py).  The oracle's restatement, phi_reads_read and the streaming
    reader must return exactly those -- including where kseq gives up (a quality string of another length ends
    the file) and what it keeps (white space inside a sequence line, a CR on an otherwise empty first line)."""
    from phi_amd import ilp_index as H
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))
    for i, v in enumerate(gold["texts"]):
        p = tmp_path / f"k{i}.fq"
        p.write_bytes(bytes.fromhex(v["text_hex"]))
        want = [(a.encode("latin1"), bytes.fromhex(b)) for a, b in v["records"]]
        assert oracle.read_reads(str(p)) == want, (i, v["text_hex"])
        bases, off, names = H.read_reads(str(p))
        assert [(names[j].encode("latin1"), bytes(bases[off[j]:off[j + 1]])) for j in range(len(names))] == want, (i, v["text_hex"])
        longest = max([len(b) for _, b in want] + [1])
        for cap, rc in ((longest, 1), (longest + 5, 3), (1 << 16, 1 << 10)):
            got = []
            for b, o in H.stream_reads(str(p), bases_cap=cap, reads_cap=rc):
                got += [bytes(b[o[j]:o[j + 1]]) for j in range(len(o) - 1)]
            assert got == [b for _, b in want], (i, cap, rc)
    c = gold["CHM13_reads.fq.gz"]
    recs = oracle.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))
    assert len(recs) == c["n"]
    assert hashlib.sha256(b"\0".join(a for a, _ in recs)).hexdigest() == c["sha256_names"]
    assert hashlib.sha256(b"\0".join(b for _, b in recs)).hexdigest() == c["sha256_seqs"]
    bases, off, names = H.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))
    assert hashlib.sha256(b"\0".join(bytes(bases[off[j]:off[j + 1]]) for j in range(len(names)))).hexdigest() == c["sha256_seqs"]
    if oracle.ref_available():                         # and, where it is built, the reference's reader itself
        assert oracle.ref_read_reads(os.path.join(DATA, "read.fa")) == oracle.read_reads(os.path.join(DATA, "read.fa"))


def _bgzf(data, block=None, rng=None):
    """Block gzip as bgzip / htslib write it: RFC 1952 members of <= 64 KB of input whose extra field 'BC' holds the
    member's size minus one, closed by an empty member."""
    import struct
    import zlib
    out = []
    pos = 0
    while True:
        n = 0 if pos >= len(data) else (block if block else int(rng.integers(1, 65280)))
        chunk = data[pos:pos + n]
        pos += n
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 18 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<HBBHH", 6, 66, 67, 2, bsize - 1) + body +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        if not chunk:
            break
    return b"".join(out)


def test_gzip_inputs_plain_block_and_concatenated(built, oracle, tmp_path, monkeypatch):
    """gz_source.h: block gzip (BGZF) is inflated by a pool of threads, member boundaries cutting records, lines and
    quality strings anywhere; plain gzip by one thread ahead of the parser; concatenated members, bytes after the
    last member and a truncated file behave as they do under gzread (what the reference's kseq reads through).
    Every variant must give the records of the uncompressed file, for the reads readers and the GFA reader."""
    import gzip
    import zlib
    from phi_amd import ilp_index as H
    rng = np.random.default_rng(77)
    recs = []
    for i in range(3000):
        L = int(rng.integers(0, 500))
        seq = bytes(rng.choice(list(b"ACGTN"), size=L).tolist())
        if i % 2:
            recs.append(b"@q%d c\n%s\n+\n%s\n" % (i, seq, bytes(rng.choice(list(b"@+>I"), size=L).tolist())))
        else:
            recs.append(b">r%d\n" % i + b"\n".join(seq[j:j + 70] for j in range(0, L, 70)) + b"\n")
    text = b"".join(recs)
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    want_b, want_o, want_n = H.read_reads(str(plain))
    variants = {
        "gzip.gz": gzip.compress(text, 5),
        "bgzf.gz": _bgzf(text, rng=rng),
        "bgzf_tiny_blocks.gz": _bgzf(text[:200_000], block=137) ,
        "members.gz": gzip.compress(text[:300_001]) + gzip.compress(text[300_001:700_000]) + gzip.compress(text[700_000:]),
        "bgzf_then_plain_member.gz": _bgzf(text[:400_000], rng=rng)[:-28] + gzip.compress(text[400_000:]),
        "trailing_garbage.gz": gzip.compress(text) + b"this is no gzip member",
    }
    for threads in ("1", "5"):
        monkeypatch.setenv("PHI_HOST_THREADS", threads)
        for name, blob in variants.items():
            p = tmp_path / name
            p.write_bytes(blob)
            b, o, n = H.read_reads(str(p))
            if name == "bgzf_tiny_blocks.gz":
                pp = tmp_path / "part.fq"
                pp.write_bytes(text[:200_000])
                eb, eo, en = H.read_reads(str(pp))
            else:
                eb, eo, en = want_b, want_o, want_n
            assert np.array_equal(o, eo) and np.array_equal(b, eb) and n == en, (name, threads)
            got = np.concatenate([x for x, _ in H.stream_reads(str(p), bases_cap=100_000, reads_cap=300)])
            assert np.array_equal(got, eb), (name, threads)
    # a truncated gzip file gives the records that inflate, as gzread does (the last one possibly cut)
    cut = tmp_path / "cut.gz"
    cut.write_bytes(gzip.compress(text)[:200_000])
    b, o, n = H.read_reads(str(cut))
    assert 100 < len(n) < len(want_n) and n[:len(n) - 1] == want_n[:len(n) - 1]
    assert np.array_equal(o[:len(n)], want_o[:len(n)])
    # a corrupted BGZF member, or a member whose CRC does not match, is an I/O error -- never a silently shorter read set
    bad = bytearray(_bgzf(text, block=60_000))
    bad[len(bad) // 2] ^= 0xFF
    (tmp_path / "bad.gz").write_bytes(bytes(bad))
    crc = bytearray(_bgzf(text, block=60_000))
    first = 18 + int.from_bytes(crc[16:18], "little") + 1 - 18      # size of the first member
    crc[first - 8] ^= 0x01                                           # its CRC32 field
    (tmp_path / "crc.gz").write_bytes(bytes(crc))
    for name in ("bad.gz", "crc.gz"):
        with pytest.raises(H.HostError) as e:
            H.read_reads(str(tmp_path / name))
        assert e.value.status == -1
        with pytest.raises(H.HostError):
            list(H.stream_reads(str(tmp_path / name)))
        with pytest.raises(H.HostError):
            list(H.text_chunks(str(tmp_path / name), 1 << 20))
    with pytest.raises(H.HostError):
        H.Graph(str(tmp_path / "bad.gz"))
    # block gzip followed by bytes that are no gzip member: ignored, as gzread ignores them -- nothing before them is lost
    for threads in ("1", "5"):
        monkeypatch.setenv("PHI_HOST_THREADS", threads)
        (tmp_path / "bgzf_garbage.gz").write_bytes(_bgzf(text, rng=rng) + b"no gzip member here")
        b, o, n = H.read_reads(str(tmp_path / "bgzf_garbage.gz"))
        assert np.array_equal(o, want_o) and np.array_equal(b, want_b) and n == want_n
        # the raw text source gives the file's bytes whatever the chunk size
        for name in ("gzip.gz", "bgzf.gz", "members.gz", "bgzf_garbage.gz", "r.fq"):
            assert b"".join(c.tobytes() for c in H.text_chunks(str(tmp_path / name), 99_991)) == text, name
    # the GFA reader through the same source
    g1 = H.Graph(os.path.join(DATA, "MHC_4.gfa.gz"))
    raw = gzip.open(os.path.join(DATA, "MHC_4.gfa.gz"), "rb").read()
    (tmp_path / "g.bgzf.gfa.gz").write_bytes(_bgzf(raw, rng=rng))
    g2 = H.Graph(str(tmp_path / "g.bgzf.gfa.gz"))
    for name in ("seq_off", "adj_off", "adj", "walk_off", "walk_vtx", "top_order_map"):
        assert np.array_equal(getattr(g1, name), getattr(g2, name)), name
    assert bytes(g1.seq_concat) == bytes(g2.seq_concat) and list(g1.hap_id2name) == list(g2.hap_id2name)


@pytest.mark.parametrize("env", [{"PHI_GFA_SLICE": "64"}, {"PHI_GFA_SLICE": "5000", "PHI_HOST_THREADS": "3"}, {"PHI_GFA_NAMES_SERIAL": "1"}])
def test_host_gfa_reader_slices_and_name_entry(built, monkeypatch, env):
    """The reader cuts the text fine among short lines and coarse among long ones, and enters <prefix><number> segment names
    on all threads at once: whatever the cuts, and with the names entered one by one instead, the graph is the same."""
    from phi_amd import ilp_index as H
    path = os.path.join(DATA, "MHC_4.gfa.gz")
    want = H.Graph(path)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    g = H.Graph(path)
    assert g.hap_id2name == want.hap_id2name and g.seg_names == want.seg_names
    for f in ("seq_off", "seq_concat", "adj_off", "adj", "walk_off", "walk_vtx", "top_order_map"):
        assert np.array_equal(getattr(g, f), getattr(want, f)), f


def test_deferred_walks_resolved_by_the_host_after_all(built, tmp_path):
    """phi_gfa_read_deferred + phi_graph_resolve_walks == phi_gfa_read (what a caller does when the walk text is not of the
    kind the device takes); the walk texts it hands out are the W-lines' walk fields (tags included: the device cuts them)."""
    from phi_amd import ilp_index as H
    import ctypes as C
    for name in ("test.gfa", "MHC_4.gfa.gz"):
        path = os.path.join(DATA, name)
        want = H.Graph(path)
        g = H.DeferredGraph(path)
        assert g.walk_off is None and g.num_walks == want.num_walks
        texts = g.walk_texts()
        assert len(texts) == want.num_walks
        for (addr, n), h in zip(texts, range(want.num_walks)):
            txt = C.string_at(addr, n).decode()
            assert txt.count(">") + txt.count("<") == want.walk_off[h + 1] - want.walk_off[h]
        g.resolve_on_host()
        for f in ("seq_off", "seq_concat", "adj_off", "adj", "walk_off", "walk_vtx", "top_order_map"):
            assert np.array_equal(getattr(g, f), getattr(want, f)), f
        assert g.hap_id2name == want.hap_id2name
    # a walk error comes out of the deferred resolution as it does out of phi_gfa_read
    p = tmp_path / "rev.gfa"
    p.write_text("S\ts1\tACGT\nS\ts2\tGGA\nS\ts3\tTT\nL\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts3\t+\t0M\nW\ta\t0\tc\t0\t1\t>s1>s2>s3\nW\tb\t0\tc\t0\t1\t>s1<s2>s3\n")
    with pytest.raises(H.HostError) as e1:
        H.Graph(str(p))
    g = H.DeferredGraph(str(p))
    with pytest.raises(H.HostError) as e2:
        g.resolve_on_host()
    assert e1.value.args == e2.value.args
