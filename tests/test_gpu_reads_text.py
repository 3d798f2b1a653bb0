"""GPU parity tests of the device-side record splitter (phi_add_reads_text, phi_amd/csrc/reads_text.hip) against the
host reader (the exact restatement of the reference's kseq, src/kseq.h:192-233, ILP_index.cpp:313-328) and the kseq
golden vectors made by the reference's own header (tests/golden/kseq_vectors.json).

The contract: (records the device takes) + (records the host state machine finds in what the device hands back,
followed by the rest of the stream) == kseq's records of the whole text -- for regular text (the device takes all but
the last record), for irregular text (the device takes a prefix, or nothing) and for any way of cutting the stream."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(ctx_factory, oracle):
    g = oracle.parse_gfa(os.path.join(DATA, "test.gfa"))
    A = g.arrays()
    c = ctx_factory(k=3, w=2, threshold=1.0, recombination=100)
    c.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    return c


def _records(bases, off):
    raw = bytes(bases)
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def split_on_device(ctx, text, call_bytes, max_chunk=None, totals_only=False, park=None):
    """Records of `text` with the device taking what it can, in calls of call_bytes.  Returns (records, bytes the
    device took, went irregular)."""
    from phi_amd import ilp_index as H
    ctx.reset_reads()
    ctx.reads_text_begin(max_chunk or max(call_bytes, 64))
    recs = []
    irregular, rest_at = False, len(text)
    for j, i in enumerate(range(0, len(text), call_bytes)):
        if park is not None and j % 3 != 2:
            # the piece waits in device memory first (phi_text_park_*): two of three pieces, the third from the host
            idx = park.add(text[i:i + call_bytes])
            assert park.fetch(idx) == text[i:i + call_bytes]
            irr = ctx.add_reads_text_parked(park, idx)
            park.release(idx)
        else:
            irr = ctx.add_reads_text(text[i:i + call_bytes])
        if irr:
            irregular, rest_at = True, i + call_bytes
            break
        recs += _records(*ctx.reads_text_last_batch())
    pending, taken = ctx.reads_text_end()
    hb, ho = H.reads_of_text(pending, [text[rest_at:]] if rest_at < len(text) else [], stream_offset=taken)
    host_recs = _records(hb, ho)
    st = ctx.reads_stats()
    if not totals_only:
        assert max_chunk is None or max_chunk >= call_bytes
        # what the device took is what the context counted (the host part is not added here)
        assert st["n_reads"] == len(recs) and st["n_bases"] == sum(map(len, recs))
        return recs + host_recs, taken, irregular
    # a call longer than the device buffers is cut into pieces inside the library: only the totals are seen from here
    return (st["n_reads"], st["n_bases"], host_recs), taken, irregular


def kseq_records(text):
    from phi_amd import ilp_index as H
    return _records(*H.reads_of_text(text))


def test_kseq_golden_vectors_through_the_device_splitter(ctx):
    """Every text of the reference's kseq vectors (regular, wrapped, malformed, CRLF, empty lines, truncated ...):
    device + host rest == the records the reference's kseq.h returned, whole and in calls of 1 .. 64 bytes."""
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))
    n_regular = 0
    for case in gold["texts"]:
        text = bytes.fromhex(case["text_hex"])
        want = [bytes.fromhex(s) for _, s in case["records"]]
        assert kseq_records(text) == want                       # (the host reader itself, as tests/test_cpu_abi_host.py pins it)
        if not text:
            continue
        for call in (len(text), 1, 2, 3, 5, 7, 16, 64):
            got, taken, irregular = split_on_device(ctx, text, call, max_chunk=max(64, call))
            assert got == want, (case["text_hex"], call, irregular)
            n_regular += not irregular
    assert n_regular > 20


def test_parked_pieces_are_taken_as_host_pieces_are(ctx):
    """phi_add_reads_text_parked: the piece's bytes are in device memory already (a park filled before the graph was there) --
    the same records, the same bytes handed back at the end or at irregular text (fetched from the device then: nobody has
    them on the host), pieces from the park and from the host mixed in one stream.  The reference's kseq vectors and random
    texts with anomalies, in calls of many sizes."""
    from phi_amd.context import TextPark
    park = TextPark(0)
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))
    n_regular = 0
    for case in gold["texts"]:
        text = bytes.fromhex(case["text_hex"])
        want = [bytes.fromhex(s) for _, s in case["records"]]
        if not text:
            continue
        for call in (len(text), 1, 3, 7, 16, 64):
            got, taken, irregular = split_on_device(ctx, text, call, max_chunk=max(64, call), park=park)
            assert got == want, (case["text_hex"], call, irregular)
            n_regular += not irregular
    assert n_regular > 20
    rng = np.random.default_rng(77)
    for _ in range(120):
        text = _random_text(rng)
        want = kseq_records(text)
        call = int(rng.choice([len(text), 64, 200, 1000, int(rng.integers(1, 5000))]))
        got, taken, irregular = split_on_device(ctx, text, max(call, 1), max_chunk=max(64, call), park=park)
        assert got == want, (text[:300], call, irregular)
    park.close()


def test_block_boundary_vectors_through_the_device_splitter(ctx):
    """Texts whose size is (or is next to) a multiple of kseq's 65 536-byte block, ending in a bare header character, a
    lone CR, an unfinished FASTQ record: what kseq makes of the last bytes depends on the size (kseq.h:81,113,242).  The
    device takes the regular records, the host reader gets the rest with its offset in the stream: together, the records
    the reference's kseq.h returned."""
    import hashlib
    gold = json.load(open(os.path.join(GOLDEN, "kseq_vectors.json")))["block_boundary"]
    for c in gold:
        text = bytes.fromhex(c["head_hex"]) + bytes.fromhex(c["unit_hex"]) * c["count"] + bytes.fromhex(c["tail_hex"])
        for call in (len(text), 30_000):
            got, taken, irregular = split_on_device(ctx, text, call, max_chunk=max(64, call))
            assert not irregular or c["tail_hex"].startswith("40") or "0d" in c["tail_hex"]   # a '+' line or a CR: irregular for a FASTA stream
            assert len(got) == c["n_records"] and hashlib.sha256(b"\0".join(got)).hexdigest() == c["sha256_seqs"], (c["tail_hex"], c["size"], call)


def _random_text(rng):
    """FASTA / FASTQ text, mostly regular, with the anomalies a reads file can hold mixed in."""
    fastq = rng.random() < 0.5
    n = int(rng.integers(1, 40))
    out = []
    anomaly = rng.random() < 0.45
    for i in range(n):
        L = int(rng.integers(0 if rng.random() < 0.1 else 1, 200))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), size=L).tolist())
        hdr = (b"@" if fastq else b">") + b"r%d" % i + (b" some comment" if rng.random() < 0.3 else b"")
        if fastq:
            q = bytes(rng.choice(list(b"@+>I#5"), size=L).tolist())
            rec = [hdr, seq, b"+" + (b"r%d" % i if rng.random() < 0.2 else b""), q]
            if anomaly and rng.random() < 0.15 and L > 4:       # wrapped record
                rec = [hdr, seq[:L // 2], seq[L // 2:], b"+", q[:L // 2], q[L // 2:]]
            if anomaly and rng.random() < 0.05:
                rec[-1] = q[:-1] if L else q                   # quality of another length
        else:
            width = int(rng.choice([0, 0, 60, 7]))
            lines = [seq[j:j + width] for j in range(0, L, width)] if width and L else [seq]
            rec = [hdr] + lines
            if anomaly and rng.random() < 0.1:
                rec.insert(1 + int(rng.integers(0, len(lines) + 1)), b"")   # an empty line
            if anomaly and rng.random() < 0.05:
                rec.append(b"+")                                # a quality block in a FASTA file
                rec.append(b"I" * L)
        out += rec
    if anomaly and rng.random() < 0.2:
        out.insert(0, bytes(rng.choice(list(b"xyz \t"), size=int(rng.integers(1, 5))).tolist()))   # text before the first header
    nl = b"\r\n" if anomaly and rng.random() < 0.2 else b"\n"
    text = nl.join(out) + (b"" if rng.random() < 0.3 else nl)
    if anomaly and rng.random() < 0.1:
        text = text[:int(rng.integers(0, len(text) + 1))]      # cut anywhere
    return text


def test_random_texts_device_plus_host_equals_kseq(ctx):
    rng = np.random.default_rng(20261004)
    n_irregular = n_device_reads = 0
    for it in range(260):
        text = _random_text(rng)
        if not text:
            continue
        want = kseq_records(text)
        for call in (len(text), int(rng.integers(1, 90)), int(rng.integers(90, 2000))):
            got, taken, irregular = split_on_device(ctx, text, call, max_chunk=max(64, call))
            assert got == want, (it, call, irregular, text[:300])
            n_irregular += irregular
            n_device_reads += taken > 0
        # the whole text in one call, cut into pieces of 64 .. 300 bytes by the library itself
        (n_dev, b_dev, host_recs), taken, irregular = split_on_device(ctx, text, len(text), max_chunk=int(rng.integers(64, 300)), totals_only=True)
        assert n_dev + len(host_recs) == len(want) and want[n_dev:] == host_recs and b_dev == sum(map(len, want[:n_dev])), (it, text[:300])
    assert n_irregular > 20 and n_device_reads > 300


def test_large_regular_files_score_like_host_parsed_reads(ctx_factory, oracle, tmp_path):
    """The reference's own read set (test/CHM13_reads.fq.gz, 16 401 x 150 bp) and a wrapped FASTA of long reads: the
    text path (chunks of 1 MB and of 50 kB, then whatever the device hands back through the host reader) leaves the
    context in the state that adding the host-parsed reads leaves it in -- counters, hit vector, spectrum -- and the
    solve gives the same result.  Also: chunk logs of novel hashes that spill into an overflow list too small for them -- the
    list is grown and the chunk replayed -- and a log that starts over with every chunk."""
    from phi_amd import ilp_index as H
    g = oracle.parse_gfa(os.path.join(DATA, "MHC_4.gfa.gz"))
    A = g.arrays()
    c = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
    c.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    fq = gzip.open(os.path.join(DATA, "CHM13_reads.fq.gz"), "rb").read()
    hb, ho = H.reads_of_text(fq)
    rng = np.random.default_rng(5)
    long_reads = []
    hap = bytes(g.arrays()["seq_concat"][:400_000])
    for i in range(60):
        L = int(rng.integers(2_000, 30_000))
        s = int(rng.integers(0, len(hap) - L))
        long_reads.append(b">long%d\n" % i + b"\n".join(hap[s + j:s + j + 80] for j in range(0, L, 80)) + b"\n")
    fa = b"".join(long_reads)

    def state(c):
        import torch
        from phi_amd import dist as pdist
        st = c.reads_stats()
        p, n = c.hits_buffer()
        hits = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy().copy()
        p, m = c.spectrum_export()
        sp = np.sort(torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64)) if m else np.zeros(0, np.uint64)
        res = c.solve()
        return st, hits, sp, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in res.items()}

    # a context whose chunk logs hold four novel hashes and whose overflow list starts at 40 entries (read at phi_set_graph / at
    # the list's first use): every chunk of text spills, fills the list, has it grown and is replayed; the log starts over with every chunk
    tight_env = {"PHI_NOV_SHIFT": "2", "PHI_OVLIST_CAP": "40", "PHI_NOVLOG_BUDGET": "65536"}
    os.environ.update(tight_env)
    try:
        c_tight = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
        c_tight.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    finally:
        for k_ in tight_env:
            del os.environ[k_]

    for text in (fq, fa):
        c.reset_reads()
        c.add_reads(H.reads_of_text(text))
        want = state(c)
        for chunk, env, cc in ((1 << 20, {}, c), (50_000, {}, c), (1 << 20, tight_env, c_tight)):
            for k_, v_ in env.items():
                os.environ[k_] = v_
            try:
                cc.reset_reads()
                cc.reads_text_begin(chunk)
                for i in range(0, len(text), chunk):
                    assert not cc.add_reads_text(text[i:i + chunk])
                pending, taken = cc.reads_text_end()
                # (FASTQ: every whole group of four lines is taken; FASTA: the last record waits for the end of the file)
                assert (0 < len(pending) < 70_000 if text is fa else pending == b"") and taken + len(pending) == len(text)
                cc.add_reads(H.reads_of_text(pending))
                got = state(cc)
            finally:
                for k_ in env:
                    del os.environ[k_]
            assert got[0] == want[0], (chunk, env)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[3] == want[3], (chunk, env)


def test_chunks_taken_in_turn_by_two_contexts(ctx_factory, oracle):
    """The several-GPU plumbing on one GPU: two contexts take the chunks of one stream in turn, the unfinished rest going
    from one to the next (phi_reads_text_detach_carry); together they take exactly what one context takes."""
    from phi_amd import ilp_index as H
    g = oracle.parse_gfa(os.path.join(DATA, "test.gfa"))
    A = g.arrays()
    cs = []
    for _ in range(2):
        c = ctx_factory(k=3, w=2, threshold=1.0, recombination=100)
        c.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        cs.append(c)
    rng = np.random.default_rng(9)
    recs = [b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), size=L).tolist()), b"I" * L)
            for i, L in enumerate(rng.integers(1, 120, size=300).tolist())]
    text = b"".join(recs)
    want = kseq_records(text)
    for chunk in (97, 1000):
        got, carry = [], b""
        for c in cs:
            c.reset_reads()
            c.reads_text_begin(1 << 12)
        for n, i in enumerate(range(0, len(text), chunk)):
            c = cs[n % 2]
            if carry:
                assert not c.add_reads_text(carry)
                got += _records(*c.reads_text_last_batch())
            assert not c.add_reads_text(text[i:i + chunk])
            got += _records(*c.reads_text_last_batch())
            carry = c.reads_text_detach_carry()
        for c in cs:
            pend, _ = c.reads_text_end()
            assert pend == b"" or pend == carry
        got += kseq_records(carry)
        assert got == want, chunk


def test_reads_of_one_length_without_offsets(ctx_factory, oracle):
    """Reads of one length need no offsets array (phi_add_reads_device with d_read_off = NULL; phi_add_reads detects them;
    the device-side record splitter hands them over that way): the kernel computes the read starts.  Same counters, hit
    vector, spectrum and solve as the same reads with their offsets -- for lengths around the kernel's limits (32: the
    shortest it takes, below: offsets are made), lengths that divide a chunk and lengths that do not, bases outside ACGT."""
    import torch
    from phi_amd import dist as pdist
    from phi_amd import ilp_index as H
    g = oracle.parse_gfa(os.path.join(DATA, "MHC_4.gfa.gz"))
    A = g.arrays()
    c = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
    c.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    hap = np.frombuffer(bytes(A["seq_concat"][:3_000_000]), np.uint8).copy()
    rng = np.random.default_rng(11)

    def state():
        st = c.reads_stats()
        p, n = c.hits_buffer()
        hits = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy().copy()
        p, m = c.spectrum_export()
        sp = np.sort(torch.as_tensor(pdist.DevArray(p, m, "<i8"), device="cuda").clone().cpu().numpy().view(np.uint64)) if m else np.zeros(0, np.uint64)
        return st, hits, sp

    for L, n in ((150, 20_000), (32, 5_000), (33, 5_000), (31, 3_000), (64, 9_000), (512, 3_000), (1000, 2_000), (4099, 700), (55, 1)):
        starts = rng.integers(0, len(hap) - L, size=n)
        bases = hap[(starts[:, None] + np.arange(L)[None, :])].reshape(-1).copy()
        bad = rng.integers(0, len(bases), size=max(1, len(bases) // 5000))
        bases[bad] = ord("N")
        low = rng.integers(0, len(bases), size=len(bases) // 50)
        bases[low] |= 0x20
        off = np.arange(n + 1, dtype=np.int64) * L
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off).cuda()
        c.reset_reads()
        c.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), n, n * L)
        want = state()
        c.reset_reads()
        c.add_reads_device(d_b.data_ptr(), None, n, n * L)
        got = state()
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (L, n)
        c.reset_reads()
        c.add_reads((bases, off))                              # the host path finds the one length by itself
        got = state()
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (L, n, "host")
        # as FASTQ text through the device-side splitter
        text = b"".join(b"@r\n" + bases[i * L:(i + 1) * L].tobytes() + b"\n+\n" + b"I" * L + b"\n" for i in range(min(n, 4000)))
        c.reset_reads()
        c.add_reads((bases[:min(n, 4000) * L], off[:min(n, 4000) + 1]))
        want_t = state()
        c.reset_reads()
        c.reads_text_begin(1 << 20)
        for i in range(0, len(text), 1 << 20):
            assert not c.add_reads_text(text[i:i + (1 << 20)])
        pend, _ = c.reads_text_end()
        assert pend == b""
        got = state()
        assert got[0] == want_t[0] and np.array_equal(got[1], want_t[1]) and np.array_equal(got[2], want_t[2]), (L, n, "text")
    with pytest.raises(Exception):
        c.add_reads_device(d_b.data_ptr(), None, 7, 100)        # 100 bases are not 7 reads of one length
