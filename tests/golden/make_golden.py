"""Regenerates the golden fixtures under tests/golden/ (run in the build container, where
/root/reference exists and oracle/_ref/libphi_ref.so has been built by `make -C oracle`).

  murmur_vectors.json   inputs + outputs of the REFERENCE's MurmurHash3_x64_128 (folded h1^h2 as in
                        ILP_index.cpp:10-18), produced by the reference's own MurmurHash3.cpp
  gfa_flatten.json      segments / arcs / walks of test/test.gfa as the REFERENCE's gfa_read() parses
                        them, flattened as ILP_index::read_gfa does; plus digests for test/MHC_4.gfa.gz
  hap_names.json        outputs of the reference's get_hap_name() (misc.cpp:58-87)
  kseq_vectors.json     small FASTA / FASTQ texts, well-formed and malformed, with the records the REFERENCE's
                        own kseq.h returns for them (kseq_read loop of ILP_index.cpp:313-328) and digests
                        for test/CHM13_reads.fq.gz
  counters.json         NOT regenerated here: the counters the reference logged on its own fixtures,
                        recorded in SURVEY.md section 8(c) (the ILP_index.cpp TU needs gurobi_c++.h,
                        absent from this image, so it cannot be rebuilt)
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    R = O.ref()
    rnd = random.Random(20251003)
    vec = []
    for n in list(range(0, 49)) + [63, 64, 65, 127, 255]:
        for _ in range(3):
            b = bytes(rnd.choice(b"ACGT") for _ in range(n)) if rnd.random() < 0.7 else bytes(rnd.randrange(256) for _ in range(n))
            vec.append({"hex": b.hex(), "hash": str(R.ref_hash128_to_64(b, n))})
    json.dump(vec, open(os.path.join(HERE, "murmur_vectors.json"), "w"), indent=0)

    data = os.path.join(HERE, "data")
    g = O.ref_parse_gfa(os.path.join(data, "test.gfa"))
    out = {"test.gfa": {"seg_names": g.seg_names, "node_seq": [s.decode() for s in g.node_seq],
                         "adj": [sorted(a) for a in g.adj], "paths": g.paths, "hap_names": g.hap_names}}
    g = O.ref_parse_gfa(os.path.join(data, "MHC_4.gfa.gz"))
    A = g.arrays()
    adj_sorted = [sorted(a) for a in g.adj]
    out["MHC_4.gfa.gz"] = {
        "n_vtx": g.n_vtx, "n_edges": int(A["adj_off"][-1]), "hap_names": g.hap_names,
        "walk_len": [len(p) for p in g.paths],
        "sha256_seq": hashlib.sha256(A["seq_concat"]).hexdigest(),
        "sha256_seq_off": hashlib.sha256(A["seq_off"].tobytes()).hexdigest(),
        "sha256_adj_sorted": hashlib.sha256(json.dumps(adj_sorted).encode()).hexdigest(),
        "sha256_walk_vtx": hashlib.sha256(A["walk_vtx"].tobytes()).hexdigest(),
    }
    json.dump(out, open(os.path.join(HERE, "gfa_flatten.json"), "w"))

    names = []
    for gfa, rd in [("test/MHC_4.gfa.gz", "test/CHM13_reads.fq.gz"), ("a.gfa", "r.fa"), ("/x/y/z.v1.gfa", "../q/reads.fastq.gz"),
                    ("graph", "reads"), ("dir.d/graph.gfa", "dir.e/reads")]:
        names.append({"gfa": gfa, "reads": rd, "name": O.ref_hap_name(gfa, rd)})
    json.dump(names, open(os.path.join(HERE, "hap_names.json"), "w"), indent=1)

    # kseq: texts + what the reference's reader returns
    import tempfile
    texts = [
        b">a\nACGT\n>b desc\nAC\nGT\n\nTT\n", b"@r\nACGT\n+\nIIII\n@s x\nAC\nGT\n+s\nII\nII\n",
        b"@r0\r\nAC\r\nGT\r\n+r0\r\n!>\r\n!#\r\n@r1\r\nNatNT\r\n+r1\r\nI~>I~\r\n",
        b"@r5 c\n\n+\n@r6\nAC\n+\nII\n",                     # empty sequence: its quality loop eats the next header, then -2
        b"@a\nACGT\n+\nII\n@b\nAC\n+\nII\n",                 # short quality: reading stops
        b"@a\nACGT\n+\n@III\n@b\nAC\n+\n+I\n",               # quality lines that start with '@' and '+'
        b"junk @x >y\n>a\nAC GT\tA\n>b\n\r\nAC\n",           # header bytes inside a line, white space kept, a CR-only line
        b">a\nACGT", b"@a\nACGT\n+", b"@a\nACGT\n+\n", b">\nAC\n> x\nGG\n", b"", b"\n\n", b">a\n>b\n>c\nA\n",
        b">a\nAC\n+\nII\n>b\nGG\n", b"@a\nAC\r\n+\r\nI\r\n",
        # kseq strips ONE trailing CR from the accumulated string after every line: two CRs and an empty line lose both (found by
        # fuzz_reads_reader at seed 9507: the host reader's quality length kept one)
        b"@a\nACGTA\n+\nIIIII\r\r\n\n@b\nAC\n+\nII\n", b"@a\nACGTA\n+\nIII\r\r\nII\n@b\nAC\n+\nII\n", b"@a\nACG\n+\n\r\r\r\n\r\nI\n@b\nA\n+\nI\n",
        b">a\nAC\r\r\n\r\n\nGT\n>b\n\r\r\nA\n",
    ]
    rnd2 = random.Random(7)
    for _ in range(40):                                            # random mixtures of the same ingredients
        parts = []
        for i in range(rnd2.randrange(1, 6)):
            L = rnd2.choice([0, 1, 3, rnd2.randrange(1, 90)])
            seq = bytes(rnd2.choice(b"ACGTNacgt") for _ in range(L))
            w = rnd2.randrange(3, 40)
            lines = [seq[j:j + w] for j in range(0, L, w)] or ([b""] if rnd2.random() < 0.5 else [])
            nl = b"\r\n" if rnd2.random() < 0.2 else b"\n"
            if rnd2.random() < 0.5:
                q = bytes(rnd2.choice(b"@+>I5#") for _ in range(L if rnd2.random() < 0.8 else max(L - 1, 0)))
                parts.append(nl.join([b"@n%d c" % i] + lines + [b"+"] + [q[j:j + w] for j in range(0, len(q), w)]) + nl)
            else:
                parts.append(nl.join([b">n%d" % i] + lines) + nl)
        t = b"".join(parts)
        if rnd2.random() < 0.3:
            t = t[:rnd2.randrange(0, len(t) + 1)]
        texts.append(t)
    kv = []
    with tempfile.TemporaryDirectory() as td:
        for t in texts:
            fn = os.path.join(td, "k.fq")
            open(fn, "wb").write(t)
            kv.append({"text_hex": t.hex(), "records": [[a.decode("latin1"), b.hex()] for a, b in O.ref_read_reads(fn)]})
    # kseq reads its stream in blocks of 65 536 bytes (kseq.h:242) and calls a stream ended when a block came back short -- so what it
    # makes of a file's very last bytes (a bare header character, a lone CR, a '+' line without quality) depends on whether
    # the file's size is a multiple of 65 536.  Texts of 65 535 / 65 536 / 65 537 / 131 072 bytes with such ends: the padding
    # is a first record with a name of the length needed, then `count` copies of an 8-byte record, stored run-length coded; of the records only count, digest and the last two.
    kb = []
    unit = b">p\nACGT\n"
    with tempfile.TemporaryDirectory() as td:
        for tail in (b">", b"@", b">x", b">x\n", b">x\nAC\n\r", b">x\nAC\n\rG", b"@x\nAC\n+", b"@x\nAC\n+\n", b"@x\nAC\n+\nI", b"@x\n\n+\n", b">x\nAC\n\n", b">x\nAC"):
            for size in (65535, 65536, 65537, 131072):
                n_unit, rest = divmod(size - len(tail), len(unit))
                n_unit, rest = n_unit - 1, rest + len(unit)           # 8 .. 15 bytes of padding: a first record whose name fills them
                head = b">" + b"h" * (rest - 4) + b"\nA\n"
                t = head + unit * n_unit + tail
                assert len(t) == size
                fn = os.path.join(td, "k.fa")
                open(fn, "wb").write(t)
                recs_ = O.ref_read_reads(fn)
                kb.append({"head_hex": head.hex(), "unit_hex": unit.hex(), "count": n_unit, "tail_hex": tail.hex(), "size": size,
                           "n_records": len(recs_), "sha256_seqs": hashlib.sha256(b"\0".join(b for _, b in recs_)).hexdigest(),
                           "last_records": [[a.decode("latin1"), b.hex()] for a, b in recs_[-2:]]})
    recs = O.ref_read_reads(os.path.join(data, "CHM13_reads.fq.gz"))
    chm = {"n": len(recs), "sha256_names": hashlib.sha256(b"\0".join(a for a, _ in recs)).hexdigest(),
           "sha256_seqs": hashlib.sha256(b"\0".join(b for _, b in recs)).hexdigest()}
    json.dump({"texts": kv, "block_boundary": kb, "CHM13_reads.fq.gz": chm}, open(os.path.join(HERE, "kseq_vectors.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
