"""Randomised comparison of the reads readers (phi_reads_read and phi_reads_stream_* of libphi_host) with the
reference's own kseq.h (oracle/_ref, where /root/reference exists) and the oracle's restatement of it (oracle.read_reads) on random FASTA / FASTQ texts: multi-line records, CRLF, empty
lines, comments after the name, quality lines that start with '@' '+' '>', records without sequence, gzip.
Usage (CPU): python tests/fuzz/fuzz_reads_reader.py SEED SECONDS     -- not collected by pytest."""
import gzip
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from phi_amd import ilp_index as H


def main():
    rng = np.random.default_rng(int(sys.argv[1]))
    t_end = time.time() + float(sys.argv[2])
    n = 0
    with tempfile.TemporaryDirectory() as td:
        while time.time() < t_end:
            fastq = rng.random() < 0.5
            nl = b"\r\n" if rng.random() < 0.2 else b"\n"
            out = []
            for i in range(int(rng.integers(0, 30))):
                L = int(rng.choice([0, 1, 5, int(rng.integers(1, 400))]))
                seq = bytes(rng.choice(list(b"ACGTNacgtn"), size=L).tolist())
                w = int(rng.integers(5, 90))
                name = b"r%d" % i + (b" some comment" if rng.random() < 0.3 else b"")
                lines = [seq[j:j + w] for j in range(0, L, w)] or ([b""] if rng.random() < 0.5 else [])
                if fastq:
                    qual = bytes(rng.choice(list(b"@+>I5#!~"), size=L).tolist())
                    ql = [qual[j:j + w] for j in range(0, L, w)]
                    rec = [b"@" + name] + lines + [b"+" + (name if rng.random() < 0.2 else b"")] + ql
                else:
                    rec = [b">" + name] + lines
                if rng.random() < 0.1:
                    rec.append(b"")
                out.append(nl.join(rec) + nl)
            txt = b"".join(out)
            if rng.random() < 0.3:                 # malformed input: kseq's behaviour there is part of the contract
                mode = int(rng.integers(0, 6))
                cut = int(rng.integers(0, len(txt) + 1))
                if mode == 0: txt = txt[:cut]                                           # truncated file
                elif mode == 1: txt = txt[:cut] + b"junk @x >y\n" + txt[cut:]            # a header character inside a line
                elif mode == 2: txt = txt[:cut] + b"AC GT\tAC\n" + txt[cut:]             # white space inside a sequence
                elif mode == 3: txt = txt[:cut] + b"\r\n" + txt[cut:]                    # a CR-only line
                elif mode == 4: txt = txt[:cut] + b"+\n" + txt[cut:]                     # a stray '+'
                else: txt = b"leading junk\n" + txt
            if txt.endswith(nl) and rng.random() < 0.3:
                txt = txt[:-len(nl)]
            path = os.path.join(td, "r.fq.gz" if rng.random() < 0.3 else "r.fq")
            with (gzip.open(path, "wb") if path.endswith(".gz") else open(path, "wb")) as f:
                f.write(txt)
            exp = O.read_reads(path)
            if O.ref_available():                  # the reference's own kseq.h
                assert O.ref_read_reads(path) == [(a if isinstance(a, bytes) else a.encode(), b) for a, b in exp], txt[:1500]
            bases, off, names = H.read_reads(path)
            got = [(names[i], bytes(bases[off[i]:off[i + 1]])) for i in range(len(names))]
            assert got == [(a.decode() if isinstance(a, bytes) else a, b) for a, b in exp], txt[:1500]
            longest = max([len(b) for _, b in exp] + [1])
            cap = int(rng.choice([longest, longest + 7, 4096 + longest]))
            sb, sl = [], []
            for b, o in H.stream_reads(path, bases_cap=cap, reads_cap=int(rng.integers(1, 9))):
                sb.append(b); sl.append(np.diff(o))
            assert (np.concatenate(sl).tolist() if sl else []) == [len(b) for _, b in exp], txt[:1500]
            assert (np.concatenate(sb).tobytes() if sb else b"") == b"".join(b for _, b in exp), txt[:1500]
            n += 1
    print("fuzz ok:", n, "reads files")


if __name__ == "__main__":
    main()
