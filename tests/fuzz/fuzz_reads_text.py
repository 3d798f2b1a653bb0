"""Randomised comparison of the device-side record splitter (phi_add_reads_text, phi_amd/csrc/reads_text.hip) + the host
reader for what it hands back, with the reference's own kseq.h (oracle/_ref where /root/reference exists; else the oracle's
restatement of it): random FASTA / FASTQ texts, regular and with every anomaly a reads file can hold (CRLF, wrapped FASTQ,
'+' lines in FASTA, empty lines and reads, text before the first header, header characters inside lines, truncation),
cut into calls of random sizes, with device buffers (and carry capacity) small enough that the carry, the cutting of long
calls and the "record longer than the buffers" case all happen.  Checked: the records (device + host) are kseq's, and the
context counted exactly the device's share.
Usage (GPU box): python tests/fuzz/fuzz_reads_text.py SEED SECONDS     -- not collected by pytest."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phi_amd
from oracle import oracle as O
from phi_amd import ilp_index as H
import test_gpu_parity as T                      # noqa: F401  (conftest-free helpers)
from graphgen import random_graph


def records(bases, off):
    raw = bytes(bases)
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def random_text(rng):
    fastq = rng.random() < 0.5
    nl = b"\r\n" if rng.random() < 0.08 else b"\n"
    out = []
    for i in range(int(rng.integers(1, 60))):
        L = int(rng.choice([0, 1, 5, 31, 150, int(rng.integers(1, 700))], p=[0.03, 0.05, 0.05, 0.07, 0.4, 0.4]))
        seq = bytes(rng.choice(list(b"ACGTNacgtn"), size=L).tolist())
        name = b"r%d" % i + (b" some comment" if rng.random() < 0.3 else b"")
        if fastq:
            qual = bytes(rng.choice(list(b"@+>I5#!~"), size=L).tolist())
            if rng.random() < 0.04 and L > 3:
                w = int(rng.integers(1, L))
                rec = [b"@" + name] + [seq[j:j + w] for j in range(0, L, w)] + [b"+"] + [qual[j:j + w] for j in range(0, L, w)]
            else:
                rec = [b"@" + name, seq, b"+" + (name if rng.random() < 0.2 else b""), qual]
        else:
            w = int(rng.choice([0, 60, 80, int(rng.integers(1, 90))]))
            rec = [b">" + name] + ([seq[j:j + w] for j in range(0, L, w)] if w and L else [seq])
            if rng.random() < 0.03:
                rec.insert(1 + int(rng.integers(0, len(rec))), b"")
        out.append(nl.join(rec) + nl)
    txt = b"".join(out)
    if rng.random() < 0.15 and txt:
        mode = int(rng.integers(0, 6))
        cut = int(rng.integers(0, len(txt) + 1))
        txt = [txt[:cut], txt[:cut] + b"junk @x >y\n" + txt[cut:], txt[:cut] + b"AC GT\tAC\n" + txt[cut:], txt[:cut] + b"\r\n" + txt[cut:],
               txt[:cut] + b"+\n" + txt[cut:], b"leading junk\n" + txt][mode]
    if txt.endswith(nl) and rng.random() < 0.3:
        txt = txt[:-len(nl)]
    return txt


def main():
    rng = np.random.default_rng(int(sys.argv[1]))
    t_end = time.time() + float(sys.argv[2])
    g = random_graph(rng, n_sites=6, n_walks=3, seg_len=(4, 12), alt_len=(1, 4), p_del=0.0)
    ctx = phi_amd.Context(0)
    ctx.set_params(k=5, w=3, threshold=1.0, recombination=10)
    T._set_graph(ctx, g)
    n = n_irr = n_dev = n_parked = 0
    from phi_amd.context import TextPark
    park = TextPark(0)
    with tempfile.TemporaryDirectory() as td:
        while time.time() < t_end:
            txt = random_text(rng)
            if not txt:
                continue
            path = os.path.join(td, "r.fq")
            with open(path, "wb") as f:
                f.write(txt)
            want = [b for _, b in (O.ref_read_reads(path) if O.ref_available() else O.read_reads(path))]
            call = int(rng.choice([len(txt), int(rng.integers(1, 64)), int(rng.integers(64, 4000))]))
            os.environ["PHI_TEXT_CARRY"] = str(int(rng.choice([64, 256, 4096, 1 << 20])))
            ctx.reset_reads()
            ctx.reads_text_begin(max(64, call))
            got, irregular, rest_at = [], False, len(txt)
            parked_mode = rng.random() < 0.5              # (half the texts: pieces through device memory first, phi_text_park_*)
            for i in range(0, len(txt), call):
                if parked_mode and rng.random() < 0.7:
                    idx = park.add(txt[i:i + call])
                    irr = ctx.add_reads_text_parked(park, idx)
                    park.release(idx)
                    n_parked += 1
                else:
                    irr = ctx.add_reads_text(txt[i:i + call])
                if irr:
                    irregular, rest_at = True, i + call
                    break
                got += records(*ctx.reads_text_last_batch())
            pending, taken = ctx.reads_text_end()
            n_taken = len(got)
            st = ctx.reads_stats()
            assert st["n_reads"] == n_taken and st["n_bases"] == sum(map(len, got)), (txt[:400], call)
            got += records(*H.reads_of_text(pending, [txt[rest_at:]] if rest_at < len(txt) else [], stream_offset=taken))
            assert got == want, (call, os.environ["PHI_TEXT_CARRY"], irregular, txt[:1500])
            n += 1; n_irr += irregular; n_dev += n_taken > 0
    park.close()
    ctx.close()
    print("fuzz ok:", n, "texts,", n_irr, "went irregular,", n_dev, "with records taken by the device;", n_parked, "pieces through the text park")


if __name__ == "__main__":
    main()
