"""CPU tests: the host-side readers under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer
(`make -C phi_amd/csrc/host sanitize`: the reader sources compiled into a self-test driver, host_selftest.cpp).

The readers are multi-threaded -- slices of the GFA text, pieces of the W-lines, the block-gzip pool, parallel preads --
and parse untrusted text.  The sanitized drivers read the reference's fixtures, block-gzip variants of them and a
seed-fixed slice of the inputs the fuzz harnesses make (tests/fuzz/fuzz_gfa_reader.py, fuzz_reads_reader.py: shuffled
lines, CRLF, truncated and malformed text), must report nothing, and must print the checksums the plain build prints."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT

SAN = os.path.join(ROOT, "build", "sanitize")


@pytest.fixture(scope="module")
def drivers():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "phi_amd", "csrc", "host"), "sanitize", "-s"])
    return {n: os.path.join(SAN, "host_selftest_" + n) for n in ("plain", "asan", "tsan")}


def _inputs(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphgen import random_graph
    import fuzz_gfa_reader as FG
    from test_cpu_abi_host import _bgzf
    rng = np.random.default_rng(4242)
    gfas, reads = [os.path.join(DATA, "test.gfa"), os.path.join(DATA, "MHC_4.gfa.gz")], [os.path.join(DATA, "read.fa"), os.path.join(DATA, "CHM13_reads.fq.gz")]
    # the reference's fixtures as block gzip (the inflate pool) and as plain text (the mapped path, W-lines in pieces)
    raw = gzip.open(os.path.join(DATA, "MHC_4.gfa.gz"), "rb").read()
    (tmp_path / "mhc4.bgzf.gfa.gz").write_bytes(_bgzf(raw, rng=rng))
    (tmp_path / "mhc4.gfa").write_bytes(raw)
    gfas += [str(tmp_path / "mhc4.bgzf.gfa.gz"), str(tmp_path / "mhc4.gfa")]
    fq = gzip.open(os.path.join(DATA, "CHM13_reads.fq.gz"), "rb").read()
    (tmp_path / "chm13.bgzf.fq.gz").write_bytes(_bgzf(fq, rng=rng))
    (tmp_path / "chm13.fq").write_bytes(fq)
    (tmp_path / "chm13.cut.fq.gz").write_bytes(gzip.compress(fq)[:300_000])        # truncated: what inflates, as gzread
    bad = bytearray(_bgzf(fq[:500_000], block=60_000)); bad[len(bad) // 2] ^= 0xFF
    (tmp_path / "chm13.bad.fq.gz").write_bytes(bytes(bad))                          # corrupt: an error from every reader
    reads += [str(tmp_path / n) for n in ("chm13.bgzf.fq.gz", "chm13.fq", "chm13.cut.fq.gz", "chm13.bad.fq.gz")]
    # a slice of the GFA fuzz inputs
    for i in range(40):
        g = random_graph(rng, n_sites=int(rng.integers(1, 12)), n_walks=int(rng.integers(1, 7)), seg_len=(1, int(rng.integers(2, 40))),
                         alt_len=(1, int(rng.integers(2, 12))), p_del=float(rng.choice([0, 0.3])))
        p = str(tmp_path / (f"f{i}.gfa.gz" if rng.random() < 0.3 else f"f{i}.gfa"))
        FG.write_random_gfa(rng, g, p, rng.random() < 0.25)
        gfas.append(p)
    # a slice of the reads fuzz inputs (the generator of fuzz_reads_reader.py, restated: that file is a script)
    for i in range(60):
        fastq = rng.random() < 0.5
        nl = b"\r\n" if rng.random() < 0.2 else b"\n"
        out = []
        for r in range(int(rng.integers(0, 30))):
            L = int(rng.choice([0, 1, 5, int(rng.integers(1, 400))]))
            seq = bytes(rng.choice(list(b"ACGTNacgtn"), size=L).tolist())
            w = int(rng.integers(5, 90))
            name = b"r%d" % r + (b" some comment" if rng.random() < 0.3 else b"")
            lines = [seq[j:j + w] for j in range(0, L, w)] or ([b""] if rng.random() < 0.5 else [])
            if fastq:
                qual = bytes(rng.choice(list(b"@+>I5#!~"), size=L).tolist())
                rec = [b"@" + name] + lines + [b"+"] + [qual[j:j + w] for j in range(0, L, w)]
            else:
                rec = [b">" + name] + lines
            out.append(nl.join(rec) + nl)
        txt = b"".join(out)
        if rng.random() < 0.3 and txt:
            cut = int(rng.integers(0, len(txt) + 1))
            txt = [txt[:cut], txt[:cut] + b"junk @x >y\n" + txt[cut:], txt[:cut] + b"\r\n" + txt[cut:], txt[:cut] + b"+\n" + txt[cut:], b"leading junk\n" + txt][int(rng.integers(0, 5))]
        p = tmp_path / (f"r{i}.fq.gz" if rng.random() < 0.3 else f"r{i}.fq")
        p.write_bytes(gzip.compress(txt) if str(p).endswith(".gz") else txt)
        reads.append(str(p))
    return gfas, reads


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_readers_under_sanitizers(drivers, tmp_path, san):
    gfas, reads = _inputs(tmp_path)
    env = dict(os.environ, PHI_HOST_THREADS="6", PHI_GFA_PIECE="4096",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="exitcode=66:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    for mode, files in (("gfa", gfas), ("reads", reads)):
        want = subprocess.run([drivers["plain"], mode] + files, capture_output=True, text=True, env=env, timeout=600)
        assert want.returncode == 0, want.stdout[-2000:] + want.stderr[-2000:]
        got = subprocess.run([drivers[san], mode] + files, capture_output=True, text=True, env=env, timeout=900)
        assert got.returncode == 0, (san, mode, got.stdout[-1500:], got.stderr[-4000:])
        assert "Sanitizer" not in got.stderr and "runtime error" not in got.stderr, got.stderr[-4000:]
        assert got.stdout == want.stdout, (san, mode)
        lines = want.stdout.strip().splitlines()
        assert len(lines) == len(files)
        # the corrupt block-gzip file is an error for every reader, every other reads file is read
        if mode == "reads":
            assert sum(" error " in l for l in lines) == 1 and "chm13.bad.fq.gz error -1 (stream too)" in want.stdout
