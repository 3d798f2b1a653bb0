"""The VCF input route: phi_amd/vcf2gfa.py replaces the reference's vcf2gfa.py (a wrapper
around vg / gfa2gbwt, absent here) by its own construction.  Checked on the reference's own fixtures: the VCF and the
reference FASTA under test/ describe the haplotypes of test/MHC_4.gfa.gz, so the walks of the generated graph must
spell the sequences of that graph's walks; and on small hand-made VCFs with overlapping, multi-allelic and conflicting
records against a direct application of the records to the reference.  Plus the log scrape of the evaluation harness."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT


def _walk_seqs(path):
    from phi_amd import ilp_index as H
    g = H.Graph(path)
    seq = bytes(g.seq_concat)
    out = {}
    for h, name in enumerate(g.hap_id2name):
        v = g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]]
        out[name] = b"".join(seq[g.seq_off[x]:g.seq_off[x + 1]] for x in v.tolist())
    return g, out


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "phi_amd", "csrc", "host"), "-s", os.path.join("..", "..", "libphi_host.so")])
    return True


def test_vcf_route_regenerates_the_haplotypes_of_the_reference_graph(built, oracle, tmp_path):
    out = tmp_path / "mhc4_from_vcf.gfa"
    with open(out, "wb") as f:
        subprocess.check_call([sys.executable, "-m", "phi_amd.vcf2gfa", "-v", os.path.join(DATA, "MHC_4.vcf.gz"),
                               "-r", os.path.join(DATA, "MHC-CHM13.0.fa.gz")], stdout=f, cwd=ROOT)
    g_ref, w_ref = _walk_seqs(os.path.join(DATA, "MHC_4.gfa.gz"))
    g, w = _walk_seqs(str(out))
    assert list(w) == ["REF.0", "HG002.1", "HG002.2", "HG005.1", "HG005.2"]
    # the reference walk and HG002.1 end at the sink of MHC_4.gfa: identical sequences
    assert w["REF.0"] == w_ref["CHM13.0"].upper() and w["HG002.1"] == w_ref["HG002.1"].upper()
    # the other three walks of MHC_4.gfa stop one vertex (21 bp) before the sink; the VCF cannot
    # say so: same sequence, 21 bases longer
    for name in ("HG002.2", "HG005.1", "HG005.2"):
        assert len(w[name]) == len(w_ref[name]) + 21 and w[name].startswith(w_ref[name].upper())
    # what the readers need: segments of at most 30 bp, forward links that the walks follow, one source, one sink, no cycle
    lens = np.diff(g.seq_off)
    assert lens.min() >= 1 and lens.max() <= 30
    src = np.repeat(np.arange(g.n_vtx), np.diff(g.adj_off))
    assert np.all(g.top_order_map[g.adj] > g.top_order_map[src])
    indeg = np.bincount(g.adj, minlength=g.n_vtx)
    assert int((indeg == 0).sum()) == 1 and int((np.diff(g.adj_off) == 0).sum()) == 1
    used = np.zeros(g.n_vtx, bool)
    used[g.walk_vtx] = True
    assert used.all()                                          # every allele is on some walk
    # the oracle's restatement of the reference's parser reads the same graph
    og = oracle.parse_gfa(str(out))
    assert og.n_vtx == g.n_vtx and [len(p) for p in og.paths] == np.diff(g.walk_off).tolist()


def _consensus(ref, recs, smp, col):
    """apply the ALT alleles haplotype (smp, col) carries, left to right, skipping one that overlaps an applied one"""
    out, at = [], 0
    for (pos, r, alts, gts) in recs:
        a = gts[smp][col]
        if a <= 0 or pos < at:
            continue
        out.append(ref[at:pos]); out.append(alts[a - 1]); at = pos + len(r)
    out.append(ref[at:])
    return b"".join(out)


def test_vcf_route_on_overlapping_multiallelic_and_conflicting_records(built, tmp_path):
    from phi_amd import vcf2gfa
    rng = np.random.default_rng(3)
    for case in range(30):
        ref = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(300, 900))).tolist())
        n_s = int(rng.integers(1, 4))
        recs, pos = [], int(rng.integers(2, 20))
        while pos < len(ref) - 60:
            kind = rng.random()
            rl = 1 if kind < 0.5 else int(rng.integers(1, 40))
            r = ref[pos:pos + rl]
            alts = []
            for _ in range(int(rng.integers(1, 4))):
                al = int(rng.integers(1, 45)) if rng.random() < 0.6 else 1
                a = r[:1] + bytes(rng.choice(list(b"ACGT"), size=al - 1).tolist()) if rng.random() < 0.7 else bytes(rng.choice(list(b"ACGT"), size=al).tolist())
                if a != r and a not in alts:
                    alts.append(a)
            if alts:
                gts = [(int(rng.integers(0, len(alts) + 1)), int(rng.integers(0, len(alts) + 1))) for _ in range(n_s)]
                recs.append((pos, r, alts, gts))
            # next record: sometimes overlapping or touching this one, sometimes at the same position
            step = rng.random()
            pos += 0 if step < 0.1 else (int(rng.integers(1, max(2, rl))) if step < 0.35 else rl + int(rng.integers(0, 60)))
        vcf = tmp_path / f"c{case}.vcf"
        with open(vcf, "wb") as f:
            f.write(b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%d" % i for i in range(n_s)) + b"\n")
            for (p, r, alts, gts) in recs:
                f.write(b"chr\t%d\t.\t%s\t%s\t60\t.\t.\tGT\t%s\n" % (p + 1, r, b",".join(alts), b"\t".join(b"%d|%d" % g for g in gts)))
        fa = tmp_path / f"c{case}.fa"
        fa.write_bytes(b">chr\n" + b"\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + b"\n")
        _, ref_seq = vcf2gfa.read_fasta_single(str(fa))
        samples, parsed, ploidy = vcf2gfa.read_vcf(str(vcf), ref_seq)
        assert ploidy == [2] * n_s
        segs, links, walks = vcf2gfa.build(ref_seq, samples, parsed, ploidy)
        srecs = sorted(recs, key=lambda x: (x[0], x[0] + len(x[1])))
        want = [ref] + [_consensus(ref, srecs, s, c) for s in range(n_s) for c in range(2)]
        got = [b"".join(segs[i] for i in ids) for _, _, ids in walks]
        assert got == want, case
        assert all(1 <= len(s) <= 30 for s in segs)
        for _, _, ids in walks:
            assert all((a, b) in links for a, b in zip(ids[:-1], ids[1:])) and all(b > a for a, b in zip(ids[:-1], ids[1:]))
        # through the file and the reader
        buf = io.BytesIO()
        vcf2gfa.write_gfa(buf, "REF#0", segs, links, walks)
        p = tmp_path / f"c{case}.gfa"
        p.write_bytes(buf.getvalue())
        _, w = _walk_seqs(str(p))
        assert list(w.values()) == want


def test_vcf_route_refuses_or_skips_what_it_cannot_apply(built, tmp_path, capsys):
    """Records of another contig, and records whose REF is not what the FASTA holds, are skipped with a warning (a VCF of
    another assembly is not applied silently); a haploid GT column gives ONE walk for that sample; a variant at the first
    or last base of the contig is a clear error of the command line, not a traceback."""
    from phi_amd import vcf2gfa
    ref = b"ACGTTGCAAGGCTTAACCGGATCGATCGGCTAAGCTTAGGCTA" * 3
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">chr\n" + ref + b"\n")
    hdr = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tD\tH\n"
    vcf = tmp_path / "v.vcf"
    vcf.write_bytes(hdr + b"chr\t11\t.\t" + ref[10:11] + b"\tT\t60\t.\t.\tGT\t0|1\t1\n"        # D diploid, H haploid
                    b"other\t20\t.\tA\tC\t60\t.\t.\tGT\t1|1\t1\n"                               # another contig
                    b"chr\t31\t.\tN\tC\t60\t.\t.\tGT\t1|1\t1\n")                                # REF is not what the FASTA holds
    _, ref_seq = vcf2gfa.read_fasta_single(str(fa))
    msgs = []
    samples, recs, ploidy = vcf2gfa.read_vcf(str(vcf), ref_seq, warn=msgs.append)
    assert len(recs) == 1 and ploidy == [2, 1]
    assert any("other contigs" in m for m in msgs) and any("REF column" in m for m in msgs)
    segs, links, walks = vcf2gfa.build(ref_seq, samples, recs, ploidy)
    assert [(s_, h) for s_, h, _ in walks] == [("REF", 0), ("D", 1), ("D", 2), ("H", 1)]
    alt = ref[:10] + b"T" + ref[11:]
    assert [b"".join(segs[i] for i in ids) for _, _, ids in walks] == [ref, ref, alt, alt]
    # a variant at the very first base: the command line says what is wrong and returns 1
    first = tmp_path / "first.vcf"
    first.write_bytes(hdr + b"chr\t1\t.\t" + ref[0:1] + b"\tT\t60\t.\t.\tGT\t0|1\t1\n")
    assert vcf2gfa.main(["-v", str(first), "-r", str(fa)]) == 1
    assert "first base of the contig" in capsys.readouterr().err
    last = tmp_path / "last.vcf"
    last.write_bytes(hdr + b"chr\t%d\t.\t" % len(ref) + ref[-1:] + b"\tG\t60\t.\t.\tGT\t0|1\t1\n")
    assert vcf2gfa.main(["-v", str(last), "-r", str(fa)]) == 1
    assert "last base of the contig" in capsys.readouterr().err


def test_eval_log_scrapes_what_the_harness_scrapes():
    from phi_amd import eval_log
    log = ("[M::main::0.139*1.02] Loaded graph from: test/MHC_4.gfa.gz\nNumber of Minimizers\nCHM13.0 : 471226\n"
           "[M::ILP_function::0.2*1.5] Indexed reads with spectrum size: 138834\n"
           "[M::ILP_function::0.2*1.5] Filtered/Retained Minimizers: 77.07/22.93%\n"
           "[M::ILP_function::0.2*1.5] 14.92% Minimizers are in ILP\nRecombination count: 0\n"
           "[M::main] Real time: 0.312 sec; CPU: 0.508 sec; Peak RSS: 0.421 GB\n")
    got = eval_log.parse_log(log)
    assert got == dict(recombination_count=0, real_time_s=0.312, peak_rss_gb=0.421, spectrum_size=138834,
                       pct_minimizers_in_ilp=14.92, pct_filtered=77.07, pct_retained=22.93)
    assert eval_log.parse_log("nothing here")["spectrum_size"] is None
    # the banded edit distance against a plain DP
    rng = np.random.default_rng(1)

    def plain(a, b):
        d = list(range(len(b) + 1))
        for i in range(1, len(a) + 1):
            nd = [i] + [0] * len(b)
            for j in range(1, len(b) + 1):
                nd[j] = min(d[j] + 1, nd[j - 1] + 1, d[j - 1] + (a[i - 1] != b[j - 1]))
            d = nd
        return d[-1]
    for _ in range(30):
        a = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 200))).tolist())
        b = bytearray(a)
        for _ in range(int(rng.integers(0, 12))):
            p = int(rng.integers(0, max(1, len(b))))
            op = rng.random()
            if op < 0.4 and b:
                b[p] = int(rng.choice(list(b"ACGT")))
            elif op < 0.7 and b:
                del b[p]
            else:
                b.insert(p, int(rng.choice(list(b"ACGT"))))
        assert eval_log.edit_distance(a, bytes(b), band=4) == plain(a, bytes(b))
