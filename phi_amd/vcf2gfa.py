#!/usr/bin/env python3
"""vcf2gfa -- phased VCF + reference FASTA -> acyclic GFA 1.1 with one W-line per haplotype: the input route of
the reference's `vcf2gfa.py`, without its external tools.

The reference's script (vcf2gfa.py:27-64) shells out to `vg construct` / `vg gbwt` / `gfa2gbwt -m 30`, none of which
exist in this image, and writes the GFA to stdout; same flags here (`-v/--vcf`, `-r/--ref`, stdout), own construction:

  * records of the VCF that overlap or touch are merged into one SITE; the alleles of a site are the DISTINCT sequences
    the haplotypes (reference + two per sample, by the phased GT) spell over the site's span -- so overlapping records,
    multi-allelic records and a haplotype that carries two conflicting records (the second is ignored, as it cannot be
    applied) all come out as plain bubbles, and every allele is on at least one walk;
  * the backbone between sites and every allele are chopped into segments of at most 30 bp (`gfa2gbwt -m 30`,
    vcf2gfa.py:53; data/chop_graph.sh:62), segment names are 1, 2, 3, .. in topological order;
  * links `L a + b + 0M`; walks `W <sample> <hap> <contig> 0 <len> >a>b..`: `REF 0` for the reference (the pipeline
    renames the contig to REF#0, vcf2gfa.py:46-47), `<sample> 1` / `<sample> 2` for the two GT columns.

What `phi_gfa_read` / the reference's reader need (forward strands, walks along links, no cycles) holds by construction.
"""
import argparse
import gzip
import sys

CHOP = 30


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def read_fasta_single(path):
    name, chunks = None, []
    with _open(path) as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    raise ValueError("the reference FASTA holds more than one record; vcf2gfa handles one contig")
                name = line[1:].split()[0].decode() if len(line) > 1 else ""
            else:
                chunks.append(line.strip())
    if name is None:
        raise ValueError("no FASTA record in the reference file")
    return name, b"".join(chunks)


def read_vcf(path, ref_seq, contig=None, warn=None):
    """-> (sample names, records, ploidy per sample); record = (start0, end0, [alt bytes...], [(a1, a2) per sample])
    sorted by start.  Records of another contig than the first one seen (or `contig`), and records whose REF column is
    not what the FASTA holds at POS, are skipped with a warning: a VCF of another assembly must not be applied silently.
    ploidy[s] = the largest number of GT alleles any record gives sample s (1 for haploid calls: no second walk then)."""
    warn = warn or (lambda msg: sys.stderr.write("[W::vcf2gfa] " + msg + "\n"))
    ref_len = len(ref_seq)
    samples, recs, ploidy = [], [], []
    n_other, n_mismatch = 0, 0
    with _open(path) as f:
        for line in f:
            if line.startswith(b"##"):
                continue
            cols = line.rstrip(b"\r\n").split(b"\t")
            if line.startswith(b"#CHROM"):
                samples = [c.decode() for c in cols[9:]]
                ploidy = [0] * len(samples)
                continue
            if len(cols) < 10:
                continue
            if contig is None:
                contig = cols[0]
            if cols[0] != contig:
                n_other += 1
                continue
            pos, ref, alts = int(cols[1]) - 1, cols[3].upper(), [a.upper() for a in cols[4].split(b",")]
            fmt = cols[8].split(b":")
            if b"GT" not in fmt or pos < 0 or pos + len(ref) > ref_len:
                continue
            if ref_seq[pos:pos + len(ref)] != ref:
                n_mismatch += 1
                continue
            gi = fmt.index(b"GT")
            ok = [a for a in alts if a and not a.startswith(b"<") and a != b"*" and b"[" not in a and b"]" not in a]
            if len(ok) != len(alts):                           # symbolic / breakend / spanning-deletion alleles: not a sequence
                continue
            gts = []
            for si, c in enumerate(cols[9:9 + len(samples)]):
                g = c.split(b":")[gi].replace(b"/", b"|").split(b"|")
                ploidy[si] = max(ploidy[si], min(2, sum(1 for x in g if x != b".")))
                g = [int(x) if x.isdigit() else 0 for x in g] + [0, 0]
                gts.append((g[0], g[1]))
            recs.append((pos, pos + len(ref), alts, gts))
    if n_other:
        warn(f"{n_other} record(s) of other contigs than {contig.decode()} skipped (vcf2gfa handles one contig)")
    if n_mismatch:
        warn(f"{n_mismatch} record(s) skipped: their REF column is not what the FASTA holds at POS (another assembly?)")
    recs.sort(key=lambda r: (r[0], r[1]))
    return samples, recs, ploidy


class InputError(ValueError):
    """An input the construction cannot turn into a graph PHI accepts (the command line reports it, without a traceback)."""


def build(ref_seq, samples, recs, ploidy=None):
    """-> (segments [bytes], links set[(a, b)], walks [(sample, hap, [segment ids])]); ids are 0-based here.  A walk is made
    for every GT column a sample actually has (ploidy: 2 unless given): a haploid call gives one walk, not a second one that
    would merely repeat the reference."""
    n_hap = 1 + 2 * len(samples)                               # haplotype 0 = reference, then (sample, GT column)
    ploidy = ploidy if ploidy is not None else [2] * len(samples)
    # sites: maximal runs of records that overlap or touch
    sites, cur = [], None
    for r in recs:
        if cur is not None and r[0] <= cur[1]:
            cur[1] = max(cur[1], r[1]); cur[2].append(r)
        else:
            cur = [r[0], r[1], [r]]
            sites.append(cur)
    segs, links, walks = [], set(), [[] for _ in range(n_hap)]
    tails = [None] * n_hap                                     # last segment of every walk so far

    def add_unit(seq, haps):
        """chop seq into segments, append them to the walks of `haps`, link them up"""
        first = len(segs)
        for a in range(0, len(seq), CHOP):
            segs.append(seq[a:a + CHOP])
        ids = list(range(first, len(segs)))
        for i in range(len(ids) - 1):
            links.add((ids[i], ids[i + 1]))
        for h in haps:
            if ids:
                if tails[h] is not None:
                    links.add((tails[h], ids[0]))
                walks[h].extend(ids)
                tails[h] = ids[-1]

    pos = 0
    all_haps = list(range(n_hap))
    for s, e, rs in sites:
        # the sequence every haplotype spells over [s, e)
        spelled = {}
        for h in all_haps:
            if h == 0:
                seq = ref_seq[s:e]
            else:
                smp, col = divmod(h - 1, 2)
                out, at = [], s
                for (rs0, re0, alts, gts) in rs:
                    a = gts[smp][col]
                    if a <= 0 or a > len(alts) or rs0 < at:    # reference allele, or in conflict with a record already applied
                        continue
                    out.append(ref_seq[at:rs0]); out.append(alts[a - 1]); at = re0
                out.append(ref_seq[at:e])
                seq = b"".join(out)
            spelled.setdefault(seq, []).append(h)
        if len(spelled) == 1:
            continue                                           # nobody differs here: stays backbone
        if s <= pos and pos == 0:
            raise InputError("a variant at the first base of the contig would leave the graph without a single source vertex "
                             "(PHI's walks must start at one): trim the record or pad the reference by a base")
        add_unit(ref_seq[pos:s], all_haps)                     # backbone up to the site (>= 1 base: sites do not touch)
        for seq in sorted(spelled, key=lambda q: (q != ref_seq[s:e], q)):     # reference allele first: ids in a stable order
            add_unit(seq, spelled[seq])
        pos = e
    if pos >= len(ref_seq):
        raise InputError("a variant at the last base of the contig would leave the graph without a single sink vertex: "
                         "trim the record or pad the reference by a base")
    add_unit(ref_seq[pos:], all_haps)
    names = [("REF", 0)] + [(smp, col + 1) for smp in samples for col in range(2)]
    keep = [0] + [1 + 2 * si + col for si in range(len(samples)) for col in range(2) if col < max(1, ploidy[si])]
    return segs, links, [(names[h][0], names[h][1], walks[h]) for h in keep]


def write_gfa(out, contig, segs, links, walks):
    out.write(b"H\tVN:Z:1.1\n")
    for i, s in enumerate(segs):
        out.write(b"S\t%d\t%s\n" % (i + 1, s))
    for a, b in sorted(links):
        out.write(b"L\t%d\t+\t%d\t+\t0M\n" % (a + 1, b + 1))
    for smp, hap, ids in walks:
        n = sum(len(segs[i]) for i in ids)
        out.write(b"W\t%s\t%d\t%s\t0\t%d\t%s\n" % (smp.encode(), hap, contig.encode(), n, b"".join(b">%d" % (i + 1) for i in ids)))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Generate GFA from VCF and FASTA/FA files.")
    ap.add_argument("-v", "--vcf", required=True, help="Input VCF file (can be gzipped).")
    ap.add_argument("-r", "--ref", required=True, help="Input reference FASTA/FA file (can be gzipped).")
    args = ap.parse_args(argv)
    try:
        _, ref_seq = read_fasta_single(args.ref)
        ref_seq = ref_seq.upper()
        samples, recs, ploidy = read_vcf(args.vcf, ref_seq)
        segs, links, walks = build(ref_seq, samples, recs, ploidy)
    except (InputError, ValueError, OSError) as e:
        sys.stderr.write(f"[E::vcf2gfa] {e}\n")
        return 1
    write_gfa(sys.stdout.buffer, "REF#0", segs, links, walks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
