#!/usr/bin/env python3
"""eval_log -- the fields the reference's evaluation harness scrapes from a PHI log.

data/postprocessing_2_MIQP.py:55-79 reads, per (sample, coverage) run, `Recombination count`, `Real time`, `Peak RSS`,
`Indexed reads with spectrum size`, `% Minimizers are in ILP` and `Filtered/Retained Minimizers` out of the stderr log
with the regular expressions below, and the edit distance of the output FASTA to a ground truth with edlib.  The log
lines of phi_amd/PHI keep those formats (phi_main.cpp); this module is the same scrape as a function and a small
command line, so that the harness' tables can be made from runs of this build:

    python -m phi_amd.eval_log run1.log [run2.log ...]            # CSV on stdout
    python -m phi_amd.eval_log --truth truth.fa --query out.fa run.log   # + edit distance (banded, exact within the band)
"""
import argparse
import csv
import re
import sys

# the expressions of data/postprocessing_2_MIQP.py:56, :60, :63, :70, :73, :77
PATTERNS = {
    "recombination_count": (r"Recombination count:\s+(\d+)", int),
    "real_time_s": (r"Real time:\s+(\d+\.\d+)\s+sec", float),
    "peak_rss_gb": (r"Peak RSS:\s+(\d+\.\d+)\s+GB", float),
    "spectrum_size": (r"Indexed reads with spectrum size:\s+(\d+)", int),
    "pct_minimizers_in_ilp": (r"(\d+\.\d+)% Minimizers are in ILP", float),
}
FILTERED = r"Filtered/Retained Minimizers:\s+(\d+\.\d+)/(\d+\.\d+)%"
FIELDS = list(PATTERNS) + ["pct_filtered", "pct_retained"]


def parse_log(text):
    """-> dict of the scraped fields (None where a line is missing, as the harness does)."""
    out = {}
    for key, (pat, conv) in PATTERNS.items():
        m = re.search(pat, text)
        out[key] = conv(m.group(1)) if m else None
    m = re.search(FILTERED, text)
    out["pct_filtered"] = float(m.group(1)) if m else None
    out["pct_retained"] = float(m.group(2)) if m else None
    return out


def read_fasta(path):
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        return b"".join(l.strip() for l in f if not l.startswith(b">")).upper()


def edit_distance(a, b, band=None):
    """Global (NW, unit costs) edit distance of two byte strings within a diagonal band: what edlib's NW mode returns
    when the true distance fits the band (the band doubles until it does).  O(len * band) time, numpy rows."""
    import numpy as np
    if len(a) < len(b):
        a, b = b, a
    n, m = len(a), len(b)
    band = band or max(64, (n - m) * 2 + 64)
    A, B = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
    while True:
        INF = 1 << 40
        w = 2 * band + 1
        prev = np.full(w, INF, np.int64)
        # row 0: D[0][j] = j for j in [0, band]
        prev[band:band + min(band, m) + 1] = np.arange(0, min(band, m) + 1)
        for i in range(1, n + 1):
            lo = i - band                                          # column of slot 0
            j = np.arange(lo, lo + w)
            valid = (j >= 0) & (j <= m)
            cur = np.full(w, INF, np.int64)
            # substitution / match: D[i-1][j-1] is slot s of prev (the band moves right by one per row)
            jj = np.clip(j - 1, 0, m - 1)
            sub = prev + (A[i - 1] != B[jj])
            sub[j < 1] = INF
            dele = np.concatenate((prev[1:], [INF])) + 1            # D[i-1][j]
            cur = np.minimum(sub, dele)
            cur[j == 0] = i
            cur[~valid] = INF
            # insertions along the row: D[i][j-1] + 1 (prefix minimum of cur[s] - s)
            base = cur - np.arange(w)
            cur = np.minimum(cur, np.minimum.accumulate(base) + np.arange(w))
            cur[~valid] = INF
            prev = cur
        d = int(prev[m - (n - band)]) if 0 <= m - (n - band) < w else INF
        if d <= band - (n - m) or band >= n:
            return d
        band *= 2


def main(argv=None):
    ap = argparse.ArgumentParser(description="Scrape PHI logs as data/postprocessing_2_MIQP.py does")
    ap.add_argument("logs", nargs="+")
    ap.add_argument("--truth", help="ground-truth FASTA (with --query: adds the edit distance)")
    ap.add_argument("--query", help="FASTA written by PHI")
    args = ap.parse_args(argv)
    w = csv.writer(sys.stdout)
    extra = ["edit_distance"] if args.truth and args.query else []
    w.writerow(["log"] + FIELDS + extra)
    for p in args.logs:
        row = parse_log(open(p, errors="replace").read())
        vals = [row[k] for k in FIELDS]
        if extra:
            vals.append(edit_distance(read_fasta(args.truth), read_fasta(args.query)))
        w.writerow([p] + vals)


if __name__ == "__main__":
    main()
