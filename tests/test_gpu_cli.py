"""GPU tests of the drop-in surface: the `PHI` command line (same flags / log lines / FASTA as the
reference's src/main.cpp) and the Python mirror of class ILP_index, on the reference's fixtures."""
import io
import json
import os
import re
import subprocess

import pytest

from conftest import DATA, GOLDEN, ROOT

pytestmark = pytest.mark.gpu

PHI = os.path.join(ROOT, "phi_amd", "PHI")


def _run_cli(args, tmp_path, env=None):
    if not os.path.exists(PHI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "phi_amd", "csrc", "host")])
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([PHI] + args, capture_output=True, text=True, cwd=str(tmp_path), timeout=300, env=e)


def test_cli_config1_logs_and_fasta(tmp_path):
    gold = json.load(open(os.path.join(GOLDEN, "counters.json")))["mhc4_chm13_k31_w25"]
    out = tmp_path / "CHM13.fa"
    r = _run_cli(["-t32", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz"), "-o", str(out)], tmp_path)
    assert r.returncode == 0, r.stderr
    log = r.stderr
    # the lines data/postprocessing_2_MIQP.py:55-79 scrapes, and the per-walk tables
    assert re.search(r"\[M::main::[\d.]+\*[\d.]+\] Loaded graph from: ", log)
    assert "Graph has 111805 vertices, 5 walks and read has 16401 reads" in log
    for name, n in zip(gold["hap_names"], gold["n_minimizers"]):
        assert f"{name} : {n}\n" in log
    assert f"Indexed reads with spectrum size: {gold['spectrum_size']}\n" in log
    for name, n in zip(gold["hap_names"], gold["n_anchors"]):
        assert f"{name} : {n}\n" in log
    assert f"Filtered/Retained Minimizers: {gold['filtered_retained_pct']}%\n" in log
    assert f"{gold['pct_in_model']}% Minimizers are in ILP\n" in log
    assert "QP model started" in log and "Using Mixed Integer Programming" in log
    assert "Recombination count: 0\n" in log
    m = re.search(r"Recombined haplotypes: >\(CHM13\.0,\[0,(\d+)\]\)\n", log)
    assert m
    assert re.search(r"Real time: [\d.]+ sec; CPU: [\d.]+ sec; Peak RSS: [\d.]+ GB", log)
    txt = out.read_text().split("\n")
    mm = re.match(r">MHC_4\.gfa_CHM13_reads\.fq LN:(\d+)$", txt[0])
    assert mm and int(mm.group(1)) == int(m.group(1)) + 1
    seq = "".join(txt[1:])
    assert len(seq) == int(mm.group(1)) and all(len(x) == 80 for x in txt[1:-2])
    assert f"Haplotype of size: {len(seq)} written to: {out}" in log


def test_cli_toy_and_python_mirror_agree(tmp_path):
    from phi_amd import ilp_index as H
    gfa, rd = os.path.join(DATA, "test.gfa"), os.path.join(DATA, "read.fa")
    out = tmp_path / "toy.fa"
    r = _run_cli(["-g", gfa, "-r", rd, "-o", str(out), "-k3", "-w2", "-q0", "-m0", "-R", "10"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "ILP model started" in r.stderr and "Using Integer Programming" in r.stderr
    assert "test_hap_4.4 : 5\n" in r.stderr and "50.00% Minimizers are in ILP" in r.stderr
    assert "Filtered/Retained Minimizers: 37.50/62.50%" in r.stderr
    cli_fa = out.read_text()
    assert cli_fa.startswith(">test_read LN:")
    # the Python mirror of main.cpp:114-140
    log = io.StringIO()
    idx = H.ILP_index(gfa, log=log)
    idx.read_gfa()
    idx.k_mer, idx.window, idx.recombination, idx.is_qclp, idx.is_mixed = 3, 2, 10, 0, False
    idx.hap_file = str(tmp_path / "toy_py.fa")
    idx.hap_name = H.get_hap_name(gfa, rd)
    reads = []
    idx.read_ip_reads(reads, rd)
    assert reads == [("test_read_1", b"ATCGATCATACTTACCATG")]
    res = idx.ILP_function(reads)
    assert res["objective"] == 4
    assert open(idx.hap_file).read() == cli_fa
    py_log = log.getvalue()
    for line in ("Number of Minimizers", "test_hap_1.0 : 10", "Indexed reads with spectrum size: 8", "Recombination count: 0"):
        assert line in py_log and line in r.stderr
    # -d1: the sharing histogram of ILP_index.cpp:591-604, same lines from both front ends
    r2 = _run_cli(["-g", gfa, "-r", rd, "-o", str(tmp_path / "toy_d.fa"), "-k3", "-w2", "-q0", "-m0", "-R", "10", "-d1"], tmp_path)
    assert r2.returncode == 0, r2.stderr
    assert "Shared fraction of unique kmers by haplotypes" in r2.stderr and "Shared fraction" not in r.stderr
    d_lines = [l for l in r2.stderr.splitlines() if l.startswith("[Haplotypes: ")]
    assert len(d_lines) == 5 and abs(sum(float(l.split(": ")[-1].rstrip("]")) for l in d_lines) - 1.0) < 1e-4
    log2 = io.StringIO()
    idx2 = H.ILP_index(gfa, log=log2)
    idx2.read_gfa()
    idx2.k_mer, idx2.window, idx2.recombination, idx2.is_qclp, idx2.is_mixed, idx2.debug = 3, 2, 10, 0, False, True
    idx2.hap_file = str(tmp_path / "toy_py_d.fa")
    idx2.hap_name = H.get_hap_name(gfa, rd)
    idx2.ILP_function([b"ATCGATCATACTTACCATG"])
    assert [l for l in log2.getvalue().splitlines() if l.startswith("[Haplotypes: ")] == d_lines


def test_cli_on_synthetic_files(tmp_path):
    """The synthetic stand-in written as GFA 1.1 (S/L/W, gz) + FASTQ (gz): the command line, the
    Python mirror on the same files and the array API on the generator's own arrays agree."""
    import numpy as np
    import phi_amd
    from phi_amd import ilp_index as H
    from phi_amd import synth
    gk, rk = synth.CONFIGS["tiny"]
    g = synth.make_graph(**gk)
    bases, off, truth = synth.make_reads(g, **rk)
    gfa, rd = str(tmp_path / "tiny.gfa.gz"), str(tmp_path / "tiny_reads.fq.gz")
    synth.write_gfa(g, gfa)
    synth.write_reads(bases, off, rd, fastq=True)
    out = tmp_path / "tiny.fa"
    r = _run_cli(["-t8", "-g", gfa, "-r", rd, "-o", str(out), "-R", "10"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert f"Graph has {g.n_vtx} vertices, {g.n_walks} walks and read has {len(off) - 1} reads" in r.stderr
    cli_fa = out.read_text()
    idx = H.ILP_index(gfa, log=io.StringIO())
    idx.read_gfa()
    idx.recombination = 10
    idx.hap_file = str(tmp_path / "tiny_py.fa")
    idx.hap_name = H.get_hap_name(gfa, rd)
    reads = []
    idx.read_ip_reads(reads, rd)
    assert len(reads) == len(off) - 1
    res = idx.ILP_function(reads)
    assert open(idx.hap_file).read() == cli_fa
    # the array API on the generator's arrays (other vertex numbering, same optimum)
    ctx = phi_amd.Context(0)
    ctx.set_params(k=31, w=25, threshold=1.0, recombination=10)
    A = g.arrays()
    ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    ctx.add_reads((bases, off))
    res2 = ctx.solve()
    ctx.close()
    assert res2["objective"] == res["objective"] and res2["spectrum_size"] == res["spectrum_size"]
    assert res2["optimal"] == 1


def test_cli_streams_reads_in_chunks(tmp_path):
    """The command line streams the reads file as text through three chunk buffers, the records being found
    on the device: with chunks of 20 kB (the fixture's 5.4 MB of FASTQ text then take ~270 of them, pinned from the
    second on) every log line and the FASTA are those of the single-chunk run."""
    args = ["-t8", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz")]
    one = _run_cli(args + ["-o", str(tmp_path / "one.fa")], tmp_path, env={"PHI_TIMING": "1"})
    many = _run_cli(args + ["-o", str(tmp_path / "many.fa")], tmp_path, env={"PHI_READ_CHUNK": "20000", "PHI_TIMING": "1"})
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    assert "main: 1 text chunk(s)" in one.stderr
    m = re.search(r"main: (\d+) text chunk\(s\) of up to 20000 bytes, pinned", many.stderr)
    assert m and int(m.group(1)) > 200
    # all of it through the device (a FASTQ file of whole records leaves nothing for the host reader)
    assert "; 0 bases through the host reader" in one.stderr and "; 0 bases through the host reader" in many.stderr

    def lines(log):
        keep = []
        for l in log.splitlines():
            if l.startswith("[phi timing]") or "Real time" in l or "CMD:" in l or "written to" in l:
                continue
            keep.append(re.sub(r"^\[M::[^\]]*\] ", "", l))
        return keep
    assert lines(one.stderr) == lines(many.stderr)
    assert "Graph has 111805 vertices, 5 walks and read has 16401 reads" in many.stderr
    assert (tmp_path / "one.fa").read_text().split("\n")[1:] == (tmp_path / "many.fa").read_text().split("\n")[1:]
    # the same reads as a file the device cannot take -- CRLF line ends -- go through the exact host reader: same result;
    # so does a file that turns irregular half-way (a wrapped record in its middle)
    import gzip
    text = gzip.open(os.path.join(DATA, "CHM13_reads.fq.gz"), "rb").read()
    (tmp_path / "crlf.fq").write_bytes(text.replace(b"\n", b"\r\n"))
    recs = text.split(b"\n")
    mid = (len(recs) // 8) * 4
    wrapped = recs[:mid] + [recs[mid], recs[mid + 1][:70], recs[mid + 1][70:], recs[mid + 2], recs[mid + 3][:70], recs[mid + 3][70:]] + recs[mid + 4:]
    (tmp_path / "wrapped.fq").write_bytes(b"\n".join(wrapped))
    for name in ("crlf.fq", "wrapped.fq"):
        r = _run_cli(["-t8", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", str(tmp_path / name), "-o", str(tmp_path / (name + ".fa"))], tmp_path,
                     env={"PHI_TIMING": "1", "PHI_READ_CHUNK": "700000"})
        assert r.returncode == 0, r.stderr
        assert "host reader from the first byte not taken" in r.stderr
        assert [l for l in lines(r.stderr) if "Loaded graph" not in l] == [l for l in lines(one.stderr) if "Loaded graph" not in l], name
        assert (tmp_path / (name + ".fa")).read_text().split("\n")[1:] == (tmp_path / "one.fa").read_text().split("\n")[1:]


def test_cli_parks_the_reads_text_in_device_memory_until_the_index_is_built(tmp_path):
    """One GPU, a large reads file: the chunks the reader thread has read before phi_set_graph is done wait in device memory
    (phi_text_park_*) instead of the reader waiting for a free buffer.  With PHI_TEXT_PARK_MIN=1 the fixture's 5.4 MB take that
    way in chunks of 100 kB: every log line and the FASTA are those of the run without (PHI_TEXT_PARK=0); a file that turns
    irregular half-way has its parked chunks fetched back for the exact host reader."""
    import gzip
    args = ["-t8", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz")]
    plain = _run_cli(args + ["-o", str(tmp_path / "plain.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_TEXT_PARK": "0", "PHI_READ_CHUNK": "100000"})
    parked = _run_cli(args + ["-o", str(tmp_path / "parked.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_TEXT_PARK_MIN": "1", "PHI_READ_CHUNK": "100000"})
    assert plain.returncode == 0 and parked.returncode == 0, plain.stderr + parked.stderr
    assert "waited in device memory" not in plain.stderr
    m = re.search(r"main: (\d+) bytes of the reads text waited in device memory for the index", parked.stderr)
    assert m and int(m.group(1)) >= 1_000_000, parked.stderr[-2000:]

    def lines(log):
        return [re.sub(r"^\[M::[^\]]*\] ", "", l) for l in log.splitlines()
                if not (l.startswith("[phi timing]") or "Real time" in l or "CMD:" in l or "written to" in l)]
    assert lines(plain.stderr) == lines(parked.stderr)
    assert "; 0 bases through the host reader" in parked.stderr
    assert (tmp_path / "plain.fa").read_text().split("\n")[1:] == (tmp_path / "parked.fa").read_text().split("\n")[1:]
    # irregular from the middle on: the parked chunks behind the irregular one come back to the host
    text = gzip.open(os.path.join(DATA, "CHM13_reads.fq.gz"), "rb").read()
    recs = text.split(b"\n")
    mid = (len(recs) // 8) * 4
    wrapped = recs[:mid] + [recs[mid], recs[mid + 1][:70], recs[mid + 1][70:], recs[mid + 2], recs[mid + 3][:70], recs[mid + 3][70:]] + recs[mid + 4:]
    (tmp_path / "wrapped.fq").write_bytes(b"\n".join(wrapped))
    r = _run_cli(["-t8", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", str(tmp_path / "wrapped.fq"), "-o", str(tmp_path / "wrapped.fa")], tmp_path,
                 env={"PHI_TIMING": "1", "PHI_TEXT_PARK_MIN": "1", "PHI_READ_CHUNK": "100000"})
    assert r.returncode == 0, r.stderr
    assert "host reader from the first byte not taken" in r.stderr and "waited in device memory" in r.stderr
    assert [l for l in lines(r.stderr) if "Loaded graph" not in l] == [l for l in lines(plain.stderr) if "Loaded graph" not in l]
    assert (tmp_path / "wrapped.fa").read_text().split("\n")[1:] == (tmp_path / "plain.fa").read_text().split("\n")[1:]


def test_cli_resolves_the_walks_on_the_device(tmp_path):
    """PHI_WALK_TEXT_MIN=1: the command line leaves the W-lines as text, uploads them from the reader's callback thread and has
    the device resolve them (the default takes this path from 1 GB of walk text on); log and FASTA of the run that resolves
    them on the host (PHI_WALKS=host)."""
    args = ["-t8", "-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz")]
    host = _run_cli(args + ["-o", str(tmp_path / "host.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_WALKS": "host"})
    dev = _run_cli(args + ["-o", str(tmp_path / "dev.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_WALK_TEXT_MIN": "1"})
    assert host.returncode == 0 and dev.returncode == 0, host.stderr + dev.stderr
    assert "resolved on the device" in dev.stderr and "resolved on the device" not in host.stderr

    def lines(log):
        return [re.sub(r"^\[M::[^\]]*\] ", "", l) for l in log.splitlines()
                if not (l.startswith("[phi timing]") or l.startswith("[phi]") or "Real time" in l or "CMD:" in l or "written to" in l)]
    assert lines(host.stderr) == lines(dev.stderr)
    assert (tmp_path / "host.fa").read_text().split("\n")[1:] == (tmp_path / "dev.fa").read_text().split("\n")[1:]


def test_cli_at_a_chromosome_arm_takes_the_device_walks_and_the_park_by_default(tmp_path):
    """200 walks over 20 Mbp from FILES (the native generator: 1.0 GB of GFA, 1.3 GB of FASTQ on the RAM disk): the command
    line resolves the W-lines on the device (0.9 GB of walk text: the threshold of 1 GB is lowered to 256 MB for the test) and,
    by its own threshold (>= 256 MB), parks the reads text in device memory while the index is built; log and FASTA equal the
    run with both switched off, the generator's truth walks come back."""
    import shutil
    import tempfile
    from phi_amd import synth
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS["C5n-mid"]
    d = tempfile.mkdtemp(prefix="phi_test_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        g = synth.NativeGraph(**gk)
        truth = g.sample(s_seed, n_mosaic)
        gfa, rd = os.path.join(d, "g.gfa"), os.path.join(d, "r.fq")
        assert g.write_gfa(gfa) > (900 << 20)
        assert g.write_reads(rd, r_seed, 0, g.n_reads(cov), fastq=True) > (256 << 20)
        names = [f"syn{h:03d}.{h % 2}" for h in truth["walks"]]
        g.close()
        dflt = _run_cli(["-t16", "-g", gfa, "-r", rd, "-o", os.path.join(d, "a.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_WALK_TEXT_MIN": str(256 << 20)})
        off = _run_cli(["-t16", "-g", gfa, "-r", rd, "-o", os.path.join(d, "b.fa")], tmp_path, env={"PHI_TIMING": "1", "PHI_WALKS": "host", "PHI_TEXT_PARK": "0"})
        assert dflt.returncode == 0 and off.returncode == 0, dflt.stderr[-3000:] + off.stderr[-3000:]
        assert "resolved on the device" in dflt.stderr and "waited in device memory" in dflt.stderr
        assert "resolved on the device" not in off.stderr and "waited in device memory" not in off.stderr

        def lines(log):
            return [re.sub(r"^\[M::[^\]]*\] ", "", l) for l in log.splitlines()
                    if not (l.startswith("[phi timing]") or l.startswith("[phi]") or "Real time" in l or "CMD:" in l or "written to" in l)]
        assert lines(dflt.stderr) == lines(off.stderr)
        assert open(os.path.join(d, "a.fa")).read() == open(os.path.join(d, "b.fa")).read()
        rec = [l for l in dflt.stderr.splitlines() if l.startswith("Recombined haplotypes")][0]
        assert re.findall(r"\((syn\d+\.\d),", rec) == names, (rec[:300], names)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_cli_errors(tmp_path):
    r = _run_cli([], tmp_path)
    assert r.returncode == 1 and r.stderr.startswith("Usage: PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta>")
    r = _run_cli(["-g", "nope.gfa", "-r", os.path.join(DATA, "read.fa"), "-o", "x.fa"], tmp_path)
    assert r.returncode == 1 and "failed to load the GFA file" in r.stderr
    r = _run_cli(["--version"], tmp_path)
    assert r.returncode == 0 and "PHI version" in r.stderr


def _write_gfa(g, path):
    """oracle.Graph -> GFA 1.1 text (S / L 0M / W), segment names as the graph holds them."""
    with open(path, "wb") as f:
        f.write(b"H\tVN:Z:1.1\n")
        for name, s in zip(g.seg_names, g.node_seq):
            f.write(b"S\t%s\t%s\n" % (name.encode(), s))
        for u, targets in enumerate(g.adj):
            for v in targets:
                f.write(b"L\t%s\t+\t%s\t+\t0M\n" % (g.seg_names[u].encode(), g.seg_names[v].encode()))
        for h, p in enumerate(g.paths):
            sample, _, hap = g.hap_names[h].rpartition(".")
            f.write(b"W\t%s\t%s\tchr\t0\t%d\t%s\n" % (sample.encode(), hap.encode(), sum(len(g.node_seq[v]) for v in p),
                                                     b"".join(b">" + g.seg_names[v].encode() for v in p)))


def test_cli_devices_list_and_run_budget(tmp_path, oracle):
    """--devices with one ordinal takes the multi-GPU code path's plumbing (context list, work queue) and
    gives the default run's output; --dp-budget counts DP runs: a hard instance (R = 0, eight walks, short
    repeats) stopped after 3 runs is written but the exit status (3) and a warning say it is not proven."""
    import numpy as np
    from graphgen import random_graph
    args = ["-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz")]
    a = _run_cli(args + ["-o", str(tmp_path / "a.fa")], tmp_path)
    b = _run_cli(args + ["-o", str(tmp_path / "b.fa"), "--devices", "0"], tmp_path)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert (tmp_path / "a.fa").read_text() == (tmp_path / "b.fa").read_text()
    bad = _run_cli(args + ["-o", str(tmp_path / "c.fa"), "--devices", "0,0"], tmp_path)
    assert bad.returncode == 1 and "twice" in bad.stderr
    # the hard instance of test_run_budget_is_deterministic_and_reports_a_proven_bound, from files
    rng = np.random.default_rng(321)
    g = random_graph(rng, n_sites=400, n_walks=8, seg_len=(8, 16), alt_len=(3, 6), p_del=0.0)
    succ = {v: sorted(t) for v, t in enumerate(g.adj)}
    v, truth, site = g.paths[0][0], [], 0
    while True:
        truth.append(v)
        nx = succ[v]
        if not nx:
            break
        v = nx[site % 2] if len(nx) == 2 else nx[0]
        site += len(nx) == 2
    hap = b"".join(g.node_seq[x] for x in truth)
    gfa, rd = str(tmp_path / "hard.gfa"), str(tmp_path / "hard.fa")
    _write_gfa(g, gfa)
    with open(rd, "wb") as f:
        for i, s in enumerate(rng.integers(0, len(hap) - 60, size=700)):
            f.write(b">r%d\n%s\n" % (i, hap[s:s + 60]))
    base = ["-g", gfa, "-r", rd, "-k7", "-w2", "-R0"]
    few = [_run_cli(base + ["-o", str(tmp_path / f"few{i}.fa"), "--dp-budget", "3"], tmp_path) for i in range(2)]
    for r in few:
        assert r.returncode == 3 and "NOT proven optimal" in r.stderr and "--dp-budget" in r.stderr, r.stderr[-2000:]
        assert "after 3 DP run(s)" in r.stderr
    assert (tmp_path / "few0.fa").read_text() == (tmp_path / "few1.fa").read_text()      # reproducible
    assert (tmp_path / "few0.fa").read_text().startswith(">hard_hard LN:")


def test_cli_through_the_vcf_route_and_the_log_scrape(tmp_path):
    """README.md:25-30 of the reference runs the same sample through vcf2gfa.py: here phi_amd/vcf2gfa.py makes the graph
    from test/MHC_4.vcf.gz + test/MHC-CHM13.0.fa.gz, PHI infers the haplotype of the CHM13 reads on it, and
    phi_amd/eval_log.py scrapes the log as data/postprocessing_2_MIQP.py:55-79 does (the inferred sequence is the
    reference walk: edit distance 0 to the CHM13 FASTA)."""
    import sys
    from phi_amd import eval_log
    gfa = tmp_path / "MHC_4_vcf.gfa"
    with open(gfa, "wb") as f:
        subprocess.check_call([sys.executable, "-m", "phi_amd.vcf2gfa", "-v", os.path.join(DATA, "MHC_4.vcf.gz"),
                               "-r", os.path.join(DATA, "MHC-CHM13.0.fa.gz")], stdout=f, cwd=ROOT)
    out = tmp_path / "CHM13_vcf.fa"
    r = _run_cli(["-t32", "-g", str(gfa), "-r", os.path.join(DATA, "CHM13_reads.fq.gz"), "-o", str(out)], tmp_path)
    assert r.returncode == 0, r.stderr
    got = eval_log.parse_log(r.stderr)
    assert all(v is not None for v in got.values()), got
    assert got["recombination_count"] == 0 and got["spectrum_size"] == 138834      # the reads' spectrum does not depend on the graph
    assert abs(got["pct_filtered"] + got["pct_retained"] - 100.0) < 0.02
    assert ">(REF.0,[0," in r.stderr
    truth, query = eval_log.read_fasta(os.path.join(DATA, "MHC-CHM13.0.fa.gz")), eval_log.read_fasta(str(out))
    assert query == truth and eval_log.edit_distance(truth[:20000], query[:20000]) == 0


def test_cli_several_read_sets_against_one_graph(tmp_path):
    """`PHI -g G -r a -o a.fa -r b -o b.fa -r c -o c.fa`: the graph is parsed and indexed once, every job prints the log
    of a run of its own and writes its own FASTA -- line for line and byte for byte what three separate commands give (the
    reference's harness runs PHI once per sample x coverage against the same graph, data/run_batch_4_miqp.py:31-46).
    Read sets of different sizes and layouts: the whole FASTQ, its first 2 000 records as FASTA, its last 500 as FASTQ."""
    import gzip
    gfa = os.path.join(DATA, "MHC_4.gfa.gz")
    fq = gzip.open(os.path.join(DATA, "CHM13_reads.fq.gz"), "rt").read().split("\n")
    recs = [fq[i:i + 4] for i in range(0, len(fq) - 1, 4)]
    a = os.path.join(DATA, "CHM13_reads.fq.gz")
    b = tmp_path / "first.fa"
    b.write_text("".join(f">{r[0][1:]}\n{r[1]}\n" for r in recs[:2000]))
    c = tmp_path / "last.fq"
    c.write_text("".join("\n".join(r) + "\n" for r in recs[-500:]))
    sets = [(a, tmp_path / "a.fa"), (str(b), tmp_path / "b.fa"), (str(c), tmp_path / "c.fa")]

    def keep(log):
        out = []
        for l in log.splitlines():
            if l.startswith("[phi timing]") or "Real time" in l or "CMD:" in l:
                continue
            out.append(re.sub(r"^\[M::[^\]]*\] ", "", l))
        return out
    singles = []
    for rd, out in sets:
        single = tmp_path / ("single_" + out.name)
        r = _run_cli(["-g", gfa, "-r", rd, "-o", str(single)], tmp_path)
        assert r.returncode == 0, r.stderr
        singles.append((keep(r.stderr.replace(str(single), str(out))), single.read_text()))
    args = ["-g", gfa]
    for rd, out in sets:
        args += ["-r", rd, "-o", str(out)]
    r = _run_cli(args, tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("Loaded graph from") == 3 and r.stderr.count("PHI Version") == 3
    want = [l for lines, _ in singles for l in lines]
    assert keep(r.stderr) == want
    for (rd, out), (_, fa) in zip(sets, singles):
        assert out.read_text() == fa
    # -r and -o come in pairs
    bad = _run_cli(["-g", gfa, "-r", a, "-r", str(b), "-o", str(tmp_path / "x.fa")], tmp_path)
    assert bad.returncode == 1 and "-r but" in bad.stderr
