"""GPU tests of the one-process multi-GPU exchange (phi_peers_*: one OR-gather kernel per GPU over peer-mapped hit vectors,
the lists of the other read hashes imported where they lie) -- on ONE GPU: two contexts on the same device, one host
thread each, which runs every line of the exchange but the xGMI loads themselves."""
import os
import threading

import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu


def _run_ranks(fns):
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:             # noqa: B902 (re-raised below)
            errs.append(e)
    th = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is stuck in the exchange"
    if errs:
        raise errs[0]


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_read_shards_merged_through_peer_mapped_memory(ctx_factory, oracle, n_ranks):
    """Config 1's reads in n shards on n contexts: after phi_peers_exchange every rank's solve is the single-context solve
    of all reads (hit vector, |Sp_R|, counters, path); phi_peers_allreduce_hits alone gives the union of the hit flags; a
    second read set goes through the same group."""
    import phi_amd
    import torch
    from phi_amd import dist as pdist
    from phi_amd import ilp_index as H
    g = oracle.parse_gfa(os.path.join(DATA, "MHC_4.gfa.gz"))
    A = g.arrays()
    bases, off, _ = H.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))

    def make():
        c = ctx_factory(k=31, w=25, threshold=1.0, recombination=100)
        c.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        return c

    def hits_of(c):
        p, n = c.hits_buffer()
        return torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy().copy()

    def key(res):
        return tuple(int(res[k]) for k in ("objective", "upper_bound", "optimal", "spectrum_size", "filtered", "retained", "n_in_model", "n_covered", "hap_len")) + \
            (res["path_vtx"].tobytes(), res["path_hap"].tobytes(), res["n_anchors"].tobytes())
    one = make()
    cs = [make() for _ in range(n_ranks)]
    group = phi_amd.Context.peers_create(n_ranks)
    _run_ranks([lambda c=c, r=r: c.peers_join(group, r) for r, c in enumerate(cs)])
    n = len(off) - 1
    for lo_frac, hi_frac in ((0.0, 1.0), (0.2, 0.7)):                  # two read sets, one after the other
        lo, hi = int(n * lo_frac), int(n * hi_frac)
        one.reset_reads()
        one.add_reads((bases[off[lo]:off[hi]], off[lo:hi + 1] - off[lo]))
        want_hits = hits_of(one)
        want = key(one.solve())
        cuts = np.linspace(lo, hi, n_ranks + 1).astype(int)
        for r, c in enumerate(cs):
            c.reset_reads()
            a, b = cuts[r], cuts[r + 1]
            c.add_reads((bases[off[a]:off[b]], off[a:b + 1] - off[a]))
        # step 1 alone
        _run_ranks([c.peers_allreduce_hits for c in cs])
        torch.cuda.synchronize()
        for c in cs:
            assert np.array_equal(hits_of(c), want_hits)
        # the whole exchange (its step 1 again: idempotent)
        _run_ranks([c.peers_exchange for c in cs])
        for c in cs:
            assert key(c.solve()) == want
    phi_amd.Context.peers_destroy(group)


def test_cli_two_contexts_take_the_chunks_in_turn(tmp_path):
    """`PHI --devices 0,0` (two contexts on the one GPU, allowed for this test): the chunks of the reads text go to the two
    contexts in turn, the unfinished rest of one chunk travelling to whichever takes the next, the hit vectors merge through
    peer-mapped memory -- and every log line and the FASTA are those of the one-GPU run.  A small reads file uses one GPU."""
    import re
    import subprocess
    from conftest import ROOT
    phi = os.path.join(ROOT, "phi_amd", "PHI")
    args = ["-g", os.path.join(DATA, "MHC_4.gfa.gz"), "-r", os.path.join(DATA, "CHM13_reads.fq.gz")]

    def run(extra, env):
        return subprocess.run([phi] + args + extra, capture_output=True, text=True, cwd=str(tmp_path), timeout=300, env=dict(os.environ, **env))

    def lines(log):
        keep = []
        for l in log.splitlines():
            if l.startswith("[phi timing]") or "Real time" in l or "CMD:" in l or "written to" in l or l.startswith("[M::main] 2 GPUs") or "reads file of" in l:
                continue
            keep.append(re.sub(r"^\[M::[^\]]*\] ", "", l))
        return keep
    one = run(["-o", str(tmp_path / "one.fa")], {})
    assert one.returncode == 0, one.stderr
    two = run(["-o", str(tmp_path / "two.fa"), "--devices", "0,0", "--shard-min-bases", "100000"],
              {"PHI_ALLOW_SAME_DEVICE": "1", "PHI_EXCHANGE": "peers", "PHI_READ_CHUNK": "300001", "PHI_TIMING": "1"})      # (peers: opt-in -- and RCCL takes no two ranks on one GPU)
    assert two.returncode == 0, two.stderr
    assert "2 GPUs; hit vector of" in two.stderr and "peer-mapped memory" in two.stderr
    assert re.search(r"main: (\d+) text chunk\(s\)", two.stderr) and int(re.search(r"main: (\d+) text chunk\(s\)", two.stderr).group(1)) >= 15
    assert lines(one.stderr) == lines(two.stderr)
    assert (tmp_path / "one.fa").read_text() == (tmp_path / "two.fa").read_text()
    # a reads file not worth a second GPU: one is used, and the log says so
    few = run(["-o", str(tmp_path / "few.fa"), "--devices", "0,0"], {"PHI_ALLOW_SAME_DEVICE": "1", "PHI_EXCHANGE": "peers"})
    assert few.returncode == 0 and "using 1 of the 2 GPUs given" in few.stderr
    assert (tmp_path / "one.fa").read_text() == (tmp_path / "few.fa").read_text()
