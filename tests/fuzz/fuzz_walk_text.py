"""Randomised comparison of the walks resolved ON THE DEVICE from the W-lines' text (phi_amd/csrc/walk_text.hip, through
phi_gfa_read_deferred + phi_walk_text_*) with the host reader (whose rules are the reference's) on the random GFA texts of
fuzz_gfa_reader.py -- shuffled lines, arbitrary names, reversed walks, CRLF, gzip, tags -- and on larger regular ones whose
walks cross many 4-KB tiles.  The contract: the device either resolves exactly what the host reader resolves or refuses the
file as a whole (then the host resolves it through the same handle): never a third result.
Usage (GPU box): python tests/fuzz/fuzz_walk_text.py SEED SECONDS     -- not collected by pytest."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
import torch  # noqa: F401  (initialises the GPU runtime before the library is loaded)

torch.cuda.init()
from graphgen import random_graph  # noqa: E402
from fuzz_gfa_reader import write_random_gfa  # noqa: E402
import phi_amd  # noqa: E402
from phi_amd import ilp_index as H  # noqa: E402


def big_regular(rng, path):
    """a chain with bubbles, thousands of segments, walks of 10^3..10^5 steps, optional tags"""
    n = int(rng.integers(2000, 40000))
    prefix = str(rng.choice(["s", "", "seg_", "utg"]))
    first = int(rng.integers(1, 1000))
    with open(path, "w") as f:
        for i in range(n):
            f.write(f"S\t{prefix}{first + i}\t{'ACGT'[i % 4] * (1 + i % 3)}\n")
        for i in range(n - 1):
            f.write(f"L\t{prefix}{first + i}\t+\t{prefix}{first + i + 1}\t+\t0M\n")
            if i + 2 < n:
                f.write(f"L\t{prefix}{first + i}\t+\t{prefix}{first + i + 2}\t+\t0M\n")
        for h in range(int(rng.integers(1, 12))):
            v, steps = int(rng.integers(0, 20)), []
            stop = int(rng.integers(1, n))
            while v < stop:
                steps.append(v)
                v += int(rng.integers(1, 3))
            body = "".join(f">{prefix}{first + x}" for x in steps) or f">{prefix}{first}"
            tag = "\tXX:Z:>1<2" if rng.random() < 0.3 else ""
            f.write(f"W\tsmp{h}\t{h % 2}\tchr\t0\t1\t{body}{tag}\n")


def main():
    rng = np.random.default_rng(int(sys.argv[1]))
    t_end = time.time() + float(sys.argv[2])
    ctx = phi_amd.Context(0)
    ctx.set_params(k=3, w=2, threshold=1.0, recombination=100)
    n = n_dev = n_big = 0
    with tempfile.TemporaryDirectory() as td:
        while time.time() < t_end:
            big = rng.random() < 0.15
            path = os.path.join(td, "g.gfa.gz" if (not big and rng.random() < 0.3) else "g.gfa")
            if big:
                big_regular(rng, path)
            else:
                g = random_graph(rng, n_sites=int(rng.integers(1, 12)), n_walks=int(rng.integers(1, 7)), seg_len=(1, int(rng.integers(2, 40))),
                                 alt_len=(1, int(rng.integers(2, 12))), p_del=float(rng.choice([0, 0.3])))
                write_random_gfa(rng, g, path, False)
            try:
                want = H.Graph(path)
            except H.HostError as e:
                # the host refuses the file: the deferred reader, or the host resolution behind it, refuses it the same way
                try:
                    d = H.DeferredGraph(path)
                    if not d.resolve_on_device(ctx):
                        d.resolve_on_host()
                except H.HostError as e2:
                    assert e2.args == e.args, (e.args, e2.args)
                    n += 1
                    continue
                raise AssertionError(("host refused, deferred path accepted", e.args, open(path, "rb").read()[:2000]))
            d = H.DeferredGraph(path)
            on_dev = d.resolve_on_device(ctx)
            if on_dev:
                got = ctx.walk_entries()
                n_dev += 1
                n_big += big
            else:
                d.resolve_on_host()
                got = d.walk_vtx
            head = open(path, "rb").read()[:1500]
            assert d.walk_off.tolist() == want.walk_off.tolist(), head
            assert np.array_equal(got, want.walk_vtx), head
            assert d.hap_id2name == want.hap_id2name and np.array_equal(d.adj, want.adj) and np.array_equal(d.top_order_map, want.top_order_map), head
            assert not big or on_dev, "a regular file was refused"
            d.close()
            n += 1
    print(f"fuzz ok: {n} GFA files, {n_dev} resolved on the device ({n_big} large regular ones)")


if __name__ == "__main__":
    main()
