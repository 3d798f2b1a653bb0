"""The walks of a GFA resolved on the device from the text of its W-lines (phi_amd/csrc/walk_text.hip) against the host
reader, whose rules are the reference's (gfa-io.cpp:367-432 parses a walk, :64-115 flips walks by majority strand,
ILP_index.cpp:96-113 copies the vertices into paths[h]).

The contract: what the device path accepts it resolves exactly as the host reader does; anything else (a reverse step, a
name that is not <prefix><canonical number>, a step naming no segment, a W-line among the S-lines) it REFUSES as a whole,
and the caller has the host resolve the walks -- never a third result."""
import os

import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu


def _ctx(ctx_factory, **kw):
    p = dict(k=3, w=2, threshold=1.0, recombination=100)
    p.update(kw)
    return ctx_factory(**p)


def _device_walks(ctx, path):
    """(DeferredGraph, went to the device)"""
    from phi_amd import ilp_index as H
    g = H.DeferredGraph(path)
    ok = g.resolve_on_device(ctx)
    if not ok:
        g.resolve_on_host()
    return g, ok


def _same_graph(g, want):
    assert g.hap_id2name == want.hap_id2name
    assert g.seq_off.tolist() == want.seq_off.tolist() and bytes(g.seq_concat) == bytes(want.seq_concat)
    assert g.adj_off.tolist() == want.adj_off.tolist() and g.adj.tolist() == want.adj.tolist()
    assert g.top_order_map.tolist() == want.top_order_map.tolist()
    assert g.walk_off.tolist() == want.walk_off.tolist()


@pytest.mark.parametrize("name", ["test.gfa", "MHC_4.gfa.gz"])
def test_device_walks_equal_the_host_readers(ctx_factory, name):
    from phi_amd import ilp_index as H
    path = os.path.join(DATA, name)
    want = H.Graph(path)
    ctx = _ctx(ctx_factory)
    g, on_dev = _device_walks(ctx, path)
    assert on_dev and g.walk_vtx is None
    _same_graph(g, want)
    got = ctx.walk_entries()
    assert got.shape == want.walk_vtx.shape and np.array_equal(got, want.walk_vtx)


def test_index_and_solve_from_device_walks(ctx_factory):
    """phi_set_graph(walk_vtx = NULL) over the entries the device resolved: the same index, anchors and recombination
    count as over the host reader's arrays (the reference's MHC_4 graph and CHM13 reads, README's test command)."""
    from phi_amd import ilp_index as H
    path = os.path.join(DATA, "MHC_4.gfa.gz")
    bases, off, _ = H.read_reads(os.path.join(DATA, "CHM13_reads.fq.gz"))
    out = []
    for deferred in (False, True):
        ctx = _ctx(ctx_factory, k=31, w=25, threshold=0.9, recombination=10000)
        if deferred:
            g, on_dev = _device_walks(ctx, path)
            assert on_dev
            g.set_graph(ctx)
        else:
            g = H.Graph(path)
            ctx.set_graph(g.seq_concat, g.seq_off, g.adj_off, g.adj, g.walk_off, g.walk_vtx, g.top_order_map)
        ctx.add_reads((bases, off))
        info = ctx.index_stats()
        res = ctx.solve()
        out.append((info["n_entries"], info["n_classes"], info["n_distinct_minimizers"], info["n_walk_minimizers"], res["n_minimizers"].tolist(),
                    res["n_anchors"].tolist(), res["recombination_count"], res["objective"], res["path_vtx"].tolist(), res["path_hap"].tolist()))
    assert out[0] == out[1]


def _write(tmp_path, name, segs, links, walks):
    p = tmp_path / name
    with open(p, "w") as f:
        f.write("H\tVN:Z:1.1\n")
        for n, s in segs:
            f.write(f"S\t{n}\t{s}\n")
        for a, b in links:
            f.write(f"L\t{a}\t+\t{b}\t+\t0M\n")
        for i, w in enumerate(walks):
            f.write(f"W\tsmp{i}\t{i % 2}\tchr\t0\t1\t{w}\n")
    return str(p)


def _chain(n, prefix="s", first=1):
    rng = np.random.default_rng(n)
    segs = [(f"{prefix}{first + i}", "".join("ACGT"[x] for x in rng.integers(0, 4, 1 + i % 5))) for i in range(n)]
    links = [(segs[i][0], segs[j][0]) for i in range(n) for j in (i + 1, i + 2) if j < n]
    return segs, links


def test_tags_behind_a_walk_are_not_steps(ctx_factory, tmp_path):
    from phi_amd import ilp_index as H
    src = open(os.path.join(DATA, "test.gfa")).read().splitlines()
    tagged = [l + "\tXX:Z:a>s3<s9>c\tYY:i:7" if l.startswith("W\t") and i % 2 == 0 else l for i, l in enumerate(src)]
    (tmp_path / "tagged.gfa").write_text("\n".join(tagged) + "\n")
    want = H.Graph(os.path.join(DATA, "test.gfa"))
    ctx = _ctx(ctx_factory)
    g, on_dev = _device_walks(ctx, str(tmp_path / "tagged.gfa"))
    assert on_dev
    _same_graph(g, want)
    assert np.array_equal(ctx.walk_entries(), want.walk_vtx)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_long_walks_across_tiles(ctx_factory, tmp_path, seed):
    """Walks of many 4-KB tiles (names of 2..7 bytes straddle the tile borders at every offset), walks of one step, a walk
    whose text ends exactly on a tile border, names without a prefix at all."""
    from phi_amd import ilp_index as H
    rng = np.random.default_rng(seed)
    n = 30000
    prefix = ["s", "", "seg_"][seed]
    segs, links = _chain(n, prefix)
    walks = []
    for wi in range(9):
        if wi == 3:
            steps = [int(rng.integers(0, n))]
        else:
            lo = int(rng.integers(0, 50))
            steps, v = [], lo
            while v < n and len(steps) < [20000, 700, 4000][wi % 3]:
                steps.append(v)
                v += int(rng.integers(1, 3))
        walks.append("".join(f">{segs[v][0]}" for v in steps))
    # one walk cut to a whole number of tiles (its last name ends on the border)
    w = walks[0]
    cut = w.rfind(">", 0, 8192 - 3)
    name_len = 8192 - cut
    if 2 <= name_len - 1 - len(prefix) <= 5:
        lo_num = 10 ** (name_len - 2 - len(prefix))
        walks.append(w[:cut] + f">{prefix}{lo_num + 7}")
        assert len(walks[-1]) == 8192
    path = _write(tmp_path, "long.gfa", segs, links, walks)
    want = H.Graph(path)
    ctx = _ctx(ctx_factory)
    g, on_dev = _device_walks(ctx, path)
    assert on_dev
    _same_graph(g, want)
    assert np.array_equal(ctx.walk_entries(), want.walk_vtx)


CASES = {
    # what the device path refuses -> (walks, why)
    "a reverse walk": ([">s1>s2>s3", "<s3<s2<s1"], 1),
    "a step naming no segment": ([">s1>s2>s3", ">s1>s99>s3"], 2),
    "a number with a leading zero": ([">s1>s2>s3", ">s1>s02>s3"], 2),
    "a name with another prefix": ([">s1>s2>s3", ">s1>t2>s3"], 2),
    "a name of ten digits": ([">s1>s2>s3", ">s1>s1000000002>s3"], 2),
    "a name that goes on behind its number": ([">s1>s2>s3", ">s1>s2x>s3"], 2),
    "a bare prefix": ([">s1>s2>s3", ">s1>s>s3"], 2),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_irregular_walk_text_is_refused_whole_and_the_host_resolves_it(ctx_factory, tmp_path, case):
    from phi_amd import ilp_index as H
    walks, bit = CASES[case]
    segs, links = _chain(6)
    path = _write(tmp_path, "irr.gfa", segs, links, walks)
    want = H.Graph(path)
    ctx = _ctx(ctx_factory)
    g = H.DeferredGraph(path)
    assert not g.resolve_on_device(ctx) and g.irregular & bit
    assert ctx.walk_entries().size == 0                       # nothing was left behind
    g.resolve_on_host()
    _same_graph(g, want)
    assert g.walk_vtx.tolist() == want.walk_vtx.tolist()
    g.set_graph(ctx)                                          # and the context takes the host's walks as ever
    assert ctx.index_stats()["n_entries"] == len(want.walk_vtx)


def test_names_of_another_form_never_reach_the_device(ctx_factory, tmp_path):
    """Segment names that are not <prefix><number> (or a W-line standing before an S-line): the reader has no direct name
    index to give (phi_graph_name_index = PHI_HOST_ERR_UNSUPPORTED) and the walks are the host's."""
    from phi_amd import ilp_index as H
    segs = [("chrA_1", "ACGT"), ("x", "GG"), ("s3", "TTA")]
    path = _write(tmp_path, "names.gfa", segs, [("chrA_1", "x"), ("x", "s3")], [">chrA_1>x>s3", ">chrA_1>x"])
    want = H.Graph(path)
    ctx = _ctx(ctx_factory)
    g, on_dev = _device_walks(ctx, path)
    assert not on_dev
    _same_graph(g, want)
    assert g.walk_vtx.tolist() == want.walk_vtx.tolist()
    # a W-line among the S-lines
    p = tmp_path / "order.gfa"
    p.write_text("S\ts1\tACGT\nS\ts2\tGGA\nW\ta\t0\tc\t0\t1\t>s1>s2>s3\nS\ts3\tTT\nL\ts1\t+\ts2\t+\t0M\nL\ts2\t+\ts3\t+\t0M\n")
    want = H.Graph(str(p))
    g, on_dev = _device_walks(ctx, str(p))
    assert not on_dev
    _same_graph(g, want)
    assert g.walk_vtx.tolist() == want.walk_vtx.tolist()


def test_upload_handed_over_by_the_reader_itself(ctx_factory):
    """phi_gfa_read_deferred's on_text callback (what the CLI uses: the text is on its way to HBM while the reader still
    enters the segment names) gives the same walks as uploading afterwards."""
    import ctypes as C
    from phi_amd import ilp_index as H
    path = os.path.join(DATA, "MHC_4.gfa.gz")
    want = H.Graph(path)
    ctx = _ctx(ctx_factory)
    seen = []

    def on_text(user, walks, n):
        seen.append(n)
        ctx._chk(ctx._L.phi_walk_text_upload(ctx._h, walks, n))

    g = H.DeferredGraph(path, on_text=on_text)
    assert seen == [want.num_walks]
    assert g.resolve_on_device(ctx, upload=False)
    assert np.array_equal(ctx.walk_entries(), want.walk_vtx)
    g.set_graph(ctx)
    assert ctx.index_stats()["n_entries"] == len(want.walk_vtx)
