"""CPU test of the N>1 path with world_size 2 over gloo: the exchange code of phi_amd/dist.py
(hit-vector all-reduce, spectrum merge, read sharding) on data produced by the oracle, checked
against the single-process result on the union of the reads."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from graphgen import mosaic_reads, random_graph
    from oracle import oracle as O
    from phi_amd import dist as pdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)                     # same graph and read set on every rank
    g = random_graph(rng, n_sites=10, n_walks=5, seg_len=(10, 40))
    reads = mosaic_reads(rng, g, n_reads=80, read_len=40, n_seg=3, err=0.02)
    k, w = 9, 5
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    lo, hi = pdist.shard_bounds(off, world, rank)
    mine = reads[lo:hi]
    # per-rank state exactly as the GPU path keeps it: distinct walk minimisers in position order
    walk_h = np.concatenate([O.sketch(b"".join(g.node_seq[v] for v in p), k, w)[0] for p in g.paths])
    _, first = np.unique(walk_h, return_index=True)
    uniq = walk_h[np.sort(first)]                       # dense id = rank of first occurrence
    my_hashes = np.unique(np.concatenate([O.sketch(r, k, w)[0] for r in mine] + [np.zeros(0, np.uint64)]))
    hit = torch.from_numpy(np.isin(uniq, my_hashes).astype(np.uint8))
    pdist.allreduce_hits(hit)
    parts = pdist.gather_spectra(torch.from_numpy(my_hashes.view(np.int64).copy()))
    union = np.unique(np.concatenate([p.numpy().view(np.uint64) for p in parts]))
    # single-process truth on all reads
    all_hashes = np.unique(np.concatenate([O.sketch(r, k, w)[0] for r in reads]))
    ok = bool(np.array_equal(hit.numpy(), np.isin(uniq, all_hashes).astype(np.uint8))) and bool(np.array_equal(union, all_hashes))
    ok = ok and (lo, hi) != (0, len(reads)) and len(mine) > 0
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {lo} {hi} {int(hit.sum())} {len(union)}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_over_gloo(oracle, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    rows = [open(tmp_path / f"rank{r}.txt").read().split() for r in range(2)]
    assert rows[0][0] == rows[1][0] == "1", rows
    assert rows[0][3:] == rows[1][3:]                   # identical merged state on both ranks
    assert int(rows[0][2]) == int(rows[1][1])           # contiguous shards


def test_shard_bounds_cover_everything():
    from phi_amd import dist as pdist
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 300, size=1000)
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    for world in (1, 2, 3, 8):
        b = [pdist.shard_bounds(off, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == len(lens)
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [off[h] - off[l] for l, h in b]
        assert max(sizes) - min(sizes) <= 2 * lens.max()
