"""One rank of a group of PROCESSES that exchange their hit vectors through mapped memory (phi_amd/csrc/phi_ipc.hip):
the body of tests/test_gpu_ipc.py's child processes (started with the `spawn` method: nothing of the parent's GPU state)."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_case(seed, n_steps):
    """The graph and the read set of every step, the same in every process."""
    from graphgen import mosaic_reads, random_graph
    rng = np.random.default_rng(seed)
    g = random_graph(rng, n_sites=60, n_walks=6, seg_len=(20, 70), alt_len=(1, 9), p_del=0.2)
    sets = []
    for s in range(n_steps):
        r = mosaic_reads(rng, g, n_reads=240, read_len=110, n_seg=3, err=0.01)
        r += [bytes(rng.choice(list(b"ACGT"), size=200).tolist()) for _ in range(20)]      # novel hashes: the lists of step 2
        sets.append(r)
    return g, sets


def run_rank(rank, n_ranks, uid, seed, n_steps, k, w, queue, skip_step=-1):
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch                                             # (torch bundles its own HIP runtime: it must initialise first, as in conftest.py)
        torch.cuda.init()
        import phi_amd
        g, sets = make_case(seed, n_steps)
        A = g.arrays()
        ctx = phi_amd.Context(0)
        ctx.set_params(k=k, w=w, threshold=1.0, recombination=6)
        ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
        ctx.ipc_init(uid, rank, n_ranks)
        assert ctx.ipc_info() == (rank, n_ranks)
        out = {"rank": rank, "hits": {}, "error": None}
        # ---- read sets one after the other, the gather of set i beside the scoring of set i + 1; the hit vector is looked
        #      at after some of them only (looking makes the context's stream wait for the gather)
        for s in range(n_steps - 1):
            ctx.reset_reads()
            ctx.add_reads(sets[s][rank::n_ranks])
            if s == skip_step and rank == n_ranks - 1:
                continue                                          # (a rank that does not come: the others must give up, not hang)
            ctx.ipc_allreduce_hits()
            if s in (0, 3, n_steps - 2):
                from phi_amd import dist as pdist
                p, n = ctx.hits_buffer()                           # (makes the context's stream wait for the gather ...)
                ctx.device_synchronize()                           # (... and torch's copy below runs on another stream)
                out["hits"][s] = torch.as_tensor(pdist.DevArray(p, n), device="cuda").cpu().numpy().copy()
        if skip_step >= 0:
            try:
                ctx.ipc_check()
                out["check"] = "ok"
            except phi_amd.PhiError as e:
                out["check"] = f"error {e.status}"
            queue.put(out)
            ctx.close()
            return
        ctx.ipc_check()
        # ---- the job's exchange: hit vectors and the lists of novel read hashes, then the solve
        ctx.reset_reads()
        ctx.add_reads(sets[n_steps - 1][rank::n_ranks])
        ctx.ipc_exchange()
        st = ctx.reads_stats()
        res = ctx.solve()
        out["stats"] = st["n_distinct"]
        out["res"] = {k_: (v.tolist() if hasattr(v, "tolist") else v) for k_, v in res.items()}
        ctx.ipc_destroy()
        # the context works on alone afterwards
        ctx.reset_reads()
        ctx.add_reads(sets[0])
        out["alone"] = ctx.solve()["spectrum_size"]
        ctx.close()
        queue.put(out)
    except Exception:
        queue.put({"rank": rank, "error": traceback.format_exc()})
